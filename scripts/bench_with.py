#!/usr/bin/env python3
"""Run bench.py in this process with test-hook attributes of the package set first (same-call A/B of decisions that are no longer
environment switches):   python scripts/bench_with.py functional._DGRAD_FIRST=1 model._FANOUT=False -- --no-cpu-baseline --repeats 12
Values are Python literals; a bare word is taken as a string."""
import ast
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
sets, rest = args[:cut], args[cut + 1:]
import mau_amd  # noqa: E402
import importlib  # noqa: E402
for s in sets:
    target, val = s.split("=", 1)
    mod, attr = target.rsplit(".", 1)
    m = importlib.import_module("mau_amd." + mod)
    assert hasattr(m, attr), f"mau_amd.{mod} has no attribute {attr}"
    try:
        v = ast.literal_eval(val)
    except (ValueError, SyntaxError):
        v = val
    setattr(m, attr, v)
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
runpy.run_path(sys.argv[0], run_name="__main__")
