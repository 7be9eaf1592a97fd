#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a `hipcc -S --cuda-device-only` listing:
    python scripts/isa_blocks.py /tmp/bn_fused.s head_bn_bwd_reduce_kernelIDF16bLi2ELb1 [min_block_size]
(how the streaming kernels' VALU issue was read: blocks of >= min instructions, their opcode-family histogram)."""
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 60
s = open(path).read()
names = [l.split(":")[0] for l in s.splitlines() if key in l and l.startswith("_Z") and ": " in l and "@" in l]
assert names, "no function matches " + key
for name in names:
    a = s.index(name + ":")
    body = s[a:s.index(".Lfunc_end", a)].split("\n")
    blocks, cur = [], None
    for l in body:
        l = l.strip()
        if not l or l.startswith(";") or (l.startswith(".") and not l.endswith(":") and not l.startswith(".LBB")):
            continue
        if l.endswith(":") or (l.startswith(".LBB") and ":" in l.split()[0]):
            cur = [l.split()[0], []]
            blocks.append(cur)
            continue
        if cur is None:
            cur = ["entry", []]
            blocks.append(cur)
        cur[1].append(l.split()[0])
    print(name, "total", sum(len(b[1]) for b in blocks))
    for n, ins in blocks:
        if len(ins) >= minb:
            c = Counter("_".join(i.split("_")[:2]) for i in ins)
            print("  ", n, len(ins), c.most_common(16))
