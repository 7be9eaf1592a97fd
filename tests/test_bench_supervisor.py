"""bench.py's supervisor (CPU only, fake workers): a data-parallel measurement must survive a capture that fails or hangs.

The supervisor never touches the GPU; it starts the workers of attempt 1 with MAU_DP_GRAPH=1, watches their stage lines,
and on a non-zero exit OR a stall kills the whole attempt and starts a FRESH set with MAU_DP_GRAPH=0 (bench.py docstring).
The fake worker below plays a rank: it behaves as told through MAU_FAKE_* and prints bench.py's stage lines / JSON line.
"""
import io
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports no torch: a supervisor stays off it)

FAKE = textwrap.dedent(r'''
    import json, os, sys, time
    tag = "[mau-bench-stage]"
    graph = os.environ["MAU_DP_GRAPH"] == "1"
    rank = int(os.environ["RANK"])
    mode = os.environ.get("MAU_FAKE_GRAPH_FAILURE", "none")      # what the CAPTURED attempt does on rank MAU_FAKE_BAD_RANK
    bad = rank == int(os.environ.get("MAU_FAKE_BAD_RANK", "0"))
    for st in ("spawned", "imported", "ready", "warm"):
        print(tag, st, file=sys.stderr, flush=True)
    print("some library banner on stdout")                       # RCCL prints one: the supervisor must not take it for the line
    if graph and mode == "exit":
        if bad:
            sys.exit(1)
        time.sleep(600)                                          # (the peers of a dead rank hang in their next collective)
    direct = os.environ.get("MAU_RCCL_DIRECT", "1") != "0"
    if direct and os.environ.get("MAU_FAKE_DIRECT_FAILURE") == "exit":
        if bad:
            sys.exit(1)                                          # the directly-called RCCL path fails, captured or eager
        time.sleep(600)
    if (graph and mode == "hang") or mode == "hang_always":
        if bad:
            time.sleep(600)                                      # the rank that hangs in its capture ...
        print(tag, "captured", file=sys.stderr, flush=True)
        time.sleep(600)                                          # ... and its peers, stuck in the next collective
    print(tag, "captured", file=sys.stderr, flush=True)
    print(tag, "timed", file=sys.stderr, flush=True)
    line = {"metric": "fake", "value": 1.0, "launch": "graph" if graph else "eager", "port": os.environ["MASTER_PORT"],
            "collectives": "direct" if direct else "pg"}
    if rank == 0 and os.environ.get("MAU_FAKE_EARLY_LINE") == "1":
        print(json.dumps(dict(line, roofline="recorded")), flush=True)      # bench.py: the line goes out BEFORE the event pass
    if graph and mode == "hang_in_timed":
        time.sleep(600)                                          # the post-region event pass runs collectives again and hangs
    if rank == 0:
        print(json.dumps(dict(line, roofline="live")), flush=True)
    print(tag, "done", file=sys.stderr, flush=True)
    if os.environ.get("MAU_FAKE_EXIT_AFTER_DONE") == "1":
        sys.exit(3)                                              # e.g. a crash in destroy_process_group: the result is already out
''')


def _attempts(tmp_path, world, extra_env):
    script = tmp_path / "fake_worker.py"
    script.write_text(FAKE)
    made = []

    def make_attempt(k, dp_graph, port=None):
        made.append((k, dp_graph))
        out = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT=str(29000 + k), MAU_DP_GRAPH="1" if dp_graph else "0", **extra_env)
            out.append(([sys.executable, str(script)], env))
        return out

    return make_attempt, made


def _run(tmp_path, capsys, monkeypatch, world=2, **extra_env):
    monkeypatch.delenv("MAU_DP_GRAPH", raising=False)
    monkeypatch.delenv("MAU_RCCL_DIRECT", raising=False)
    make_attempt, made = _attempts(tmp_path, world, extra_env)
    log = io.StringIO()
    rc = bench.run_supervised(make_attempt, log=log)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    return rc, made, lines, log.getvalue()


def test_graph_attempt_succeeds_no_fallback(tmp_path, capsys, monkeypatch):
    rc, made, lines, log = _run(tmp_path, capsys, monkeypatch)
    assert rc == 0 and made == [(0, True)]
    assert len(lines) == 1 and json.loads(lines[0])["launch"] == "graph"
    assert "banner" in log                                        # non-JSON stdout goes to the log, never to stdout


def test_falls_back_when_a_rank_exits_nonzero(tmp_path, capsys, monkeypatch):
    rc, made, lines, log = _run(tmp_path, capsys, monkeypatch, MAU_FAKE_GRAPH_FAILURE="exit", MAU_FAKE_BAD_RANK="1")
    assert rc == 0 and made == [(0, True), (1, False)]
    rec = json.loads(lines[0])
    assert len(lines) == 1 and rec["launch"] == "eager" and rec["port"] == "29001"          # fresh workers, fresh rendezvous
    assert "exited with code 1" in log and "falling back" in log


def test_falls_back_when_a_rank_hangs(tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("MAU_BENCH_STALL_SCALE", "0.01")            # stage limits of 180 s -> 1.8 s
    rc, made, lines, log = _run(tmp_path, capsys, monkeypatch, MAU_FAKE_GRAPH_FAILURE="hang", MAU_FAKE_BAD_RANK="0")
    assert rc == 0 and made == [(0, True), (1, False)]
    assert len(lines) == 1 and json.loads(lines[0])["launch"] == "eager"
    assert "made no progress" in log


def test_third_attempt_goes_through_process_group_nccl(tmp_path, capsys, monkeypatch):
    # the directly-called RCCL path fails in both launch modes: the last attempt runs eagerly with MAU_RCCL_DIRECT=0
    rc, made, lines, log = _run(tmp_path, capsys, monkeypatch, MAU_FAKE_DIRECT_FAILURE="exit", MAU_FAKE_BAD_RANK="1")
    assert rc == 0 and made == [(0, True), (1, False), (2, False)]
    rec = json.loads(lines[0])
    assert len(lines) == 1 and rec["launch"] == "eager" and rec["collectives"] == "pg" and rec["port"] == "29002"
    assert "over ProcessGroupNCCL" in log
    # MAU_RCCL_DIRECT pinned by the caller: no third attempt
    monkeypatch.setenv("MAU_RCCL_DIRECT", "1")
    make_attempt, made = _attempts(tmp_path, 2, {"MAU_FAKE_DIRECT_FAILURE": "exit"})
    assert bench.run_supervised(make_attempt, log=io.StringIO()) == 1 and made == [(0, True), (1, False)]


def test_exit_code_after_done_does_not_discard_the_result(tmp_path, capsys, monkeypatch):
    rc, made, lines, _ = _run(tmp_path, capsys, monkeypatch, MAU_FAKE_EXIT_AFTER_DONE="1")
    assert rc == 0 and made == [(0, True)] and json.loads(lines[0])["launch"] == "graph"


def test_pinned_mode_and_total_failure(tmp_path, capsys, monkeypatch):
    # MAU_DP_GRAPH=0 in the environment: eager attempts only
    monkeypatch.delenv("MAU_RCCL_DIRECT", raising=False)
    monkeypatch.setenv("MAU_DP_GRAPH", "0")
    make_attempt, made = _attempts(tmp_path, 2, {})
    assert bench.run_supervised(make_attempt, log=io.StringIO()) == 0 and made == [(0, False)]
    assert json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][0])["launch"] == "eager"
    # pinned to the captured step and it fails: no fallback, non-zero exit, no line
    monkeypatch.setenv("MAU_DP_GRAPH", "1")
    make_attempt, made = _attempts(tmp_path, 2, {"MAU_FAKE_GRAPH_FAILURE": "exit"})
    assert bench.run_supervised(make_attempt, log=io.StringIO()) == 1 and made == [(0, True)]
    assert not [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]


def test_a_hang_after_the_timed_regions_does_not_cost_the_line(tmp_path, capsys, monkeypatch):
    """rank 0 prints its line when the timed regions are over; the per-kernel event pass that follows runs the
    collectives again -- if THAT hangs, the supervisor ends the attempt and forwards the line it already holds (no fallback)."""
    monkeypatch.setenv("MAU_BENCH_STALL_SCALE", "0.01")
    rc, made, lines, log = _run(tmp_path, capsys, monkeypatch, MAU_FAKE_GRAPH_FAILURE="hang_in_timed", MAU_FAKE_EARLY_LINE="1")
    assert rc == 0 and made == [(0, True)]
    rec = json.loads(lines[0])
    assert len(lines) == 1 and rec["launch"] == "graph" and rec["roofline"] == "recorded"
    assert "made no progress" in log and "stage 'timed'" in log and "stands" in log
    # ... and the forwarded line SAYS so, machine-readably: the late failure is not reported as a clean success
    assert rec["attempt_failed_after_timed_regions"] is True and "stage 'timed'" in rec["failure"]
    # the normal case: two lines from the worker, the LAST (complete) one is forwarded
    rc, made, lines, log = _run(tmp_path, capsys, monkeypatch, MAU_FAKE_EARLY_LINE="1")
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0])["roofline"] == "live"
    assert "attempt_failed_after_timed_regions" not in json.loads(lines[0])


def test_every_attempt_hanging_ends_inside_the_budget(tmp_path, capsys, monkeypatch):
    """Three attempts that all hang: the run ends with rc 1 inside MAU_BENCH_BUDGET_S (one budget for all attempts; each
    attempt's stage limits are clipped to its share)."""
    import time
    monkeypatch.setenv("MAU_BENCH_STALL_SCALE", "0.01")            # budget 540 s -> 5.4 s, MIN_ATTEMPT_S 100 -> 1 s, stage limits 1.2-3 s
    t0 = time.monotonic()
    rc, made, lines, log = _run(tmp_path, capsys, monkeypatch, MAU_FAKE_GRAPH_FAILURE="hang_always")
    took = time.monotonic() - t0
    assert rc == 1 and not lines and 1 <= len(made) <= 3 and made[0] == (0, True)
    assert took < 5.4 + 3.0, took                                   # budget + the kill / drain slack of the last attempt


def test_short_budget_skips_to_the_most_conservative_attempt(tmp_path, capsys, monkeypatch):
    monkeypatch.delenv("MAU_DP_GRAPH", raising=False)
    monkeypatch.delenv("MAU_RCCL_DIRECT", raising=False)
    make_attempt, made = _attempts(tmp_path, 2, {})
    log = io.StringIO()
    # 150 s of budget, MIN_ATTEMPT_S = 100: not enough for an attempt AND a fallback -> straight to the last plan entry
    assert bench.run_supervised(make_attempt, log=log, budget=150.0) == 0 and made == [(2, False)]
    rec = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][0])
    assert rec["launch"] == "eager" and rec["collectives"] == "pg" and "most conservative" in log.getvalue()
    # less than one attempt's worth: nothing is started
    make_attempt, made = _attempts(tmp_path, 2, {})
    assert bench.run_supervised(make_attempt, log=io.StringIO(), budget=50.0) == 1 and made == []


REAL_RENDEZVOUS_WORKER = textwrap.dedent(r'''
    import json, os, sys
    import torch, torch.distributed as dist
    tag = "[mau-bench-stage]"
    print(tag, "imported", file=sys.stderr, flush=True)
    assert "TORCHELASTIC_USE_AGENT_STORE" not in os.environ
    dist.init_process_group("gloo")                              # env:// -- MASTER_PORT is the attempt's own port
    print(tag, "ready", file=sys.stderr, flush=True)
    t = torch.ones(1) * (dist.get_rank() + 1)
    dist.all_reduce(t)
    if os.environ["MAU_DP_GRAPH"] == "1":
        os._exit(17)                                             # the "capture" fails on every rank, after the group was up
    print(tag, "timed", file=sys.stderr, flush=True)
    if dist.get_rank() == 0:
        print(json.dumps({"metric": "fake", "sum": float(t), "port": os.environ["MASTER_PORT"], "launch": "eager"}), flush=True)
    dist.barrier()
    print(tag, "done", file=sys.stderr, flush=True)
''')


def test_under_torchrun_each_rank_supervises_one_worker_and_attempts_do_not_share_a_store(tmp_path):
    """The driver's N > 1 launch line: ``python -m torch.distributed.run ... bench.py --gpus N``.  Every launched process is a
    supervisor (no torch import, no GPU); the workers rendezvous on a port of the attempt's own -- the launcher's store has no
    per-attempt key space, and a failed attempt's gloo/RCCL addresses left in it made the second attempt fail to connect."""
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(REAL_RENDEZVOUS_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MAU_DP_GRAPH")}
    env["MAU_BENCH_WORKER_SCRIPT"] = str(script)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["sum"] == 3.0 and rec["launch"] == "eager" and rec["port"] != "29547" and "falling back" in p.stderr


ASYMMETRIC_WORKER = textwrap.dedent(r'''
    import json, os, sys, time
    tag = "[mau-bench-stage]"
    rank = int(os.environ["RANK"])
    for st in ("spawned", "imported", "ready", "warm"):
        print(tag, st, file=sys.stderr, flush=True)
    if os.environ["MAU_DP_GRAPH"] == "1":
        if rank == 1:
            os._exit(17)                                         # ONE rank's capture fails at once ...
        time.sleep(600)                                          # ... its peer hangs in the next collective
    print(tag, "captured", file=sys.stderr, flush=True)
    print(tag, "timed", file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps({"metric": "fake", "launch": "eager", "port": os.environ["MASTER_PORT"]}), flush=True)
    print(tag, "done", file=sys.stderr, flush=True)
''')


@pytest.mark.parametrize("nproc", [2, 8])
def test_under_torchrun_an_asymmetric_failure_still_falls_back(tmp_path, nproc):
    """one rank's worker dies at once, its peer hangs.  The early supervisor must not give up waiting for the next
    attempt's port before the leader's supervisor has ended ITS worker -- and the leader must not wait for the stage limit:
    the failed rank's flag ends the attempt everywhere."""
    import glob
    import subprocess
    import tempfile
    import time
    script = tmp_path / "worker.py"
    script.write_text(ASYMMETRIC_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MAU_DP_GRAPH")}
    env["MAU_BENCH_WORKER_SCRIPT"] = str(script)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", "29548",
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)]              # (8 = the driver's launch line for the scaling run)
    t0 = time.monotonic()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    took = time.monotonic() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["launch"] == "eager", p.stdout
    assert "another rank's supervisor reported a failed worker" in p.stderr and "falling back" in p.stderr
    assert took < 100, took                                         # far below the 120 s stage limit of the hung peer
    assert not glob.glob(os.path.join(tempfile.gettempdir(), "mau_bench_*_29548*"))        # the coordination directory is gone
