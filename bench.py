#!/usr/bin/env python3
"""Headline benchmark: train images/s of the Metadata-Augmented U-Net on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is the inner training step of the reference (src/train.py:243-256) on one batch of
synthetic input already resident in HBM: forward, MSE criterion, backward, AdamW step, zero_grad.
The step is captured once into a hipGraph (mau_amd.GraphedTrainStep; --no-graph = launched kernel by
kernel from Python) and the K timed steps are K replays.  N > 1 (one rank per GPU): the captured step
contains the RCCL collectives (SyncBN messages on the compute stream, gradient buckets on a
communication stream).  A capture that misbehaves at N > 1 must not cost the measurement, so every
rank runs under a SUPERVISOR that never touches the GPU: `python bench.py --gpus N` starts the N
workers itself; under torch.distributed.run each launched rank is the supervisor of its own worker.
Attempt 1 runs the workers with MAU_DP_GRAPH=1; a worker that exits non-zero or stops reporting
progress is killed (with its process group) and a FRESH set of workers runs the eager step
(MAU_DP_GRAPH=0) on a fresh rendezvous; should that fail too, a last set runs the eager step with every
collective through ProcessGroupNCCL (MAU_RCCL_DIRECT=0).  Nothing is ever exec'ed or restarted in place.
The timed region (exactly K steps between barrier + synchronize) is repeated R times back to back
(R chosen so the GPU phase lasts >= ~10 s) and the MEDIAN region is reported; the spread is in the line.
Workload (BASELINE.json configs[1] / configs[3]): U-Net, base_filters 64, 6x256x256 tiles + 4-dim
metadata, B = 32 per GPU, bf16 MFMA arithmetic with fp32 accumulation, fp32 master weights.
Prints ONE JSON line on rank 0 (contract in the task description) carrying `roofline` (dominant
kernel: the 3x3 implicit-GEMM convolution, timed live with events on the launch stream) and
`cpu_baseline` (the CPU oracle = port of the reference's torch operators, timed on the host cores).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

torch = None          # imported by main() in the process that does the work: a supervisor stays off torch and off the GPU

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_F32_TFLOPS = 157.3        # fp32 MFMA peak (same guide)


def conv_flops(N, H, W, Cin, Cout):
    return 2.0 * 9.0 * Cin * Cout * N * H * W


class ConvTimer:
    """Brackets every launch of the dominant kernel (conv3x3 implicit GEMM: forward and data-gradient
    launches of mau_conv3x3_fwd) -- and, separately, of the weight-gradient kernel -- with events recorded on
    the stream the kernel is launched on (torch's current stream, which is what the C ABI receives)."""

    def __init__(self, F_):
        self.F_ = F_
        self.records = []
        self.wrecords = []
        self.urecords = []          # mau_conv3x3_unpack_wgrad: the weight gradient's second stage (split-K sum + un-tiling)
        self.enabled = False
        self._orig = F_.call

    def install(self):
        orig = self._orig

        def call(name, *args):
            if self.enabled and name in ("mau_conv3x3_fwd", "mau_conv3x3_fwd2"):
                if name == "mau_conv3x3_fwd":
                    # args: x, ldx, C0, emb, emb_ws, E, wpk, bias, post_scale, post_shift, y, ldy, Cout, slab, dtype, N, H, W, stream
                    C0, E, Cout, N, H, W = args[2], args[5], args[12], args[15], args[16], args[17]
                else:
                    # args: x, ldx, C0, x1, ldx1, C1, emb, emb_ws, E, wpk, bias, post_scale, post_shift, y, ldy, Cout, slab, dtype, N, H, W, stream
                    C0, E, Cout, N, H, W = args[2] + args[5], args[8], args[15], args[18], args[19], args[20]
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                orig(name, *args)
                e1.record()
                self.records.append((e0, e1, conv_flops(N, H, W, C0 + E, Cout)))
            elif self.enabled and name == "mau_conv3x3_wgrad2":
                # args: x, ldx, C0, x1, ldx1, C1, emb, emb_ws, E, dy, lddy, Cout, acc, dtype, N, H, W, stream
                Cin, Cout, N, H, W = args[2] + args[5] + args[8], args[11], args[14], args[15], args[16]
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                # (the weight gradient may be launched on its own stream: the events go where the kernel goes)
                cur = torch.cuda.current_stream()
                ws = cur if args[-1] == cur.cuda_stream else torch.cuda.ExternalStream(args[-1])
                e0.record(ws)
                orig(name, *args)
                e1.record(ws)
                # algorithmic bytes of the launch: X and dY read once (16-bit), dW written once (fp32)
                self.wbytes = getattr(self, "wbytes", 0.0) + (Cin + Cout) * N * H * W * 2.0 + 9.0 * Cin * Cout * 4.0
                self.wrecords.append((e0, e1, conv_flops(N, H, W, Cin, Cout)))
            elif self.enabled and name == "mau_conv3x3_unpack_wgrad":
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                cur = torch.cuda.current_stream()
                ws = cur if args[-1] == cur.cuda_stream else torch.cuda.ExternalStream(args[-1])
                e0.record(ws)
                orig(name, *args)
                e1.record(ws)
                self.urecords.append((e0, e1, 0.0))
            else:
                orig(name, *args)

        self.F_.call = call

    @staticmethod
    def _sum(recs):
        if not recs:
            return None
        ms = [a.elapsed_time(b) for a, b, _ in recs]
        fl = [f for _, _, f in recs]
        return dict(launches=len(ms), total_ms=sum(ms), total_flop=sum(fl))

    def summary(self):
        r = self._sum(self.records)
        if r is not None:      # the launches of >= 0.3 ms alone (what the PMC figures of profiles/ are quoted on)
            long = [(a.elapsed_time(b), f) for a, b, f in self.records if a.elapsed_time(b) >= 0.3]
            r["long_launches"] = len(long)
            r["long_tflops"] = sum(f for _, f in long) / max(1e-9, sum(t for t, _ in long) * 1e-3) / 1e12 if long else None
        return r

    def wgrad_summary(self):
        r = self._sum(self.wrecords)
        if r is not None and self.urecords:
            r["unpack_ms"] = sum(a.elapsed_time(b) for a, b, _ in self.urecords)
        return r


def cpu_baseline(iters=5, warmup=2, b32=False):
    """The oracle (same torch CPU operators as the reference, fp32) on a bounded sample -- B = 2 of the same workload
    (BASELINE config 1) -- as BASELINE.md section 3 / SURVEY 8(d) ask: 2 warm-ups, median of >= 5 iterations,
    train step (fwd + MSE + bwd + AdamW), train-mode forward and eval-mode forward timed separately.
    SURVEY 8(d) also names B = 32: ONE iteration of the full-batch step costs about a minute of host time, so the default run
    does not repeat it -- ``--cpu-baseline-b32`` times that one iteration (after one untimed B = 2 step that has paged the
    operators in), and the default line quotes the committed record of it (profiles/r<N>/cpu_baseline_b32.json, newest round) beside the B = 2
    sample, saying which is which."""
    from oracle import unet_ref as R
    torch.manual_seed(0)
    flags = dict(temporal_embeddings=False, metadata_embeddings=True)
    sd = R.clone_state(R.init_state("unet", 6, 10, 64, 4, 64, 96, 2, **flags), requires_grad=True)
    params = [sd[k] for k in sd if R.is_param(k)]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-3)
    x, ts, md, tgt = R.synthetic_batch(2)

    def timed(fn):
        ts_ = []
        for i in range(warmup + iters):
            t0 = time.perf_counter()
            fn()
            if i >= warmup:
                ts_.append(time.perf_counter() - t0)
        return statistics.median(ts_)

    def fwd(training):
        with torch.no_grad():
            R.forward("unet", sd, x, ts, md, training, **flags)

    t_step = timed(lambda: R.train_step("unet", sd, opt, x, ts, md, tgt, **flags))
    t_fwd = timed(lambda: fwd(True))
    t_eval = timed(lambda: fwd(False))
    rec = {"value": round(2.0 / t_step, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "fwd_train_images_s": round(2.0 / t_fwd, 4), "fwd_eval_images_s": round(2.0 / t_eval, 4),
           "host_logical_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(),
           "sample": f"B=2 of the same workload (U-Net base 64, 6x256x256, fp32): median of {iters} iterations after {warmup} warm-ups; "
                     f"train step (fwd+MSE+bwd+AdamW) {t_step:.2f} s, train-mode forward {t_fwd:.2f} s, eval forward {t_eval:.2f} s"}
    if b32:
        x, ts, md, tgt = R.synthetic_batch(32)
        t0 = time.perf_counter()
        R.train_step("unet", sd, opt, x, ts, md, tgt, **flags)
        t32 = time.perf_counter() - t0
        rec["b32_one_iteration"] = {"value": round(32.0 / t32, 4), "unit": "images/s", "seconds_per_step": round(t32, 2), "cores": torch.get_num_threads(),
                                    "sample": "ONE iteration of the B=32 train step (the bench's own batch), after the B=2 iterations above"}
    else:
        try:
            with open(newest_record("cpu_baseline_b32.json")) as f:
                rec["b32_recorded"] = dict(json.load(f), note="NOT of this run: the committed one-iteration B=32 record (python bench.py --cpu-baseline-b32); "
                                                              "this run timed the B=2 sample only")
        except (OSError, ValueError, TypeError):
            pass
    return rec


def record_rounds():
    """profiles/r<N> directories, newest round first (the committed records a line may quote -- always labelled as such)."""
    try:
        ds = [d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()]
    except OSError:
        return []
    return sorted(ds, key=lambda d: -int(d[1:]))


def newest_record(name):
    """path of the newest round's profiles/r<N>/<name> that exists (None if no round has one)"""
    for rnd in record_rounds():
        p = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(p):
            return p
    return None


STAGE_TAG = "[mau-bench-stage]"
# seconds without a NEW stage line before a worker counts as hung (stage it is in -> limit).  Generous where a slow host decides
# (first `import torch` on a fresh box: 1-2 min; rendezvous waits for the slowest rank), tight where a broken capture or a
# mis-ordered collective would hang: `captured` = the timed regions (~10-25 s of GPU work), `timed` = the per-kernel event pass
# and the forward latency after them (seconds).  Every limit is additionally clipped to what is left of the run's budget.
STAGE_LIMITS = {"spawned": 300, "imported": 300, "ready": 150, "warm": 120, "captured": 120, "timed": 120, "done": 60}
# One budget for the whole N > 1 measurement, all attempts together (the driver ends the run at 600 s): an attempt is only
# started with enough of it left for itself (MIN_ATTEMPT_S: fresh workers, import, warm-up, ~10 s of timed regions), and a
# non-final attempt is cut off early enough to leave that much for the most conservative one.
BUDGET_S = float(os.environ.get("MAU_BENCH_BUDGET_S", "540"))
MIN_ATTEMPT_S = float(os.environ.get("MAU_BENCH_MIN_ATTEMPT_S", "100"))


def stage(name):
    """Progress line of a worker for its supervisor (stderr, so stdout stays the one JSON line)."""
    if os.environ.get("MAU_BENCH_WORKER") == "1":
        print(f"{STAGE_TAG} {name}", file=sys.stderr, flush=True)


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _metric_lines(out):
    return [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]


def supervise_workers(cmds_envs, limits=None, poll=0.2, log=sys.stderr, deadline=None, abort=None):
    """Run one attempt: start every (cmd, env) as a child in its own session, relay their stderr, watch their stage lines.
    Returns (ok, stdout of worker 0).  ok = every worker reported `done` (after which its exit code no longer matters:
    the result line is out and the ranks have passed their last barrier) or exited 0.  On the first failure -- a non-zero
    exit before `done`, no new stage line within the stage's limit, the attempt's ``deadline`` (time.monotonic()) passing,
    or ``abort()`` turning true (another rank's supervisor saw ITS worker fail) -- ALL workers of the attempt are killed
    (SIGKILL to their sessions: a rank hung in a collective ignores anything gentler)."""
    import signal
    import subprocess
    import threading
    limits = dict(STAGE_LIMITS, **(limits or {}))
    scale = float(os.environ.get("MAU_BENCH_STALL_SCALE", "1"))
    procs, state = [], []

    def pump(i, pipe):
        for line in pipe:
            if line.startswith(STAGE_TAG):
                state[i]["stage"] = line[len(STAGE_TAG):].strip()
                state[i]["t"] = time.monotonic()
            else:
                log.write(line)
                log.flush()

    for i, (cmd, env) in enumerate(cmds_envs):
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        procs.append(p)
        state.append({"stage": "spawned", "t": time.monotonic(), "out": []})
        threading.Thread(target=pump, args=(i, p.stderr), daemon=True).start()
        threading.Thread(target=lambda i=i, p=p: state[i]["out"].extend(p.stdout), daemon=True).start()
    failure = None
    while failure is None:
        alive = False
        now = time.monotonic()
        for i, p in enumerate(procs):
            rc = p.poll()
            st = state[i]
            if rc is None:
                alive = True
                if now - st["t"] > limits.get(st["stage"], 240) * scale:
                    if st["stage"] == "done":             # result out, last barrier passed: a rank stuck in its teardown is just ended
                        try:
                            os.killpg(p.pid, signal.SIGKILL)
                        except (ProcessLookupError, PermissionError):
                            pass
                        continue
                    failure = f"worker {i} made no progress for {limits.get(st['stage'], 240) * scale:.0f} s in stage '{st['stage']}'"
            elif rc != 0 and st["stage"] != "done":
                failure = f"worker {i} exited with code {rc} in stage '{st['stage']}'"
        if not alive:
            break
        pending = [i for i, p in enumerate(procs) if p.poll() is None and state[i]["stage"] != "done"]
        if failure is None and pending and deadline is not None and now > deadline:
            failure = f"the attempt ran out of its share of the {BUDGET_S:.0f} s budget (worker {pending[0]} in stage '{state[pending[0]]['stage']}')"
        if failure is None and pending and abort is not None and abort():
            failure = "another rank's supervisor reported a failed worker"
        time.sleep(poll)
    if failure is not None:
        print(f"bench.py supervisor: {failure}; killing this attempt's workers", file=log, flush=True)
    for p in procs:                                   # (also after success: nothing of the attempt may linger)
        if p.poll() is None and failure is not None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
    for p in procs:
        try:
            p.wait(timeout=30)
        except Exception:
            pass
    time.sleep(0.2)                                   # let the stdout reader threads drain
    supervise_workers.last_failure = failure          # (what ended the attempt, for the forwarded line: run_supervised)
    return failure is None, "".join(state[0]["out"]) if state else ""


class NodeCoordinator:
    """How the rank supervisors of ONE node (the processes torch.distributed.run started) stay in step without a collective
    and without the launcher's store (it has no per-attempt key space: the addresses a failed attempt leaves there poison
    the next one).  A directory named after the launcher (the parent of every rank it started: pid + start time) holds
      <seq>.plan    written by the LEADER (local rank 0): "<plan index> <port>" for attempt number seq, or "stop <rc>";
      <seq>.failed  touched by ANY supervisor whose worker failed in attempt seq: the others end theirs at once instead of
                    waiting for the stage limit of a worker hung in a collective with a dead peer;
      ack.<rank>    a follower has read the leader's "stop".
    Only the leader decides which attempt runs next (its clock, its budget); followers wait for the next plan for as long as
    the run's budget lasts (an early failure on one rank never times out against a peer that is still waiting for its own
    worker's stage limit).  The leader removes the directory when it leaves; a follower that finds it gone stops too."""

    def __init__(self):
        import tempfile
        ppid = os.getppid()
        try:                      # the launcher's start time (clock ticks since boot) tells two launchers apart that got the same pid
            with open(f"/proc/{ppid}/stat") as f:
                born = f.read().rsplit(")", 1)[1].split()[19]
        except (OSError, IndexError):
            born = "0"
        self.dir = os.path.join(tempfile.gettempdir(), f"mau_bench_{ppid}_{born}_{os.environ.get('MASTER_PORT', '0')}")
        self.local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        self.n_local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        self.leader = self.local_rank == 0
        self.seen = False
        if self.leader:
            os.makedirs(self.dir, exist_ok=True)
            for f in os.listdir(self.dir):            # (a launcher pid + start time + port cannot repeat; belt and braces)
                self._unlink(f)

    def _unlink(self, name):
        try:
            os.unlink(os.path.join(self.dir, name))
        except OSError:
            pass

    def _write(self, name, text):
        try:
            tmp = os.path.join(self.dir, f".{name}.{os.getpid()}.tmp")
            with open(tmp, "w") as f:
                f.write(text)
            os.replace(tmp, os.path.join(self.dir, name))
        except OSError:
            pass                                       # (a follower writing into a directory the leader has just removed)

    def announce(self, seq, k, port):
        self._write(f"{seq}.plan", f"{k} {port}")

    def stop(self, seq, rc, wait=5.0):
        """Leader: tell the followers the run is over, give them a moment to read it, remove the directory."""
        self._write(f"{seq}.plan", f"stop {rc}")
        t0 = time.monotonic()
        while time.monotonic() - t0 < wait:
            try:
                acks = sum(1 for f in os.listdir(self.dir) if f.startswith("ack."))
            except OSError:
                break
            if acks >= self.n_local - 1:
                break
            time.sleep(0.05)
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)

    def await_plan(self, seq, deadline):
        """Follower: ("run", k, port) | ("stop", rc).  The directory vanishing after it was seen = the leader has left."""
        path = os.path.join(self.dir, f"{seq}.plan")
        while time.monotonic() < deadline:
            try:
                with open(path) as f:
                    a, b = f.read().split()
                if a == "stop":
                    self._write(f"ack.{self.local_rank}", "")
                    return ("stop", int(b))
                return ("run", int(a), int(b))
            except (OSError, ValueError):
                if os.path.isdir(self.dir):
                    self.seen = True
                elif self.seen:
                    return ("stop", 0)
                time.sleep(0.05)
        return ("stop", 1)

    def flag_failure(self, seq):
        self._write(f"{seq}.failed", "")

    def failed(self, seq):
        return os.path.exists(os.path.join(self.dir, f"{seq}.failed"))


def run_supervised(make_attempt, log=sys.stderr, coord=None, budget=None):
    """The N > 1 measurement under ONE time budget (``MAU_BENCH_BUDGET_S``, default 540 s; the driver's limit is 600).
    Attempt 1: captured data-parallel step (MAU_DP_GRAPH=1) unless the environment pins the mode; attempt 2 (only after a
    failed attempt 1): eager step, fresh workers, fresh rendezvous; attempt 3 (only after a failed attempt 2, and unless the
    environment pins MAU_RCCL_DIRECT): the eager step with every collective through ProcessGroupNCCL (MAU_RCCL_DIRECT=0, round
    3's path: slower, but the one torch itself exercises everywhere).  ``make_attempt(k, dp_graph, port)`` -> list of (cmd, env).
    Budget rules: an attempt starts only with MIN_ATTEMPT_S of the budget left; with less than two of them left the plan jumps
    to its LAST entry (the most conservative attempt); a non-final attempt is ended MIN_ATTEMPT_S before the budget's end.
    A result line that worker 0 has already printed is never lost: rank 0 prints one as soon as the timed regions are over
    (before the per-kernel event pass) and again, complete, at the end -- the last line in hand is printed even when the
    attempt failed afterwards.  Prints that ONE JSON line -- its ``config.launch`` / ``config.collectives`` say which attempt
    produced it; exit code 0 iff a line was printed (on the rank that holds worker 0) or the leader said so (``coord``)."""
    t_start = time.monotonic()
    scale = float(os.environ.get("MAU_BENCH_STALL_SCALE", "1"))
    budget = (BUDGET_S if budget is None else budget) * scale
    min_attempt = MIN_ATTEMPT_S * scale
    deadline = t_start + budget
    pinned = os.environ.get("MAU_DP_GRAPH")
    plan = [(pinned == "1", {})] if pinned in ("0", "1") else [(True, {}), (False, {})]
    if os.environ.get("MAU_RCCL_DIRECT") is None and not plan[-1][0]:
        plan.append((False, {"MAU_RCCL_DIRECT": "0"}))
    what = {True: "captured data-parallel step", False: "eager data-parallel step"}
    leader = coord is None or coord.leader
    k, seq, rc = 0, 0, 1
    while True:
        if leader:
            left = deadline - time.monotonic()
            if k >= len(plan) or left < min_attempt:
                if k < len(plan):
                    print(f"bench.py supervisor: {left:.0f} s of the {budget:.0f} s budget left: no further attempt", file=log, flush=True)
                break
            if k < len(plan) - 1 and left < 2 * min_attempt:
                print(f"bench.py supervisor: {left:.0f} s of the budget left: skipping to the most conservative attempt", file=log, flush=True)
                k = len(plan) - 1
            port = _free_port()
            if coord is not None:
                coord.announce(seq, k, port)
        else:
            got = coord.await_plan(seq, deadline + 30.0 * scale)
            if got[0] == "stop":
                return got[1]
            _, k, port = got
        dp_graph, extra = plan[k]
        if seq > 0:
            print(f"bench.py supervisor: falling back to the {what[dp_graph]}" + (" over ProcessGroupNCCL" if extra else "") + " with fresh workers",
                  file=log, flush=True)
        attempt = [(cmd, dict(env, **extra)) for cmd, env in make_attempt(k, dp_graph, port)]
        last = k == len(plan) - 1
        ok, out = supervise_workers(attempt, log=log, deadline=deadline - (0.0 if last else min_attempt),
                                    abort=(lambda s=seq: coord.failed(s)) if coord is not None else None)
        if not ok and coord is not None:
            coord.flag_failure(seq)
        lines = _metric_lines(out)
        for ln in out.splitlines():
            if ln not in lines:
                print(ln, file=log)
        if lines:                                     # (only the supervisor of worker 0 ever holds one)
            line = lines[-1]
            if not ok:
                # the value stands (it was measured and printed before the failure), but the line must SAY that something behind it
                # failed -- a fault or hang in the per-kernel event pass, the forward-latency pass or the teardown is not a success
                print("bench.py supervisor: the attempt failed AFTER its timed regions; the result line it had printed stands", file=log, flush=True)
                try:
                    rec = json.loads(line)
                    rec["attempt_failed_after_timed_regions"] = True
                    rec["failure"] = str(getattr(supervise_workers, "last_failure", None))
                    rl = rec.get("roofline")
                    if isinstance(rl, dict):
                        rl["source"] = "recorded" if str(rl.get("timing", "")).startswith("NOT of this run") else "live"
                    line = json.dumps(rec)
                except ValueError:
                    pass
            print(line, flush=True)
            rc = 0
            break
        if ok and leader:                             # (no worker 0 here, or a stand-in that prints nothing)
            rc = 0
            break
        if ok and not leader:                         # this rank is through; the leader says how the run ended
            seq += 1
            got = coord.await_plan(seq, deadline + 30.0 * scale)
            return got[1] if got[0] == "stop" else 1
        k, seq = k + 1, seq + 1
    if coord is not None and leader:
        coord.stop(seq + 1 if rc == 0 else seq, rc)
    return rc


def self_launch(args, argv):
    """``python bench.py --gpus N`` (N > 1) outside torch.distributed.run: this process -- it has not touched the GPU and never
    execs -- is the supervisor of all N ranks."""
    def make_attempt(k, dp_graph, port):
        out = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), MAU_BENCH_WORKER="1", MAU_DP_GRAPH="1" if dp_graph else "0")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
            out.append(([sys.executable, os.path.abspath(__file__)] + argv, env))
        return out

    raise SystemExit(run_supervised(make_attempt))


def rank_supervisor(argv):
    """Under torch.distributed.run (RANK / WORLD_SIZE set by the launcher): this launched process stays off the GPU and supervises
    ONE worker -- its rank.  The supervisor of local rank 0 leads (``NodeCoordinator``): it picks each attempt and its
    rendezvous port (rank 0 of the workers hosts a store there -- a port of the attempt's own), the others follow; a failed
    worker on ANY rank ends the attempt on every rank at once.  One node only: the coordination is through the node's /tmp."""
    if int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])) != int(os.environ["WORLD_SIZE"]):
        raise SystemExit("bench.py: the supervised N > 1 run is for ONE node (--nnodes=1): its rank supervisors agree through the node's /tmp")
    coord = NodeCoordinator()

    def make_attempt(k, dp_graph, port):
        env = dict(os.environ, MAU_BENCH_WORKER="1", MAU_DP_GRAPH="1" if dp_graph else "0", MASTER_PORT=str(port))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        # (MAU_BENCH_WORKER_SCRIPT: the CPU tests put a stand-in for the GPU worker here)
        return [([sys.executable, os.environ.get("MAU_BENCH_WORKER_SCRIPT") or os.path.abspath(__file__)] + argv, env)]

    raise SystemExit(run_supervised(make_attempt, coord=coord))


def workload_key(args):
    mode = "infer" if args.infer else "train"
    extra = "" if args.seq_len == 10 and not args.temporal_embeddings else f"_T{args.seq_len}{'_temb' if args.temporal_embeddings else ''}"
    return f"{args.model_type}_{args.precision}_b{args.batch}_s{args.size}_c{args.channels}_{mode}{extra}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE: 32)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--model-type", default="unet", choices=["unet", "unet++"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-b32", action="store_true", help="also time ONE iteration of the oracle's B=32 train step on the host (about a minute)")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of mau_amd.AdamW (same update rule)")
    ap.add_argument("--no-graph", action="store_true", help="launch the train step kernel by kernel from Python instead of replaying "
                                                            "its hipGraph (always so for N > 1 and for --infer)")
    ap.add_argument("--infer", action="store_true", help="inference-only images/s (eval mode, no_grad): BASELINE configs[4]")
    ap.add_argument("--channels", type=int, default=6, help="input channels (6 = BASELINE configs, 23 = the app's real shape)")
    ap.add_argument("--meta", type=int, default=4)
    ap.add_argument("--no-sync-bn", action="store_true", help="per-GPU BatchNorm statistics (reference semantics per device)")
    ap.add_argument("--seq-len", type=int, default=10, help="length of the temperature series (reference data: 828, conf/config.yaml:20)")
    ap.add_argument("--temporal-embeddings", action="store_true", help="U-Net with the LSTM TemporalEncoder on the path (always on for unet++)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal on ONE GPU: take the data-parallel code path (SyncBN all-reduces, "
                                                              "bucketed gradient all-reduce from autograd hooks, eager launches) under a 1-rank RCCL group")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass; default: the "
                         "figure committed under profiles/ for the same workload (profiles/r2/pmc_summary.json)")
    ap.add_argument("--repeats", type=int, default=0, help="how many times the K-step timed region is run back to back (median reported); "
                                                           "0 = as many as make the GPU phase last about 10 s")
    args = ap.parse_args()
    if os.environ.get("MAU_BENCH_WORKER") != "1":
        # supervisors: before anything touches the GPU (or even imports torch)
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            self_launch(args, sys.argv[1:])
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            rank_supervisor(sys.argv[1:])
    stage("spawned")
    global torch
    import torch
    stage("imported")
    # the CPU baseline runs FIRST (rank 0, N = 1): the GPU phase that follows is then one contiguous stretch of the run
    cpu_rec = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu_rec = cpu_baseline(b32=args.cpu_baseline_b32)

    os.environ.setdefault("MAU_QUIET", "1")        # stdout carries the one JSON line: the constructor's announcement stays off it
    import mau_amd
    from mau_amd import functional as F_
    from mau_amd.dist import GradSync, init_process_group_from_env
    import torch.distributed as dist

    rank, local, world = init_process_group_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started {world} rank(s)")
    rccl_ranks = dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- model / synthetic data --------------------------------------------------
    torch.manual_seed(0)                       # identical replicas on every rank
    flags = {} if args.model_type == "unet++" else dict(temporal_embeddings=args.temporal_embeddings, metadata_embeddings=True)
    net = mau_amd.UrbanPredictor(args.model_type, args.channels, args.seq_len, 64, args.meta, 64, 96, 2, base_filters=64, **flags)
    net = net.to(dev).set_precision(args.precision).train()
    g = torch.Generator().manual_seed(1234 + rank)
    B, S = args.batch, args.size
    x = torch.randn(B, args.channels, S, S, generator=g).to(dev)
    ts = torch.randn(B, args.seq_len, generator=g).to(dev)
    md = torch.randn(B, args.meta, generator=g).to(dev)
    tgt = torch.randn(B, 2, S, S, generator=g).to(dev)
    # AdamW lr 1e-4 wd 1e-3 (conf/config.yaml:41,48,52): mau_amd.AdamW = torch.optim.AdamW's update with the re-pack of the convolution
    # weights fused into the same kernel (--torch-adamw: torch's fused AdamW + a separate multi-tensor pack launch)
    opt = (torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-3, fused=True) if args.torch_adamw
           else mau_amd.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-3))
    sync = None
    if args.force_dist and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1 or args.force_dist:
        if not args.no_sync_bn:
            net.set_sync_bn(dist.group.WORLD)
        sync = GradSync(net, dist.group.WORLD)
    losses = torch.zeros(args.steps + args.warmup, device=dev)

    session = [None]            # --infer on one GPU: the app's path, an inference.GraphedInference session (one hipGraph replay per call)

    def infer_step(i):
        if session[0] is not None:
            session[0](x, ts, md)
            return
        with torch.no_grad():
            net(x, ts, md)

    graphed = None
    # data parallel: MAU_DP_GRAPH=1 captures the step with its collectives (the supervisor's first attempt); 0 = eager, the
    # collectives launched from autograd hooks (the supervisor's fallback, and the default of an unsupervised worker)
    dp_graph = sync is not None and os.environ.get("MAU_DP_GRAPH", "0") == "1"
    if not args.infer and not args.no_graph and ((world == 1 and sync is None) or dp_graph):
        criterion = lambda o, t: mau_amd.compute_loss_mse(o, t)          # noqa: E731  (src/train.py:218-219)
        graphed = mau_amd.GraphedTrainStep(net, opt, criterion, warmup=min(3, max(1, args.warmup)), copy_inputs=False, grad_sync=sync)

    def step(i):
        if args.infer:
            return infer_step(i)
        i %= losses.numel()
        if graphed is not None:                       # fwd + MSE + bwd + AdamW step: eager while warming up, then one graph replay
            losses[i] = graphed(x, ts, md, tgt)
            return
        out = net(x, ts, md)
        loss = mau_amd.compute_loss_mse(out, tgt)["total"]
        if sync is not None:
            sync.begin()
        loss.backward()
        if sync is not None:
            sync.finish()
        opt.step()
        opt.zero_grad()
        losses[i] = loss.detach()

    if args.infer:
        net.eval().freeze_inference()      # inference session: packed weights / folded BN coefficients computed once
        if not args.no_graph:          # (N > 1: every rank is an independent replica with a session of its own, no collective)
            session[0] = mau_amd.GraphedInference(net, x, ts, md)
    timer = ConvTimer(F_)
    timer.install()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stage("ready")
    try:
        for i in range(args.warmup):
            step(i)
            if graphed is not None and i + 1 == graphed.warmup:
                torch.cuda.synchronize()
                stage("warm")                                 # the next call captures
        if graphed is not None and graphed.graph is None:     # fewer warm-up steps than the capture needs: capture before the clock starts
            for i in range(graphed.warmup + 1 - args.warmup):
                step(0)
    except RuntimeError as e:
        # N > 1: no in-rank recovery -- the worker fails, its supervisor starts a fresh eager set (a child spawned from here would
        # inherit a rendezvous the other ranks still hold)
        if world > 1 and graphed is not None:
            # leave at once, skipping every teardown (a process group whose peers are mid-collective, a half-built graph): the
            # supervisor sees the exit code and starts the eager set
            print(f"bench.py worker (rank {rank}): {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            os._exit(17)
        if graphed is None or "capturing the train step failed" not in str(e):
            raise
        # A broken capture leaves this process's stream state untrustworthy: run the eager bench in a CHILD process (never an
        # exec: this process has initialised the GPU), relay its line and exit with its code.
        import subprocess
        print(f"bench.py: {e}; re-running eagerly in a child process", file=sys.stderr)
        proc = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--no-graph"], stdout=subprocess.PIPE, text=True)
        sys.stdout.write(proc.stdout)
        raise SystemExit(proc.returncode)
    barrier()
    stage("captured")
    # per-kernel events inside the timed region only when every kernel has the GPU to itself there: not under a graph replay (no
    # events in a graph) and not with the weight gradients on their own stream (two kernels share the chip: neither's events
    # measure it alone) -- then the same K steps are launched again afterwards, eagerly, on one stream, with the brackets
    separate_pass = session[0] is not None if args.infer else (graphed is not None or bool(F_._OVERLAP_WGRAD))
    timer.enabled = not separate_pass

    def max_over_ranks(vals):
        if world == 1:
            return list(vals)
        tt = torch.tensor(list(vals), dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.tolist()

    def timed_region():
        """EXACTLY K steps between barrier + synchronize on both sides (the contract's timed region)."""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        barrier()
        return time.perf_counter() - t0

    instrumented_region = None
    if timer.enabled:                # per-launch events inside a region make it slower: that region is measured, but not reported
        instrumented_region = timed_region()
        timer.enabled = False
    regions = [timed_region()]
    repeats = args.repeats
    if repeats <= 0:      # ~10 s of contiguous GPU work, the same count on every rank (derived from the max-over-ranks first region)
        repeats = max(1, min(100, int(10.0 / max(1e-3, max_over_ranks(regions)[0]) + 0.5)))
    for _ in range(repeats - 1):
        regions.append(timed_region())
    regions = max_over_ranks(regions)                        # per region: the slowest rank's clock
    elapsed = statistics.median(regions)
    # the reference loop reads the loss back every step (src/train.py:258, `.detach().cpu().item()`): the same K steps with that
    # read-back (one host synchronisation per step) -- reported beside the headline figure, not instead of it
    readback_ms = None
    if not args.infer:
        barrier()
        t0 = time.perf_counter()
        acc = 0.0
        for i in range(args.steps):
            step(args.warmup + i)
            acc += float(losses[(args.warmup + i) % losses.numel()])
        barrier()
        readback_ms = max_over_ranks([time.perf_counter() - t0])[0] / args.steps * 1e3
    stage("timed")

    def finish():
        if world > 1:
            dist.barrier()                 # every rank is through its work: from here on an exit code cannot cost the measurement
        stage("done")
        if world > 1:
            from mau_amd.dist import destroy_comms
            destroy_comms()                # the directly-driven RCCL communicators go before the group they were built over
            dist.destroy_process_group()

    wkey = workload_key(args)
    peak = PEAK_F32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS      # fp16 and bf16 MFMA run at the same rate

    def pmc_record():
        """PMC counters cannot be read inside this process: HBM traffic, the MFMA-busy fraction and the clock of the dominant
        kernel come from the separate rocprofv3 --pmc passes of THIS workload committed under profiles/ (scripts/profile.sh +
        scripts/summarize_profile.py).  The counters describe the LIBRARY they were measured on: the summary carries the sha256
        of libmau_hip.so, and figures of another binary are not quoted (traffic: null + the reason) -- a kernel change without a
        re-profile cannot go stale silently."""
        if args.traffic_bytes is not None:
            return args.traffic_bytes, "--traffic-bytes", {}
        import hashlib
        from mau_amd import _lib
        with open(_lib.LIB_PATH, "rb") as f:
            lib_sha = hashlib.sha256(f.read()).hexdigest()
        src = "no PMC record of this workload under profiles/"
        for rnd in record_rounds():
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")) as f:
                    cand = json.load(f)[wkey]
            except (OSError, KeyError, ValueError):
                continue
            if cand.get("lib_sha256") != lib_sha:
                src = (f"profiles/{rnd}/pmc_summary.json[{wkey}] was measured on another build of libmau_hip.so "
                       f"(sha256 {str(cand.get('lib_sha256'))[:12]} vs loaded {lib_sha[:12]}): not quoted; re-run scripts/profile.sh")
                break
            return cand["hbm_bytes_per_launch"], (f"profiles/{rnd}/pmc_summary.json[{wkey}] (same libmau_hip.so, sha256 {lib_sha[:12]}): "
                                                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate rocprofv3 --pmc passes"), cand
        return None, src, {}

    def live_roofline(conv, wg, timing_pass):
        traffic, traffic_src, rec = pmc_record()
        achieved = conv["total_flop"] / (conv["total_ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "conv3x3_bf16_kernel (forward + data-gradient launches)",
                    "launches": conv["launches"], "avg_launch_ms": round(conv["total_ms"] / conv["launches"], 4),
                    "flop_per_launch_avg": conv["total_flop"] / conv["launches"],
                    "share_of_step_time": round(conv["total_ms"] / args.steps / (elapsed / args.steps * 1e3), 4),
                    "timing": timing_pass,
                    # PMC figures of the >= 0.3 ms dispatches only (GRBM_GUI_ACTIVE reads high on shorter ones): frac ~ busy x clock / 2.4 GHz
                    "mfma_busy": rec.get("mfma_busy_long"), "clock_ghz": rec.get("clock_ghz_long"),
                    "frac_executed_long_dispatches_pmc": rec.get("frac_long"),
                    "frac_long_launches_live": round(conv["long_tflops"] / peak, 4) if conv.get("long_tflops") else None,
                    "hbm_gbps": round(traffic / (conv["total_ms"] * 1e-3 / conv["launches"]) / 1e9, 1) if traffic else None}
        if wg is not None:
            wach = wg["total_flop"] / (wg["total_ms"] * 1e-3) / 1e12
            wrec = rec.get("wgrad", {})
            wflop = wg["total_flop"] / wg["launches"]
            roofline["wgrad"] = {"kernel": "wgrad16_kernel (+ wgrad_bf16_kernel on small images)", "achieved": round(wach, 2), "frac": round(wach / peak, 4),
                                 "launches": wg["launches"], "avg_launch_ms": round(wg["total_ms"] / wg["launches"], 4),
                                 "traffic": wrec.get("hbm_bytes_per_launch_pmc"), "algorithmic_bytes": round(timer.wbytes / wg["launches"]),
                                 "traffic_ratio": round(wrec["hbm_bytes_per_launch_pmc"] / (timer.wbytes / wg["launches"]), 3) if wrec.get("hbm_bytes_per_launch_pmc") else None,
                                 "mfma_busy": wrec.get("mfma_busy_long"), "clock_ghz": wrec.get("clock_ghz_long"), "flop_per_launch_avg": wflop,
                                 # the second stage (mau_conv3x3_unpack_wgrad: fixed-order split-K sum + un-tiling into the gradient arena) counted in
                                 "unpack_ms_per_step": round(wg.get("unpack_ms", 0.0) / args.steps, 4),
                                 "frac_with_unpack": round(wg["total_flop"] / ((wg["total_ms"] + wg.get("unpack_ms", 0.0)) * 1e-3) / 1e12 / peak, 4)}
        return roofline

    def recorded_roofline():
        """N > 1, before this run's own per-kernel event pass has completed: the dominant kernel's figures of the committed
        one-GPU record of the same workload (the kernel does not know how many ranks there are) -- replaced by the live
        figures in the final line; this one only survives if the event pass hangs."""
        for rnd in record_rounds()[:2]:
            for name in ("bench_default.json", f"bench_{wkey}.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", rnd, name)) as f:
                        rec = json.load(f)
                except (OSError, ValueError):
                    continue
                rl = rec.get("roofline")
                if rl and rec.get("dtype") == args.precision and str(rec.get("config", {}).get("workload", "")).startswith(f"metadata-{args.model_type} "):
                    rl = dict(rl)
                    rl["timing"] = (f"NOT of this run: the one-GPU record profiles/{rnd}/{name} (this run's own per-kernel event pass follows the "
                                    "timed regions and had not completed when this line was printed)")
                    return rl
        return {"bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "timing": "this run's per-kernel event pass had not completed when this line was printed, and no one-GPU record of the workload is committed"}

    def result_line(roofline, fwd):
        result = {
            "metric": ("inference images/sec" if args.infer else "train images/sec") + f" ({S}x{S}x{args.channels}->2 {'U-Net' if args.model_type == 'unet' else 'U-Net++'}, B={B}/GPU)",
            "value": round(B * world * args.steps / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "timed_regions": {"repeats": len(regions), "reported": "value / ms_per_step = the MEDIAN region (each region: exactly K steps between barrier + synchronize, max over ranks)",
                              "ms_per_step_min": round(min(regions) / args.steps * 1e3, 3), "ms_per_step_max": round(max(regions) / args.steps * 1e3, 3),
                              "ms_per_step_first": round(regions[0] / args.steps * 1e3, 3),
                              "instrumented_region_excluded": bool(instrumented_region)},
            "ms_per_step_with_loss_readback": None if readback_ms is None else round(readback_ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": (f"metadata-{args.model_type} base_filters=64, {B}x{args.channels}x{S}x{S} tiles + {args.meta}-dim metadata per GPU, "
                                    + (f"temperature series of {args.seq_len} months, " if (args.seq_len != 10 or args.temporal_embeddings) else "")
                                    + ("eval-mode forward only (inference)" if args.infer else "fwd+MSE+bwd+AdamW (src/train.py:243-256)")),
                       "global_batch": B * world, "parallelism": f"dp{world}" + (" (data-parallel path forced under a 1-rank RCCL group)" if args.force_dist and world == 1 else ""),
                       "sync_bn": bool(sync is not None and not args.no_sync_bn),
                       "collectives": (None if sync is None else "RCCL called directly (ncclAllReduce on the compute / communication stream)"
                                       if sync.comm is not None else f"torch.distributed ({dist.get_backend()})"),
                       "launch": ("hipGraph replay of the captured step" if graphed is not None else
                                  "hipGraph replay (GraphedInference session: input copies + replay + output clone per call)" if session[0] is not None
                                  else "eager (kernel by kernel)")},
            "rccl_ranks": rccl_ranks,
            "final_loss": float(losses[-1]),
            "roofline": roofline,
        }
        result.update(fwd)
        if cpu_rec is not None:
            result["cpu_baseline"] = cpu_rec
        return json.dumps(result)

    if world > 1 and rank == 0:
        # The measurement is in hand: print it NOW.  What follows (the event pass runs the collectives again) can only lose it;
        # the supervisor forwards the LAST line this worker printed, so the complete line below replaces this one.
        print(result_line(recorded_roofline(), {"fwd_ms_per_tile": None}), flush=True)

    timing_pass = "HIP events on the launch stream around every launch, inside an instrumented region of its own (not among the reported ones)"
    if separate_pass:
        # the same K steps launched eagerly (kernel by kernel, same resident batch, same kernels, ONE stream) right after the timed
        # region, with the event brackets, for the per-kernel durations
        opt.zero_grad(set_to_none=True)
        timer.enabled = True
        overlap, F_._OVERLAP_WGRAD = F_._OVERLAP_WGRAD, 0          # one stream: a kernel's events bracket that kernel alone
        for i in range(args.steps):
            if args.infer:
                with torch.no_grad():
                    net(x, ts, md)
                continue
            out = net(x, ts, md)
            loss = mau_amd.compute_loss_mse(out, tgt)["total"]
            if sync is not None:
                sync.begin()
            loss.backward()
            if sync is not None:
                sync.finish()
            opt.step()
            opt.zero_grad()
        torch.cuda.synchronize()
        timer.enabled = False
        F_._OVERLAP_WGRAD = overlap
        timing_pass = (f"HIP events on the launch stream around every launch over {args.steps} eager steps of the same workload run "
                       "on one stream right after the timed region (the timed steps are hipGraph replays / run the weight gradients on a second stream)")

    # ---- forward latency per tile (eval mode, no_grad), outside the timed region: what the app calls per click
    # (app/model_utils.py:102-109) = a frozen inference session replayed as one hipGraph; the eager un-frozen forward beside it ----
    def per_tile(fn, nf=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nf):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / nf / B * 1e3

    net.eval()
    with torch.no_grad():
        fwd_eager = per_tile(lambda: net(x, ts, md))
    fwd = {"fwd_ms_per_tile": round(fwd_eager, 4), "fwd_ms_per_tile_eager": round(fwd_eager, 4), "fwd_eager_frozen": bool(args.infer), "fwd_ms_per_tile_path": "eager eval forward (no inference session)"}
    if not args.infer and world == 1:
        sess = mau_amd.GraphedInference(net, x, ts, md)             # (freezes the model; net.train() below unfreezes it)
        fwd["fwd_ms_per_tile"] = round(per_tile(lambda: sess(x, ts, md)), 4)
        fwd["fwd_ms_per_tile_path"] = f"freeze_inference + GraphedInference replay at B={B} (the product's inference path)"
        del sess
    elif session[0] is not None:
        fwd["fwd_ms_per_tile"] = round(per_tile(lambda: session[0](x, ts, md)), 4)
        fwd["fwd_ms_per_tile_path"] = f"freeze_inference + GraphedInference replay at B={B} (the product's inference path)"
    net.train()

    if rank != 0:
        return finish()

    print(result_line(live_roofline(timer.summary(), timer.wgrad_summary(), timing_pass), fwd), flush=True)
    finish()


if __name__ == "__main__":
    main()
