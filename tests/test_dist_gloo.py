"""Data-parallel path on CPU: world_size 2, gloo backend (the same code runs over RCCL on the GPUs).

 * GradSync: bucketed, hook-driven gradient averaging == single-process gradients on the joint batch,
   including a parameter that never receives a gradient.
 * SyncBN exchange: the [sum | sum of squares | local pixel count] all-reduce of functional.ConvBNReLU (the message layout of
   mau_bn_stats_sums_f64 / mau_bn_finalize_train with count = 0) reproduces the reference's single-device statistics on the
   joint batch with RAGGED per-rank batches (fixture G8, generated from the reference).  The kernels that produce the sums
   and apply the result need a GPU: that path (real module, kernels, hooks; 2 ranks, ragged batches) is
   tests/test_gpu_dist_rehearsal.py; here the exchange itself (product code: functional._all_reduce_ -> dist.all_reduce_sum)
   and the count convention run on CPU over gloo.
"""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import unet_ref as R
from tests.helpers import load_npz, rel_err, sub, t

WORLD = 2


class TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.unused = torch.nn.Linear(3, 3)             # never used in forward -> grad stays None
        self.a = torch.nn.Linear(5, 16)
        self.b = torch.nn.Linear(16, 16)
        self.c = torch.nn.Linear(16, 2)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker_gradsync(rank, init_file, out_dir):
    import mau_amd
    from mau_amd.dist import GradSync
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=WORLD)
    torch.manual_seed(0)
    net = TinyNet()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 5, generator=g)
    y = torch.randn(8, 2, generator=g)
    xs, ys = x.chunk(WORLD)[rank], y.chunk(WORLD)[rank]
    sync = GradSync(net, bucket_bytes=256)              # tiny buckets -> several collectives, exercised in order
    assert len(sync.buckets) > 2
    for step in range(2):                               # second step checks re-arming after zero_grad
        sync.begin()
        torch.nn.functional.mse_loss(net(xs), ys).backward()
        sync.finish()
        grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}
        net.zero_grad(set_to_none=True)
    torch.save(grads, os.path.join(out_dir, f"grads_{rank}.pt"))
    dist.destroy_process_group()


def test_gradsync_world2_matches_single_process(tmp_path):
    init = tempfile.mktemp(dir=tmp_path)
    mp.spawn(_worker_gradsync, args=(init, str(tmp_path)), nprocs=WORLD, join=True)
    torch.manual_seed(0)
    net = TinyNet()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 5, generator=g)
    y = torch.randn(8, 2, generator=g)
    torch.nn.functional.mse_loss(net(x), y).backward()      # mean over the joint batch == mean of per-rank means
    g0 = torch.load(os.path.join(tmp_path, "grads_0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "grads_1.pt"))
    for k, p in net.named_parameters():
        if k.startswith("unused"):
            assert g0[k] is None and g1[k] is None
            continue
        assert torch.equal(g0[k], g1[k]), k                  # identical on every rank
        assert rel_err(g0[k], p.grad) < 1e-6, k


def _worker_syncbn(rank, init_file, out_dir):
    import mau_amd
    from mau_amd.functional import BNState, _all_reduce_
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=WORLD)
    d = load_npz("g8_syncbn.npz")
    sd0 = sub(d, "sd0")
    xa = t(d["x"])
    x = xa[:xa.shape[0] - 1] if rank == 0 else xa[xa.shape[0] - 1:]          # ragged: all but one sample | one sample
    y1 = torch.nn.functional.conv2d(x, sd0["conv1.weight"], sd0["conv1.bias"], padding=1)
    C = y1.shape[1]
    # what the conv epilogue + mau_bn_stats_sums_f64 produce on each rank: [sum(y) | sum(y^2) | local pixel count] in fp64
    sums = torch.cat([y1.double().sum(dim=(0, 2, 3)), (y1.double() ** 2).sum(dim=(0, 2, 3)), torch.tensor([float(y1.numel() // C)], dtype=torch.float64)])
    st = BNState(training=True, C0=x.shape[1], group=dist.group.WORLD, world=WORLD)
    _all_reduce_(sums, st)                                   # the product's exchange
    count = float(sums[2 * C])                               # the global count travelled with the sums (mau_bn_finalize_train, count = 0)
    mean = sums[:C] / count
    var = sums[C:2 * C] / count - mean * mean
    scale = sd0["bn1.weight"].double() / torch.sqrt(var + 1e-5)
    shift = sd0["bn1.bias"].double() - mean * scale
    a1 = torch.relu(y1.double() * scale[None, :, None, None] + shift[None, :, None, None]).float()
    rm = 0.9 * sd0["bn1.running_mean"].double() + 0.1 * mean
    rv = 0.9 * sd0["bn1.running_var"].double() + 0.1 * var * count / (count - 1)
    torch.save({"a1": a1, "rm": rm.float(), "rv": rv.float()}, os.path.join(out_dir, f"bn_{rank}.pt"))
    dist.destroy_process_group()


def test_syncbn_exchange_world2_equals_reference_single_device(tmp_path):
    init = tempfile.mktemp(dir=tmp_path)
    mp.spawn(_worker_syncbn, args=(init, str(tmp_path)), nprocs=WORLD, join=True)
    d = load_npz("g8_syncbn.npz")
    sd0, sd1 = sub(d, "sd0"), sub(d, "sd1")
    x = t(d["x"])
    # reference: single device, joint batch (first half of the VGG block)
    y1 = torch.nn.functional.conv2d(x, sd0["conv1.weight"], sd0["conv1.bias"], padding=1)
    ref = torch.relu(torch.nn.functional.batch_norm(y1, None, None, sd0["bn1.weight"], sd0["bn1.bias"], True, 0.1, 1e-5))
    parts = [torch.load(os.path.join(tmp_path, f"bn_{r}.pt")) for r in range(WORLD)]
    got = torch.cat([p["a1"] for p in parts], 0)
    assert rel_err(got, ref) < 1e-5
    for p in parts:                                          # running stats identical on every rank == reference's
        assert rel_err(p["rm"], sd1["bn1.running_mean"]) < 1e-5
        assert rel_err(p["rv"], sd1["bn1.running_var"]) < 1e-5


def test_comm_cache_is_keyed_on_the_group_object_and_destroy_comms_clears_it(monkeypatch):
    """``_COMMS`` was keyed by ``id(group)`` alone -- after destroy_process_group() CPython may hand the id to a NEW
    group and the stale communicator came back.  The entry now holds the group object and a lookup checks identity."""
    from mau_amd import dist as D

    class FakeComm:
        made = 0

        def __init__(self, group):
            FakeComm.made += 1
            self.group, self.destroyed, self.aborted = group, False, False

        def destroy(self):
            self.destroyed = True

        def abort(self):
            self.aborted = True

    monkeypatch.setattr(D, "RcclComm", FakeComm)
    monkeypatch.setattr(D.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(D.torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(D.dist, "get_backend", lambda g=None: "nccl")
    monkeypatch.delenv("MAU_RCCL_DIRECT", raising=False)
    D._COMMS.clear()
    g1, g2 = object(), object()
    c1 = D.rccl_comm(g1, "bn")
    assert D.rccl_comm(g1, "bn") is c1 and D.rccl_comm(g1, "grad") is not c1 and FakeComm.made == 2
    # a different group object that lands on the same key (what id reuse after a destroyed group looks like) gets its OWN communicator
    key = (id(g1), "bn", 0)
    D._COMMS[(id(g2), "bn", 0)] = D._COMMS[key]                      # a stale entry sitting under g2's id
    c2 = D.rccl_comm(g2, "bn")
    assert c2 is not c1 and c2.group is g2
    # explicit choice beats the environment, the environment beats the default
    assert D.rccl_comm(g1, "x", direct=False) is None
    monkeypatch.setenv("MAU_RCCL_DIRECT", "0")
    assert D.rccl_comm(g1, "bn") is None and D.rccl_comm(g1, "bn", direct=True) is c1
    comms = [c for _, c in D._COMMS.values()]
    D.destroy_comms()
    assert not D._COMMS and all(c.destroyed for c in comms)
    monkeypatch.delenv("MAU_RCCL_DIRECT")
    c3 = D.rccl_comm(g1, "bn")
    D.destroy_comms(abort=True)
    assert c3.aborted and not D._COMMS


def test_collective_watchdog_aborts_and_exits_when_no_step_finishes():
    """A directly-issued ncclAllReduce has no timeout: the training driver's watchdog turns a hung collective into
    ncclCommAbort + a non-zero exit."""
    import time
    from mau_amd import dist as D

    class FakeComm:
        aborted = False

        def abort(self):
            FakeComm.aborted = True

    D._COMMS.clear()
    D._COMMS[(1, "grad", 0)] = (object(), FakeComm())
    codes = []
    wd = D.CollectiveWatchdog(timeout_s=0.4, _exit=codes.append)
    for _ in range(4):                                               # steps keep finishing: nothing happens
        time.sleep(0.2)
        wd.kick()
    assert not wd.fired and not codes
    time.sleep(1.2)                                                  # ... then one never does
    assert wd.fired and codes == [86] and FakeComm.aborted and not D._COMMS
    wd.close()


def test_collective_watchdog_pauses_closes_and_bounds_a_blocking_abort():
    """a validation pass / checkpoint write is not a hung collective (``paused()``); a closed watchdog never fires (the
    training loop closes it in ``finally``); an ``ncclCommAbort`` that blocks does not keep the process from leaving."""
    import threading
    import time
    from mau_amd import dist as D

    D._COMMS.clear()
    codes = []
    wd = D.CollectiveWatchdog(timeout_s=0.3, _exit=codes.append)
    with wd.paused():
        time.sleep(1.0)                                              # three timeouts' worth of "no step": paused, nothing happens
    assert not wd.fired and not codes
    time.sleep(0.15)
    assert not wd.fired                                              # ... and leaving the block restarted the clock
    with wd:                                                         # close() on the way out, also when the body raises
        pass
    time.sleep(0.8)
    assert not wd.fired and not codes

    release = threading.Event()

    class BlockingComm:
        def abort(self):
            release.wait(10.0)

    D._COMMS[(2, "grad", 0)] = (object(), BlockingComm())
    wd = D.CollectiveWatchdog(timeout_s=0.3, _exit=codes.append)
    wd.abort_timeout_s = 0.3
    t0 = time.monotonic()
    while not codes and time.monotonic() - t0 < 5.0:
        time.sleep(0.05)
    assert codes == [86] and time.monotonic() - t0 < 3.0             # left although the abort is still blocked
    release.set()
    wd.close()
    time.sleep(0.2)
    D._COMMS.clear()
