"""Rehearsal of the N>1 path on ONE GPU: two ranks share cuda:0, collectives over the gloo backend
(staged through host memory).  Checks that 2 ranks x batch b with SyncBN + GradSync reproduce a
single process with batch 2b, through the real module, kernels and hooks."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["MAU_ROOT"])
import torch.distributed as dist
import mau_amd
from mau_amd.dist import GradSync, init_process_group_from_env
rank, local, world = init_process_group_from_env()
torch.cuda.set_device(local)
torch.manual_seed(0)
kw = dict(model_type="unet", spatial_channels=6, seq_len=10, temporal_dim=8, meta_features=4, meta_dim=8, lstm_dim=12,
          out_channels=2, base_filters=8, temporal_embeddings=False, metadata_embeddings=True)
net = mau_amd.UrbanPredictor(**kw).cuda().set_precision("fp32").train()
g = torch.Generator().manual_seed(3)
x = torch.randn(4, 6, 32, 32, generator=g); ts = torch.randn(4, 10, generator=g); md = torch.randn(4, 4, generator=g); tgt = torch.randn(4, 2, 32, 32, generator=g)
if world > 1:
    # RAGGED local batches (3 + 1): the pixel count travels with the BatchNorm sums, so the ranks still agree on the global moments;
    # the MSE criterion is a mean over the local batch, so each rank weights its loss by its share of the global batch (what a
    # DistributedSampler with equal shards makes implicit)
    sl = slice(0, 3) if rank == 0 else slice(3, 4)
    x, ts, md, tgt = x[sl], ts[sl], md[sl], tgt[sl]
    net.set_sync_bn(dist.group.WORLD)
    sync = GradSync(net, bucket_bytes=64 << 10)
x, ts, md, tgt = x.cuda(), ts.cuda(), md.cuda(), tgt.cuda()
out = net(x, ts, md)
loss = mau_amd.compute_loss_mse(out, tgt)["total"]
if world > 1: loss = loss * (x.shape[0] * world / 4.0)        # GradSync averages over ranks: local mean x (local share x world) = global mean
if world > 1: sync.begin()
loss.backward()
if world > 1: sync.finish()
torch.cuda.synchronize()
res = {"out": out.detach().cpu(), "grads": {k: (p.grad.cpu().clone() if p.grad is not None else None) for k, p in net.named_parameters()},
       "rm": net.model.conv0_0.bn1.running_mean.cpu(), "rv": net.model.conv2_0.bn2.running_var.cpu()}
torch.save(res, os.path.join(os.environ["MAU_OUT"], f"res_w{world}_r{rank}.pt"))
if world > 1: dist.destroy_process_group()
'''


def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MAU_ROOT=ROOT, MAU_OUT=str(tmp_path), MAU_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, str(script)], check=True, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29517", str(script)], check=True, env=env, timeout=300)
    one = torch.load(tmp_path / "res_w1_r0.pt")
    r0, r1 = torch.load(tmp_path / "res_w2_r0.pt"), torch.load(tmp_path / "res_w2_r1.pt")
    out2 = torch.cat([r0["out"], r1["out"]], 0)
    assert float((out2 - one["out"]).abs().max() / one["out"].abs().max()) < 1e-4
    for k, gref in one["grads"].items():
        if gref is None:
            assert r0["grads"][k] is None
            continue
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k            # averaged gradients identical on both ranks
        err = float((r0["grads"][k] - gref).abs().max() / gref.abs().max().clamp_min(1e-12))
        assert err < 2e-3 or float((r0["grads"][k] - gref).abs().max()) < 1e-6, (k, err)
    for key in ("rm", "rv"):
        assert float((r0[key] - one[key]).abs().max() / one[key].abs().max()) < 1e-4
        assert torch.equal(r0[key], r1[key])


RCCL_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["MAU_ROOT"])
import torch.distributed as dist
import mau_amd
from mau_amd.dist import GradSync, all_reduce_sum, rccl_comm, RcclComm, NCCL_AVG, NCCL_SUM
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29519", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
DIRECT = os.environ.get("MAU_RCCL_DIRECT", "1") != "0"
# the collectives the data-parallel path issues, on device memory, through RCCL itself
t64 = torch.arange(128, dtype=torch.float64, device="cuda"); all_reduce_sum(t64, dist.group.WORLD)
t32 = torch.ones(1 << 20, device="cuda"); h = all_reduce_sum(t32[17:], dist.group.WORLD, async_op=True); h.wait()
torch.cuda.synchronize()
assert torch.equal(t64.cpu(), torch.arange(128, dtype=torch.float64)) and float(t32.sum()) == float(1 << 20)
if DIRECT:
    # ... and the way the product issues them: ncclAllReduce called directly on a named stream (dist.RcclComm), fp64 SUM on the
    # current stream (SyncBN message), fp32 AVG of an arena segment on a side stream ordered by events (gradient bucket)
    c = rccl_comm(dist.group.WORLD, "bn"); assert isinstance(c, RcclComm) and c.world == 1 and rccl_comm(dist.group.WORLD, "bn") is c
    c2 = rccl_comm(dist.group.WORLD, "grad"); assert c2 is not c
    t64 = torch.arange(257, dtype=torch.float64, device="cuda") * 0.5; c.all_reduce(t64, NCCL_SUM)
    side = torch.cuda.Stream(); arena = torch.zeros(1 << 22, device="cuda"); arena[1000:].fill_(3.0)
    side.wait_stream(torch.cuda.current_stream()); c2.all_reduce(arena[1000:], NCCL_AVG, side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    assert torch.equal(t64.cpu(), torch.arange(257, dtype=torch.float64) * 0.5) and float(arena.sum()) == 3.0 * ((1 << 22) - 1000)
    try:
        c.all_reduce(torch.zeros(4)); raise SystemExit("a host tensor must be refused")
    except RuntimeError:
        pass
else:
    assert rccl_comm(dist.group.WORLD, "bn") is None
kw = dict(model_type="unet", spatial_channels=6, seq_len=10, temporal_dim=8, meta_features=4, meta_dim=8, lstm_dim=12,
          out_channels=2, base_filters=8, temporal_embeddings=False, metadata_embeddings=True)
g = torch.Generator().manual_seed(3)
x = torch.randn(4, 6, 32, 32, generator=g).cuda(); ts = torch.randn(4, 10, generator=g).cuda()
md = torch.randn(4, 4, generator=g).cuda(); tgt = torch.randn(4, 2, 32, 32, generator=g).cuda()
res = {}
for prec in ("fp32", "bf16"):
    for synced in (False, True):
        torch.manual_seed(0)
        net = mau_amd.UrbanPredictor(**kw).cuda().set_precision(prec).train()
        sync = None
        if synced:
            net.set_sync_bn(dist.group.WORLD)
            sync = GradSync(net, dist.group.WORLD, bucket_bytes=64 << 10)
            assert len(sync.buckets) > 1 and (sync.comm is not None) == DIRECT and (net.model._rt.comm is not None) == DIRECT
        for step in range(2):                       # gradients accumulate over the two passes (no optimizer: Adam
            loss = mau_amd.compute_loss_mse(net(x, ts, md), tgt)["total"]     # would amplify rounding noise of ~0 grads)
            if sync: sync.begin()
            loss.backward()
            if sync: sync.finish()
        torch.cuda.synchronize()
        res[(prec, synced)] = {k: v.detach().float().cpu() for k, v in net.state_dict().items() if "running" in k}
        res[(prec, synced)].update({"grad." + k: p.grad.float().cpu() for k, p in net.named_parameters() if p.grad is not None})
if os.environ.get("MAU_TEST_SHORT") == "1":
    torch.save(res, os.path.join(os.environ["MAU_OUT"], "rccl1.pt"))
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
# U-Net++ with the temporal branch: the LSTM runs on its side stream under the process group (GradSync waits for every stream that
# produced a gradient of a bucket); 3 steps with fused AdamW, synced vs plain, must agree
kw2 = dict(model_type="unet++", spatial_channels=6, seq_len=24, temporal_dim=16, meta_features=4, meta_dim=16, lstm_dim=24, out_channels=2, base_filters=16)
ts2 = torch.randn(4, 24, generator=g).cuda()
for synced in (False, True):
    torch.manual_seed(1)
    net = mau_amd.UrbanPredictor(**kw2).cuda().set_precision("bf16").train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, fused=True)
    sync = None
    if synced:
        net.set_sync_bn(dist.group.WORLD)
        sync = GradSync(net, dist.group.WORLD, bucket_bytes=256 << 10)
    for step in range(3):
        loss = mau_amd.compute_loss_mse(net(x, ts2, md), tgt)["total"]
        if sync: sync.begin()
        loss.backward()
        if sync: sync.finish()
        if sync: assert all(p.grad is None or p.grad.data_ptr() == p._mau_grad_slot.data_ptr() for p in net.parameters())   # gradients live in the arena
        opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    res[("unet++", synced)] = {k: v.detach().float().cpu() for k, v in net.state_dict().items() if v.is_floating_point()}
# the data-parallel step as ONE hipGraph (GraphedTrainStep(grad_sync=...): the SyncBN and bucket all-reduces are captured RCCL nodes):
# five steps, captured at the third, against the same five steps launched eagerly -- bit for bit
kw3 = dict(model_type="unet", spatial_channels=6, seq_len=10, temporal_dim=8, meta_features=4, meta_dim=8, lstm_dim=12, out_channels=2,
           base_filters=16, temporal_embeddings=False, metadata_embeddings=True)
for graphed in (False, True):
    torch.manual_seed(2)
    net = mau_amd.UrbanPredictor(**kw3).cuda().set_precision("bf16").train()
    opt = mau_amd.AdamW(net.parameters(), lr=1e-3)
    net.set_sync_bn(dist.group.WORLD)
    sync = GradSync(net, dist.group.WORLD, bucket_bytes=64 << 10)
    crit = mau_amd.compute_loss_mse
    gstep = mau_amd.GraphedTrainStep(net, opt, crit, warmup=2, grad_sync=sync) if graphed else None
    gg = torch.Generator().manual_seed(9)
    losses = []
    for step in range(5):
        xb, mb, tb = torch.randn(4, 6, 32, 32, generator=gg).cuda(), torch.randn(4, 4, generator=gg).cuda(), torch.randn(4, 2, 32, 32, generator=gg).cuda()
        if graphed:
            losses.append(float(gstep(xb, ts, mb, tb)))
        else:
            loss = crit(net(xb, ts, mb), tb)["total"]
            sync.begin(); loss.backward(); sync.finish()
            opt.step(); opt.zero_grad()
            losses.append(float(loss))
    torch.cuda.synchronize()
    if graphed: assert gstep.graph is not None
    res[("dpgraph", graphed)] = {"losses": torch.tensor(losses), **{k: v.detach().float().cpu() for k, v in net.state_dict().items() if v.is_floating_point()}}
    sync.remove()
# the training driver on the data-parallel path (SyncBN + GradSync + eager step + validate() + best-validation checkpoint)
from mau_amd import train
from mau_amd.config import CONFIG
CONFIG.MODELS_DIR = os.environ["MAU_OUT"]
r = train.run(device="gpu", temporal_embeddings=False, metadata_embeddings=True, model_type="unet", jobid="rccl", epochs=2, steps_per_epoch=2,
              precision="bf16", val_batches=1, force_dist=True)
assert r["checkpoint_path"] and os.path.exists(r["checkpoint_path"]) and len(r["history"]) == 2 and all(v == v for e in r["history"] for v in e)
res["train_best"] = torch.tensor(r["best"])
torch.save(res, os.path.join(os.environ["MAU_OUT"], "rccl1.pt"))
dist.barrier(); dist.destroy_process_group()
'''


def test_rccl_collectives_one_rank_group(tmp_path):
    """The RCCL ("nccl") code path itself -- device-memory all-reduce of the fp64 BatchNorm sums, async bucketed
    gradient all-reduce with handles -- on a ONE-rank group (RCCL refuses two ranks on one GPU): two forward/backward
    passes with SyncBN + GradSync over the group must equal the same passes without a group."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MAU_ROOT=ROOT, MAU_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, str(script)], check=True, env=env, timeout=300)
    res = torch.load(tmp_path / "rccl1.pt")
    # fp32: the two statistics paths (fused finalize vs reduce + all-reduce + finalize) agree to rounding;
    # bf16: a last-bit difference in a BN scale can flip bf16 roundings downstream
    for prec, tol in (("fp32", 1e-4), ("bf16", 3e-2)):
        a, b = res[(prec, False)], res[(prec, True)]
        assert a.keys() == b.keys() and len(a) > 40
        for k in a:
            err = float((a[k] - b[k]).abs().max())
            assert err < tol * float(a[k].abs().max()) or err < 1e-6, (prec, k, err)
    # U-Net++ (LSTM side stream under the group, gradients written into the arena), 3 AdamW steps: parameters stay together
    a, b = res[("unet++", False)], res[("unet++", True)]
    assert a.keys() == b.keys()
    num = sum(float(((a[k] - b[k]).double() ** 2).sum()) for k in a)
    den = sum(float((a[k].double() ** 2).sum()) for k in a)
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5
    # the captured data-parallel step == the eager data-parallel step
    a, b = res[("dpgraph", False)], res[("dpgraph", True)]
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert float(res["train_best"]) < float("inf")


def test_process_group_nccl_path_still_works(tmp_path):
    """MAU_RCCL_DIRECT=0: the same messages through ProcessGroupNCCL (round 3's path, kept as the A/B switch)."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.replace("29519", "29521"))
    env = dict(os.environ, MAU_ROOT=ROOT, MAU_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0", MAU_RCCL_DIRECT="0", MAU_TEST_SHORT="1")
    subprocess.run([sys.executable, str(script)], check=True, env=env, timeout=300)
    res = torch.load(tmp_path / "rccl1.pt")
    a, b = res[("fp32", False)], res[("fp32", True)]
    for k in a:
        err = float((a[k] - b[k]).abs().max())
        assert err < 1e-4 * float(a[k].abs().max()) or err < 1e-6, (k, err)


def test_bench_self_launches_two_ranks(tmp_path):
    """``python bench.py --gpus 2`` outside torch.distributed.run must start its own two ranks (child processes; the
    parent never touches the GPU), and print ONE JSON line with n_gpus 2; N = 1 keeps printing its line directly.
    Two ranks share cuda:0 here, so the collectives run over gloo (MAU_DIST_BACKEND) -- rccl_ranks is then 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MAU_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--batch", "2", "--size", "64", "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["steps"] == 2 and rec["rccl_ranks"] == 0
    assert rec["config"]["sync_bn"] is True and rec["value"] > 0 and rec["roofline"]["achieved"] > 0
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env, capture_output=True, text=True, timeout=600)
    assert p1.returncode == 0, p1.stderr[-3000:]
    rec1 = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    assert rec1["n_gpus"] == 1 and rec1["rccl_ranks"] == 1 and "cpu_baseline" not in rec1


def test_bench_inference_replicas_two_ranks():
    """BASELINE configs[4] names N-GPU inference: ``bench.py --infer --gpus 2`` = two independent replicas (no collective on
    the data path), each with its GraphedInference session; one line, value = images of both ranks / max-over-ranks time."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MAU_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--infer", "--steps", "3", "--warmup", "1", "--batch", "2", "--size", "64", "--no-cpu-baseline", "--precision", "fp16", "--repeats", "2"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["metric"].startswith("inference images/sec")
    assert "hipGraph replay" in rec["config"]["launch"] and rec["timed_regions"]["repeats"] == 2 and rec["value"] > 0


def test_bench_under_torchrun_each_rank_supervises_its_worker():
    """The driver's launch line for N > 1 (``python -m torch.distributed.run ... bench.py --gpus N``): every launched process is the
    supervisor of one worker; the workers meet on a rendezvous of their own and rank 0's line comes out exactly once."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MAU_DP_GRAPH")}
    env.update(MAU_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "3", "--batch", "2", "--size", "64", "--no-cpu-baseline", "--repeats", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    # two ranks share one GPU here, so the collectives are gloo's (host-staged: not capturable) -- the captured attempt fails in
    # every worker and the supervisors fall back to fresh eager workers: the measurement survives
    assert rec["n_gpus"] == 2 and rec["config"]["sync_bn"] is True and rec["value"] > 0
    assert rec["config"]["launch"].startswith("eager") and "falling back" in p.stderr
