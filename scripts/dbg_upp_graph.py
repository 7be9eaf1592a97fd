#!/usr/bin/env python3
"""Triage of a hipGraph capture of the U-Net++ train step: python scripts/dbg_upp_graph.py <mode> [batch] [size]
modes: fwd (eval forward only) | fwdtrain (train-mode forward, no grad) | fb (forward + loss + backward) | full (+ mau AdamW) | fulltorch (+ torch AdamW)"""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import model as M

mode = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 16; S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
mt = os.environ.get("MT", "unet++")
torch.manual_seed(0)
flags = {} if mt == "unet++" else dict(temporal_embeddings=True, metadata_embeddings=True)
net = mau_amd.UrbanPredictor(mt, 6, 10, 64, 4, 64, 96, 2, base_filters=int(os.environ.get("BF", 64)), **flags).cuda().set_precision("bf16")
x = torch.randn(B, 6, S, S).cuda(); ts = torch.randn(B, 10).cuda(); md = torch.randn(B, 4).cuda(); tgt = torch.randn(B, 2, S, S).cuda()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, fused=True, capturable=True) if mode == "fulltorch" else mau_amd.AdamW(net.parameters(), lr=1e-4)
if mode == "fulltorch" or mode == "full":
    from mau_amd.train_graph import _make_capturable
    _make_capturable(opt)

def body():
    if mode in ("fwd", "fwdtrain"):
        with torch.no_grad():
            return net(x, ts, md)
    out = net(x, ts, md)
    loss = mau_amd.compute_loss_mse(out, tgt)["total"]
    loss.backward()
    if mode in ("full", "fulltorch"):
        opt.step()
    return loss

net.train(mode != "fwd")
with M.lstm_overlap_disabled():
    for _ in range(3):
        body()
        if mode == "fb":
            net.zero_grad(set_to_none=True)
        if mode in ("full", "fulltorch"):
            opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    print("eager ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r = body()
    print("capture ok", flush=True)
    g.replay(); torch.cuda.synchronize()
    print("replay ok", float(r.float().mean()), flush=True)
