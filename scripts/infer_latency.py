#!/usr/bin/env python3
"""App-path latency (app/model_utils.py:102-109): one 1x23x512x512 tile, eval forward: eager vs hipGraph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mau_amd
for B, S in ((1, 512), (1, 250), (8, 512), (50, 250)):
    torch.manual_seed(0)
    net = mau_amd.UrbanPredictor("unet", 23, 828, 64, 8, 64, 96, 2, temporal_embeddings=False, metadata_embeddings=True).cuda().eval()
    x, ts, md = torch.randn(B, 23, S, S).cuda(), torch.randn(B, 828).cuda(), torch.randn(B, 8).cuda()
    def bench(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        te = bench(lambda: net(x, ts, md))
    sess = mau_amd.GraphedInference(net, x, ts, md)
    tg = bench(lambda: sess(x, ts, md))
    print(f"B={B:3d} {S}x{S}x23: eager {te:.3f} ms   hipGraph replay {tg:.3f} ms   ({tg / B:.3f} ms/tile)")
