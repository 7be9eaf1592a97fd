#!/usr/bin/env python3
"""Host-side profile of the eager train step (cProfile over 10 steps): where the Python time of one step goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
torch.manual_seed(0)
net = mau_amd.UrbanPredictor("unet", 6, 10, 64, 4, 64, 96, 2, base_filters=64, temporal_embeddings=False, metadata_embeddings=True).cuda().set_precision("bf16").train()
opt = mau_amd.AdamW(net.parameters(), lr=1e-4)
B = 32
x = torch.randn(B, 6, 256, 256).cuda(); ts = torch.randn(B, 10).cuda(); md = torch.randn(B, 4).cuda(); tgt = torch.randn(B, 2, 256, 256).cuda()
def step():
    loss = mau_amd.compute_loss_mse(net(x, ts, md), tgt)["total"]
    loss.backward(); opt.step(); opt.zero_grad()
for _ in range(5): step()
torch.cuda.synchronize()
# host time per step when the GPU is not the limit: issue 10 steps, measure the time until the LAST LAUNCH RETURNS (not the sync)
t0 = time.perf_counter()
for _ in range(10): step()
t_issue = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 10
print(f"host issue time per step {t_issue*1e3:.2f} ms; wall per step {t_all*1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(25)
