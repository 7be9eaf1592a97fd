#!/usr/bin/env python3
"""Host -> network-input throughput of the two input paths at the reference's tile shape (23 x 250 x 250), PCIe included:
   dense   : 23 fp32 planes per tile from pinned host memory + NCHW->NHWC-ld conversion (what the reference ships)
   compact : 2 uint8 class maps + 5 fp32 planes + device-side one-hot / flip / cast (mau_amd.data)
Prints images/s and bytes per image for both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mau_amd
from mau_amd import functional as F_

B, H, W = int(os.environ.get("B", 32)), 250, 250
rng = np.random.default_rng(0)
a = torch.from_numpy(rng.integers(0, 9, (B, H, W)).astype(np.uint8)).pin_memory()
b = torch.from_numpy(rng.integers(0, 9, (B, H, W)).astype(np.uint8)).pin_memory()
cont = torch.from_numpy(rng.standard_normal((B, 5, H, W)).astype(np.float32)).pin_memory()
flip = torch.from_numpy((rng.random(B) < 0.5).astype(np.uint8)).pin_memory()
dense = torch.from_numpy(np.stack([mau_amd.data.expand_input(a[i].numpy(), b[i].numpy(), cont[i].numpy()) for i in range(B)])).pin_memory()

def run(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def dense_path():
    return F_.ToNHWC.apply(dense.to("cuda", non_blocking=True), torch.bfloat16)
def compact_path():
    return mau_amd.data.pack_tiles(a.to("cuda", non_blocking=True), b.to("cuda", non_blocking=True), cont.to("cuda", non_blocking=True),
                                   flip.to("cuda", non_blocking=True), torch.bfloat16)
da, db, dc, df = a.cuda(), b.cuda(), cont.cuda(), flip.cuda()
def pack_only():
    return mau_amd.data.pack_tiles(da, db, dc, df, torch.bfloat16)
td, tc, tk = run(dense_path), run(compact_path), run(pack_only)
bd = dense.numel() * 4 / B; bc = (a.numel() * 2 + cont.numel() * 4 + B) / B
print(f"dense   path: {B / td:9.0f} images/s  ({bd / 1e6:.2f} MB/image over PCIe, {bd * B / td / 1e9:.1f} GB/s)")
print(f"compact path: {B / tc:9.0f} images/s  ({bc / 1e6:.2f} MB/image over PCIe, {bc * B / tc / 1e9:.1f} GB/s)   x{td / tc:.2f}")
print(f"pack kernel alone (inputs resident): {tk * 1e6:.1f} us per batch of {B} = {B / tk:.0f} images/s, "
      f"{(bc + 24 * 2 * H * W) * B / tk / 1e9:.0f} GB/s of HBM traffic")
