#!/usr/bin/env python3
"""Time the persistent HIP LSTM (TemporalEncoder recurrence) forward / backward at the reference's shape: B=32 (or B env),
T=828, hidden 96; prints us per launch and ns per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd._lib import call, lib
B = int(os.environ.get("B", 32)); T = int(os.environ.get("T", 828)); H = int(os.environ.get("H", 96))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, generator=g).cuda()
lstm = torch.nn.LSTM(1, H, batch_first=True).cuda()
wi, wh, bi, bh = lstm.weight_ih_l0.detach(), lstm.weight_hh_l0.detach(), lstm.bias_ih_l0.detach(), lstm.bias_hh_l0.detach()
h = torch.empty(B, H, device="cuda"); gates = torch.empty(B, T, 4 * H, device="cuda"); cells = torch.empty(B, T, H, device="cuda")
dh = torch.randn(B, H, device="cuda")
dwi, dwh, dbi, dbh = torch.empty(4 * H, device="cuda"), torch.empty(4 * H, H, device="cuda"), torch.empty(4 * H, device="cuda"), torch.empty(4 * H, device="cuda")
ws = torch.empty(lib.mau_lstm_bwd_ws_elems(B, T, H), device="cuda")
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
f = lambda: call("mau_lstm_fwd", x.data_ptr(), wi.data_ptr(), wh.data_ptr(), bi.data_ptr(), bh.data_ptr(), h.data_ptr(), gates.data_ptr(), cells.data_ptr(), B, T, H, st)
fi = lambda: call("mau_lstm_fwd", x.data_ptr(), wi.data_ptr(), wh.data_ptr(), bi.data_ptr(), bh.data_ptr(), h.data_ptr(), None, None, B, T, H, st)
b = lambda: call("mau_lstm_bwd", x.data_ptr(), wh.data_ptr(), gates.data_ptr(), cells.data_ptr(), dh.data_ptr(), dwi.data_ptr(), dwh.data_ptr(), dbi.data_ptr(), dbh.data_ptr(), ws.data_ptr(), B, T, H, st)
tf, tfi, tb = timeit(f), timeit(fi), timeit(b)
with torch.no_grad():
    ref = lstm(x.unsqueeze(-1))[1][0][-1]
print(f"B={B} T={T} H={H}: fwd(train) {tf:.1f} us ({tf/T*1e3:.0f} ns/step), fwd(infer) {tfi:.1f} us, bwd(+dW GEMM+sums) {tb:.1f} us ({tb/T*1e3:.0f} ns/step); max|h - torch| {float((h-ref).abs().max()):.2e}")
