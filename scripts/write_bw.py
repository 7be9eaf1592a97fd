import torch
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (134, 268, 537, 1074):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(x)
    tf = timeit(lambda: x.fill_(1.0)); tc = timeit(lambda: y.copy_(x)); tz = timeit(lambda: x.zero_())
    print(f"{mb} MB: fill {tf*1e6:.1f} us {mb/1e6/tf:.2f} TB/s | zero {tz*1e6:.1f} us {mb/1e6/tz:.2f} TB/s | copy {tc*1e6:.1f} us {2*mb/1e6/tc:.2f} TB/s (r+w)")
