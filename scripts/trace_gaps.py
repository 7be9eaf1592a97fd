"""GPU idle time inside the timed steps of a rocprofv3 kernel trace: python scripts/trace_gaps.py <kernel_trace.csv> [steps]."""
import csv, sys, statistics
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
st = [int(r['Start_Timestamp']) for r in rows]; en = [int(r['End_Timestamp']) for r in rows]
names = [r['Kernel_Name'] for r in rows]
# step boundaries: the fused-Adam launch group that ends every step
marks = [i for i, n in enumerate(names) if 'multi_tensor_apply' in n]
ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 20]
print("steps found", len(ends))
for a, b in zip(ends[:-1], ends[1:]):
    lo, hi = a + 1, b
    busy = sum(en[i] - st[i] for i in range(lo, hi + 1))
    cur = en[lo]; gap = 0; gaps = []
    for i in range(lo + 1, hi + 1):
        if st[i] > cur:
            gap += st[i] - cur; gaps.append((st[i] - cur, names[i - 1][:60], names[i][:60]))
        cur = max(cur, en[i])
    print(f"step: span {(en[hi]-st[lo])/1e6:.3f} ms  busy {busy/1e6:.3f}  idle {gap/1e6:.3f}  kernels {hi-lo+1}  median gap {statistics.median(g[0] for g in gaps)/1e3:.2f} us")
gaps.sort(reverse=True)
for g in gaps[:12]: print("   ", g)
