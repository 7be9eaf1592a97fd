"""Per-kernel totals of ONE train step from a rocprofv3 kernel trace of bench.py --no-graph: python scripts/step_trace.py <kernel_trace.csv> [step]"""
import csv, sys, re, collections
f = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
marks = [i for i, n in enumerate(names) if 'multi_tensor_apply' in n]
ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 20]
a, b = ends[which], ends[which + 1]
agg = collections.OrderedDict(); tot = 0
for r in rows[a + 1:b + 1]:
    n = re.sub(r'\(.*', '', r['Kernel_Name'])[:100]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = agg.setdefault(n, [0, 0]); k[0] += 1; k[1] += d; tot += d
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{c:4d} {d/1e3:9.1f} us  {n}")
span = int(rows[b]['End_Timestamp']) - int(rows[a + 1]['Start_Timestamp'])
print(f"total kernel time {tot/1e3:.1f} us, span {span/1e3:.1f} us, kernels {b - a}")
