"""Aggregate a rocprofv3 kernel trace (rocpd sqlite .db or *_kernel_trace.csv) per kernel name."""
import csv, re, sqlite3, sys
path = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = {}
if path.endswith('.db'):
    cur = sqlite3.connect(path).cursor()
    for n, c, ns in cur.execute("select name, count(*), sum(end-start) from kernels group by name"):
        rows[n] = (c, ns / 1e6)
else:
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        c, t = rows.get(n, (0, 0.0)); rows[n] = (c + 1, t + d)
tot = sum(t for _, t in rows.values())
print(f"total kernel time {tot:.2f} ms over {steps:g} steps = {tot/steps:.2f} ms/step")
for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    short = re.sub(r'\(anonymous namespace\)::', '', n)[:100]
    print(f"{t/steps:8.3f} ms/step {100*t/tot:5.1f}%  n/step={c/steps:7.1f} avg={1e3*t/c:8.1f}us  {short}")
