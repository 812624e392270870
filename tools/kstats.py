"""Aggregate a rocprofv3 kernel trace (*_kernel_trace.csv, or a rocpd sqlite .db) per kernel name.

    python tools/kstats.py <trace> [steps] [rows] [--all]

With a CSV trace the table is STEADY-STATE by default: the launches are put in start order, the period of the kernel-name sequence at the end of
the trace is found (the smallest P >= 10 whose last two windows hold the same multiset of names) and only the trailing whole periods that repeat it are counted --
model construction, the optimizer's lazy state initialisation (two fills per parameter in the first step), weight-pack warm-up and the like
are left out instead of being divided by the step count.  --all (or no period found): everything, divided by `steps`."""
import csv, re, sqlite3, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
path = args[0]; steps = float(args[1]) if len(args) > 1 else 1.0; nrows = int(args[2]) if len(args) > 2 else 40
rows = {}
note = ""
if path.endswith('.db'):
    cur = sqlite3.connect(path).cursor()
    for n, c, ns in cur.execute("select name, count(*), sum(end-start) from kernels group by name"):
        rows[n] = (c, ns / 1e6)
else:
    recs = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))))
    names = [r[2] for r in recs]
    use = recs
    if "--all" not in sys.argv and len(names) > 40:
        last = names[-1]
        cands = [len(names) - 1 - i for i in range(len(names) - 2, -1, -1) if names[i] == last]   # distances to earlier launches of the last kernel
        from collections import Counter
        # (launches of two streams interleave differently from step to step: a period is a window with the same MULTISET of kernel names)
        same = lambda a, b: Counter(a) == Counter(b)
        P = next((p for p in cands if p >= 10 and 2 * p <= len(names) and same(names[-p:], names[-2 * p:-p])), None)
        if P:
            m = 1
            while (m + 1) * P <= len(names) and same(names[-(m + 1) * P:-m * P], names[-P:]):
                m += 1
            use, steps = recs[-m * P:], float(m)
            wall = (use[-1][1] - use[0][0]) / 1e6
            note = f" [steady state: the last {m} identical steps of {P} launches, {len(names) - m * P} earlier launches left out; first start to last end {wall / m:.2f} ms/step]"
    for s, e, n in use:
        c, t = rows.get(n, (0, 0.0)); rows[n] = (c + 1, t + (e - s) / 1e6)
    # device idle inside the counted span: the span minus the union of the kernel intervals; the largest gaps with the kernels around them
    cur_end, idle, gaps, prev = use[0][0], 0, [], None
    for s, e, n in use:
        if s > cur_end:
            idle += s - cur_end
            gaps.append((s - cur_end, prev, n))
        if e > cur_end:
            cur_end, prev = e, n
    nst = steps if note else 1.0
    note += f"\n  device idle inside the span: {idle / 1e6 / nst:.2f} ms/step in {len(gaps) / nst:.0f} gaps/step; gaps > 10 us: {sum(1 for g in gaps if g[0] > 10000) / nst:.1f}/step = {sum(g[0] for g in gaps if g[0] > 10000) / 1e6 / nst:.2f} ms/step"
    for g, a, b in sorted(gaps, key=lambda t: -t[0])[:6]:
        note += f"\n    {g / 1e3:8.1f} us between {re.sub(r'[(<].*', '', str(a))[-40:]} and {re.sub(r'[(<].*', '', b)[-40:]}"
tot = sum(t for _, t in rows.values())
print(f"total kernel time {tot:.2f} ms over {steps:g} steps = {tot/steps:.2f} ms/step{note}")
for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:nrows]:
    short = re.sub(r'\(anonymous namespace\)::', '', n)[:100]
    print(f"{t/steps:8.3f} ms/step {100*t/tot:5.1f}%  n/step={c/steps:7.1f} avg={1e3*t/c:8.1f}us  {short}")
