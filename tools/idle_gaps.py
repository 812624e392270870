"""Device idle time inside the steady-state steps of a rocprofv3 kernel trace (csv or csv.gz):  python tools/idle_gaps.py <trace> [marker]
A step ends with the last launch of a run of `marker` kernels (default adamw_kernel: the optimizer closes a training step); without marker
launches the last 60 % of the trace is one span.  Idle = span minus the union of the kernel intervals (all streams).  Prints per step: launches,
span, idle, the largest gaps with the kernel that follows them -- a host synchronisation shows up as a multi-millisecond gap in front of a
step's first kernels (tools/kernel_table.sh keeps the trace next to its table)."""
import csv, gzip, io, sys
path = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
fh = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
recs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(fh))
idx = [i for i, r in enumerate(recs) if marker in r[2]]
groups = []
for i in idx:
    if groups and i - groups[-1][-1] < 50: groups[-1].append(i)
    else: groups.append([i])
ends = [g[-1] for g in groups]
spans = [(a + 1, b + 1) for a, b in zip(ends[:-1], ends[1:])][-4:] if len(ends) >= 3 else [(int(0.4 * len(recs)), len(recs))]
for a, b in spans:
    step = recs[a:b]
    cur, idle, gaps = step[0][0], 0, []
    for s, e, n in step:
        if s > cur:
            idle += s - cur; gaps.append(((s - cur) / 1e3, n.replace("(anonymous namespace)::", "").replace("void ", "")[:44]))
        cur = max(cur, e)
    big = ", ".join(f"{g:.0f} us -> {n}" for g, n in sorted(gaps, reverse=True)[:4])
    print(f"{b - a:5d} launches  span {(step[-1][1] - step[0][0]) / 1e6:8.2f} ms  device idle {idle / 1e6:6.2f} ms in {len(gaps):4d} gaps   largest: {big}")
