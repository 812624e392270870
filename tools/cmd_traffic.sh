#!/bin/bash
# fabric traffic per kernel of an arbitrary command: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KiB, FETCH x 2 on gfx950) + kernel-trace
# durations of the same passes ->  one line per kernel name: launches, avg us, read / write MB per launch, TB/s.   tools/cmd_traffic.sh <out.txt> <command...>
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$1; shift
mkdir -p $(dirname $OUT); cd /tmp; export TMPDIR=/tmp
D=$(mktemp -d)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$c -o t -- "$@" > $D/log_$c.txt 2>&1
done
python - $D > $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
val = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.Counter(); us = collections.defaultdict(float)
for c in val:
    f = glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True)
    t = glob.glob(f"{d}/{c}/**/*kernel_trace.csv", recursive=True)
    if not f:
        print("no counter file for", c); print(open(f"{d}/log_{c}.txt").read()[-1500:]); continue
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t[0]))} if t else {}
    seen = set()
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
        if r["Counter_Name"] != c: continue
        val[c][n] += float(r["Counter_Value"])
        if c == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[n] += 1; us[n] += dur.get(r["Dispatch_Id"], 0.0)
tot = 0.0
rows = []
for n in cnt:
    k = cnt[n]
    rd = val["FETCH_SIZE"][n] * 2 * 1024 / 1e6 / k; wr = val["WRITE_SIZE"][n] * 1024 / 1e6 / k; u = us[n] / k
    rows.append((us[n], n, k, u, rd, wr))
for t_, n, k, u, rd, wr in sorted(rows, reverse=True):
    print(f"{n:64s} n={k:5d} avg {u:8.1f} us  total {t_/1e3:8.2f} ms  read {rd:8.1f} MB  write {wr:8.1f} MB  -> {(rd+wr)/max(u,1e-9):5.2f} TB/s")
PY
tail -1 $D/log_FETCH_SIZE.txt >> $OUT
rm -rf $D
