#!/bin/bash
# HBM bytes per launch of every kernel a command launches: FETCH_SIZE and WRITE_SIZE in separate PMC passes (kernel-trace only)
# usage: tools/pmc_hbm.sh <tag> <regex> <command...>
TAG=$1; PAT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hbm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o x -- "$@" > $OUT/$c.log 2>&1
done
cd $ROOT
python - "$OUT" "$PAT" <<'PY'
import csv, glob, collections, re, sys
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'^void ', '', k).split('(')[0][:60]
        agg[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for k, d in sorted(agg.items()):
    if not re.search(pat, k) or 'FETCH_SIZE' not in d: continue
    rd = sum(v for v, _ in d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 1024 * 2   # KiB, x2: gfx950 wide-read correction (MI355X_MICROARCH.md)
    wr = sum(v for v, _ in d.get('WRITE_SIZE', [(0, 0)])) / max(1, len(d.get('WRITE_SIZE', [1]))) * 1024
    us = sum(t for _, t in d['FETCH_SIZE']) / len(d['FETCH_SIZE']) / 1e3
    print(f"{k:55s} n={len(d['FETCH_SIZE']):4d} avg {us:8.1f} us  read {rd/1e6:8.1f} MB  write {wr/1e6:8.1f} MB  -> {(rd+wr)/us/1e6:6.2f} TB/s")
PY
find $OUT -name "*.csv" -size +20M -delete
