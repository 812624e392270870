#!/bin/bash
# (every pass runs under its own timeout: a TA_/TCP_ pass hung a box for 15 minutes once)
# PMC counters per kernel for one level's NAFBlock forward + backward (tools/level_trace.py), one rocprofv3 --pmc pass per counter group
#   tools/pmc_kernel.sh <level> <fp32|bf16> <kernel-name-substring> "<counters of pass 1>" ["<pass 2>" ...]
R=$PWD; L=$1; M=$2; K=$3; shift 3
cd /tmp; export TMPDIR=/tmp
for pass in "$@"; do
  D=$(mktemp -d)
  timeout 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $D -o t -- python $R/tools/level_trace.py $L $M > /dev/null 2>&1
  python - "$D" "$K" <<'PY'
import csv, glob, sys, collections
d, key = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if key not in n: continue
    short = n.replace("(anonymous namespace)::", "").split("(")[0][-70:]
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    k = (short, r["Dispatch_Id"])
    if k not in seen:
        seen.add(k); cnt[short] += 1
for s in acc:
    print(s, "launches", cnt[s], " ".join(f"{c}={v / cnt[s]:.4g}" for c, v in sorted(acc[s].items())))
PY
  rm -rf $D
done
