#!/bin/bash
# PMC counters + durations per kernel for an arbitrary command (one rocprofv3 pass; --pmc alone with --kernel-trace, as gpurun requires):
#   tools/pmc_cmd.sh "<counters>" <kernel-name-substring> <command...>      e.g.  tools/pmc_cmd.sh "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" ws_gemm tools/ubench/ws_gemm_f32
R=$PWD; C="$1"; K="$2"; shift 2
CMD="$@"
case "$1" in /*) ;; *) if [ -e "$R/$1" ]; then CMD="$R/$@"; fi;; esac   # (a path inside the repo; anything else, e.g. python, is found on PATH)
cd /tmp; export TMPDIR=/tmp
D=$(mktemp -d)
timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o t -- $CMD > $D/log.txt 2>&1
python - "$D" "$K" <<'PY'
import csv, glob, sys, collections
d, key = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print("no counter file"); print(open(d + "/log.txt").read()[-2000:]); sys.exit()
dur = {}
for r in csv.DictReader(open(t[0])) if t else []:
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); us = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if key not in n: continue
    short = n.split("(")[0][-60:] + " grid=" + r.get("Grid_Size", "?")
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    k = (short, r["Dispatch_Id"])
    if k not in seen:
        seen.add(k); cnt[short] += 1; us[short] += dur.get(r["Dispatch_Id"], 0.0)
for s in acc:
    a = {c: v / cnt[s] for c, v in acc[s].items()}
    line = f"{s}: launches {cnt[s]} avg {us[s] / cnt[s]:.1f} us (under the profiler)  " + " ".join(f"{c}={v:.4g}" for c, v in sorted(a.items()))
    if "GRBM_GUI_ACTIVE" in a and us[s] > 0:
        line += f"  | clock {a['GRBM_GUI_ACTIVE'] / 8 / (us[s] / cnt[s]) / 1e3:.2f} GHz (GUI_ACTIVE is summed over the 8 XCDs)"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            line += f"  MfmaUtil {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8 * 1024):.1f} %"
    print(line)
PY
rm -rf $D
