"""profiles/pmc_traffic.json from a tools/profile_gpu.sh directory: HBM bytes per launch of each MFMA GEMM class, taken
over the launches of the class's MODAL shape (duration within 15 % of the median -- the 29 level-3 NAFBlocks dominate),
FETCH_SIZE KiB x2 (gfx950 wide-read correction of MI355X_MICROARCH.md) + WRITE_SIZE KiB, separate PMC passes."""
import csv, glob, json, os, re, statistics, sys
from collections import defaultdict

root = sys.argv[1]
ALOAD = {0: "plain", 1: "ln", 2: "scale", 3: "sg", 4: "gather", 5: "conv3", 6: "lnbf"}
EPI = {0: "plain", 1: "bias", 2: "resid", 3: "sgbwd", 4: "scatter", 5: "scatter_add", 6: "addscaled", 7: "mul", 8: "biasgate", 9: "dotcol", 10: "lnbwd", 11: "resid+ln"}


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'^void ', '', n).split('(')[0]


rows = defaultdict(lambda: defaultdict(list))  # symbol -> counter -> [(dur, value)]
for f in glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            rows[short(r['Kernel_Name'])][r['Counter_Name']].append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp']), float(r['Counter_Value'])))
out = {}
for sym, d in rows.items():
    m = re.match(r'gemm_(nt|tn)_kernel<(?:128, 128, 2, 2|128, 64, 4, 1), (\d+), (\d+)', sym)
    if not m or 'FETCH_SIZE' not in d or 'WRITE_SIZE' not in d:
        continue
    vals = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        med = statistics.median(t for t, _ in d[c])
        sel = [v for t, v in d[c] if abs(t - med) <= 0.15 * med]
        vals[c] = (sum(sel) / len(sel), len(sel), med / 1e3)
    rd, wr = vals['FETCH_SIZE'][0] * 1024 * 2, vals['WRITE_SIZE'][0] * 1024
    a, b = int(m.group(2)), int(m.group(3))
    name = f"gemm_nt<a={ALOAD[a]},epi={EPI[b]}>" if m.group(1) == 'nt' else f"gemm_tn<x={ALOAD[a]},y={ALOAD[b]}>"
    if name in out and out[name]["launches_used"] >= vals['FETCH_SIZE'][1]:
        continue   # two tile shapes of one class: keep the one with more launches (the level-3 blocks)
    out[name] = dict(hbm_bytes_per_launch=round(rd + wr), read_bytes=round(rd), write_bytes=round(wr), symbol=sym,
                     launches_used=vals['FETCH_SIZE'][1], median_us=round(vals['FETCH_SIZE'][2], 1))
try:
    dig = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dcpt_amd", "lib", "libdcpt_hip.digest")).read().strip()
except OSError:
    dig = None
# lib_digest: the build of libdcpt_hip.so that was profiled (run this tool with the same tree); bench.py flags the figures stale when
# the running library's digest differs
json.dump({"source": __doc__.strip() + f"  Raw passes: {os.path.basename(root)}.", "lib_digest": dig, "kernels": out},
          open('profiles/pmc_traffic.json', 'w'), indent=1)
for k, v in out.items():
    print(k, v)
