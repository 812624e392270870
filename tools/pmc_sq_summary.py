"""Summarise a tools/pmc_sq.sh directory: mean SQ counters per kernel symbol, as % of SQ_WAVE_CYCLES."""
import csv, glob, collections, re, sys
out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'gemm|ubench'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'^void ', '', k).split('(')[0][:70]
        agg[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for k, d in agg.items():
    if not re.search(pat, k):
        continue
    n = len(next(iter(d.values())))
    dur = sum(t for _, t in next(iter(d.values()))) / n / 1e3
    print(f"{k}  n={n}  avg {dur:.1f} us")
    m = {c: sum(x for x, _ in v) / len(v) for c, v in d.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0) or 1
    for c, v in sorted(m.items()):
        print(f"   {c:28s} {v:14.4g}  {v / wc * 100:7.1f}% of WAVE_CYCLES")
