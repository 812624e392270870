"""rocprofv3 kernel trace (csv) -> per (kernel, grid size) launch count and average duration, for kernels whose name contains a substring:
which SIZES of a bandwidth kernel are slow.   python tools/kernel_by_grid.py <kernel_trace.csv> <substring> [<substring> ...]"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if not any(k in n for k in sys.argv[2:]):
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][-56:]
    key = (short, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])))
    a = acc[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (k, g), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / 1e3:9.3f} ms  n={c:5d}  avg={t / c:9.1f} us  blocks={g:8d}  {k}")
