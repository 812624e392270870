"""Per-launch durations of the kernels whose name contains a pattern, in launch order (rocprofv3 kernel-trace CSV):
    python tools/launch_list.py <kernel_trace.csv> <pattern> [max]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"{d:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}x{r.get('Grid_Size_Y', '')}  {r['Kernel_Name'][:90]}")
