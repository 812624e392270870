"""Print the kernel timeline (both streams) of a window of the backward pass from a rocprofv3 kernel-trace CSV."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n).split('(')[0][:44]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r.get('Stream_Id', r.get('Thread_Id', '?')), n))
rows.sort()
# find the last ln_bwd_kernel<1> region? pick a window in the middle of the last step's backward: locate sgbwd GEMMs (epilogue 3)
idx = [i for i, r in enumerate(rows) if 'gemm_nt_kernel<128, 128, 2, 2, 0, 3' in r[4]]
i0 = idx[len(idx) // 2 + 10]
t0 = rows[i0][0]
for s, e, q, st, n in rows[i0: i0 + int(sys.argv[2]) if len(sys.argv) > 2 else i0 + 40]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{q:>3s}  {n}")
