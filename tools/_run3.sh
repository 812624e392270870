R=$PWD; O=$R/gpurun_out/r4_tn3; mkdir -p $O
for v in base NOLOAD NOMFMA NOEPI; do
  if [ $v = base ]; then tools/build_variant.sh tn_$v "gemm_tn_bf16_256.hip" > $O/build_$v.log 2>&1; else tools/build_variant.sh tn_$v "gemm_tn_bf16_256.hip" -DTN_ABL_$v > $O/build_$v.log 2>&1; fi
done
tools/build_variant.sh tn_NOLOADEPI "gemm_tn_bf16_256.hip" -DTN_ABL_NOLOAD -DTN_ABL_NOEPI > $O/build_x.log 2>&1
cd /tmp; export TMPDIR=/tmp
for v in base NOLOAD NOMFMA NOEPI NOLOADEPI; do
  D=$(mktemp -d)
  DCPT_TOOL_LIB=$R/experiments/lib/libdcpt_hip_tn_$v.so rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/tn256_probe.py > $O/probe_$v.txt 2>&1
  python - $D $v <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
import collections
seq = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'gemm_tn_bf16_256' in r['Kernel_Name'] or 'wgrad_finish' in r['Kernel_Name']]
# 7 shapes x 23 calls x 2 kernels
tn = [d for n, d in seq if 'gemm_tn' in n]; fi = [d for n, d in seq if 'finish' in n]
per = len(tn) // 7
print(sys.argv[2], "TN us per shape:", [round(sum(tn[i*per+3:(i+1)*per]) / (per-3), 1) for i in range(7)], "finish:", [round(sum(fi[i*per+3:(i+1)*per]) / (per-3), 1) for i in range(7)])
PY
  rm -rf $D
done
