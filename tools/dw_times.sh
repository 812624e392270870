#!/bin/bash
# per-level durations of the depthwise kernels (rocprofv3 kernel trace of tools/level_trace.py), fp32 and bf16
#   tools/dw_times.sh <outdir>
R=$PWD; OUT=$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for d in fp32 bf16; do for l in 0 1 2 3; do
  rm -rf $OUT/tr; rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python $R/tools/level_trace.py $l $d > /dev/null 2>&1
  python $R/tools/kstats.py $(find $OUT/tr -name "*kernel_trace.csv") 5 60 | grep -E "dw_|dwr_|ln_" | sed "s/^/$d L$l /" | cut -c1-150
done; done
rm -rf $OUT/tr
