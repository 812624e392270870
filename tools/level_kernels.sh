#!/bin/bash
# per-kernel table of one NAFBlock forward + backward at a level (rocprofv3 kernel trace of tools/level_trace.py)
#   tools/level_kernels.sh <level> <fp32|bf16> <out.txt>      (environment knobs pass through)
R=$PWD; cd /tmp; export TMPDIR=/tmp
D=$(mktemp -d)
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/level_trace.py $1 $2 > /dev/null 2>&1
python $R/tools/kstats.py $(find $D -name "*kernel_trace.csv") 5 40 | grep -v "at::native\|rocclr" > $3
rm -rf $D
