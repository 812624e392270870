#!/bin/bash
# compile-time ablations of the bf16x3 kernels: per-kernel averages of one level-3 block (tools/level_kernels.sh 3 x3) per variant library
O=$PWD/gpurun_out/x3_ablate; mkdir -p $O
tools/level_kernels.sh 3 x3 $O/base.txt
for v in "$@"; do DCPT_TOOL_LIB=$PWD/experiments/lib/libdcpt_hip_x3_$v.so tools/level_kernels.sh 3 x3 $O/$v.txt; done
cd $O; grep -H "x3_kernel" base.txt $(for v in "$@"; do echo $v.txt; done) | sed 's/ms\/step.*avg=/ /' | cut -c1-110
