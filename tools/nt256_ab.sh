#!/bin/bash
# A/B of the 256 x 256 bf16 NT kernel on the GPU box: level timings + per-kernel tables with the kernel on (default) and off.
# The switch (DCPT_NT256) only exists in a tuning build:   tools/build_variant.sh TUNE "gemm_bf16.hip"   (before gpurun)
#   tools/nt256_ab.sh <outdir>
OUT=$PWD/${1:-gpurun_out/nt256}; mkdir -p $OUT
export DCPT_TOOL_LIB=$PWD/experiments/lib/libdcpt_hip_TUNE.so
for v in 0 1; do
  (DCPT_NT256=$v timeout 300 python tools/level_probe.py bf16 2>&1 | tail -7) > $OUT/level_probe_nt256_$v.txt 2>&1
  DCPT_NT256=$v timeout 300 tools/level_kernels.sh 3 bf16 $OUT/level3_kernels_nt256_$v.txt
done
cat $OUT/level_probe_nt256_0.txt $OUT/level_probe_nt256_1.txt; grep gemm_nt $OUT/level3_kernels_nt256_*.txt
