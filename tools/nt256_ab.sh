#!/bin/bash
# A/B of the 256 x 256 bf16 NT kernel on the GPU box: parity (forced on the small test shapes) + level timings + per-kernel tables.
#   tools/nt256_ab.sh <outdir>
OUT=$PWD/${1:-gpurun_out/nt256}; mkdir -p $OUT
(DCPT_NT256=2 timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "nafblock_bf16_oracle or nafnet_bf16 or dcpt_step" 2>&1 | tail -15) > $OUT/parity_forced.log 2>&1
for v in 0 1; do
  (DCPT_NT256=$v timeout 300 python tools/level_probe.py bf16 2>&1 | tail -7) > $OUT/level_probe_nt256_$v.txt 2>&1
  DCPT_NT256=$v timeout 300 tools/level_kernels.sh 3 bf16 $OUT/level3_kernels_nt256_$v.txt
done
(timeout 300 python bench_extra.py --workload naf --dtype bf16 2>&1 | tail -1) > $OUT/naf_bf16.json 2>&1
tail -4 $OUT/parity_forced.log; cat $OUT/level_probe_nt256_0.txt $OUT/level_probe_nt256_1.txt; grep gemm_nt $OUT/level3_kernels_nt256_*.txt; cat $OUT/naf_bf16.json
