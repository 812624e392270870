R=$PWD; O=$R/gpurun_out/r4_t12; mkdir -p $O
tools/build_variant.sh mr "gemm_tn_bf16_256.hip nafblock_bf16.hip dchead_bf16.hip" > $O/build.log 2>&1
L=$R/experiments/lib/libdcpt_hip_mr.so
for v in 256 1024 2048 256 1024 2048; do
  echo "MIN_ROWS=$v dcpt128: $(DCPT_TOOL_LIB=$L DCPT_TN_MIN_ROWS=$v python tools/bench_extra_variant.py --workload dcpt --dtype bf16 2>&1 | tail -1 | cut -c140-175)"
done
