R=$PWD; O=$R/gpurun_out/r4_tn5; mkdir -p $O
tools/build_variant.sh tn "nafblock_bf16.hip gemm_tn_bf16_256.hip" > $O/build.log 2>&1
L=$R/experiments/lib/libdcpt_hip_tn.so
for v in 2 1 0 2 1 0; do
  echo "== naf bf16 step, DCPT_BF16_SIDE=$v"; DCPT_TOOL_LIB=$L DCPT_BF16_SIDE=$v python tools/bench_extra_variant.py --workload naf --dtype bf16 2>&1 | tail -1 | cut -c1-130
done
for v in 2 1 0; do
  echo "== dcpt bf16 128, DCPT_BF16_SIDE=$v"; DCPT_TOOL_LIB=$L DCPT_BF16_SIDE=$v python tools/bench_extra_variant.py --workload dcpt --dtype bf16 2>&1 | tail -1 | cut -c1-200
  echo "== dcpt bf16 256, DCPT_BF16_SIDE=$v"; DCPT_TOOL_LIB=$L DCPT_BF16_SIDE=$v python tools/bench_extra_variant.py --workload dcpt --dtype bf16 --size 256 2>&1 | tail -1 | cut -c1-200
done
