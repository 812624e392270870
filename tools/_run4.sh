R=$PWD; O=$R/gpurun_out/r4_tn4; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bf16.py -q -x 2>&1 | tail -4) | tee $O/t.log
tools/build_variant.sh tn "nafblock_bf16.hip gemm_tn_bf16_256.hip" > $O/build.log 2>&1
L=$R/experiments/lib/libdcpt_hip_tn.so
for lv in 3 2 4; do
for v in 1 0; do
  DCPT_TOOL_LIB=$L DCPT_TN256=$v tools/level_kernels.sh $lv bf16 $O/l${lv}_tn256_$v.txt; echo "== level $lv serialized, DCPT_TN256=$v"; head -12 $O/l${lv}_tn256_$v.txt | cut -c1-140
done; done
for v in 1 0 1 0; do
  echo "== naf bf16 step, DCPT_TN256=$v"; DCPT_TOOL_LIB=$L DCPT_TN256=$v python tools/bench_extra_variant.py --workload naf --dtype bf16 2>&1 | tail -1 | cut -c1-200
done
for v in 1 0; do
  echo "== naf bf16 step serialized, DCPT_TN256=$v"; DCPT_TOOL_LIB=$L DCPT_TN256=$v python tools/bench_extra_variant.py --workload naf --dtype bf16 --side-stream 0 2>&1 | tail -1 | cut -c1-200
done
(timeout 300 python tools/tn256_probe.py 2>&1 | tail -8) | tee $O/probe.txt
