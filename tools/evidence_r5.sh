#!/bin/bash
# Round-5 evidence on ONE GPU box (gpurun): full GPU test log, rocprofv3 kernel stats + PMC passes of the headline command, step budget of the
# default two-stream command, default bench line (with every secondary), per-kernel tables of the bf16 steps (NAFNet-64, DCPT 128 / 256) and of
# Restormer, level probes, the level-3 bf16 block with the PMC traffic of its grouped weight-gradient launch, DDP probe.  -> gpurun_out/evidence_<tag>/
TAG=${1:-r5}; R=$PWD; O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
(timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -6) > $O/pytest_gpu_full.log 2>&1
tools/profile_gpu.sh $TAG > $O/profile_gpu.log 2>&1
cp $R/gpurun_out/prof_$TAG/summary.txt $O/rocprofv3_summary_serialized.txt
cp $(find $R/gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv 2>/dev/null
python tools/pmc_traffic.py $R/gpurun_out/prof_$TAG > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp; export TMPDIR=/tmp; D=$(mktemp -d)
rocprofv3 --kernel-trace --output-format csv -d $D -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > $O/step_budget_bench.log 2>&1
cd $R; python tools/step_budget.py $(find $D -name "bench_kernel_trace.csv") --json $O/step_budget.json > $O/step_budget_default_command.txt 2>&1; rm -rf $D
cp $O/step_budget.json profiles/step_budget.json   # (so that the bench line below reads the budget of THIS library: step_budget.stale = false)
T0=$SECONDS; (timeout 900 python bench.py 2>/dev/null | tail -1) > $O/bench_default_command.json; echo "python bench.py (default command, every secondary + cpu baseline): $((SECONDS - T0)) s wall" > $O/bench_time.txt
tools/kernel_table.sh $O/naf_bf16_kernels.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2
tools/kernel_table.sh $O/naf_bf16_kernels_serialized.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2 --side-stream 0
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
tools/kernel_table.sh $O/restormer_balanced_kernels.txt 5 python $R/bench_extra.py --workload restormer --steps 3 --warmup 2
tools/kernel_table.sh $O/x3_step_kernels.txt 5 python $R/bench.py --no-cpu-baseline --no-secondary --no-prof --gemm-precision bf16x3 --steps 4 --warmup 1
(timeout 300 python tools/level_probe.py 2>&1 | tail -6) > $O/level_probe_fp32.txt
(timeout 300 python tools/level_probe.py bf16 2>&1 | tail -6) > $O/level_probe_bf16.txt
tools/level_kernels.sh 3 bf16 $O/bf16_block_level3_kernels.txt
tools/level_kernels.sh 0 bf16 $O/bf16_block_level0_kernels.txt
(echo "# rocprofv3 --pmc (one counter group per pass) of tools/level_trace.py 3 bf16; FETCH_SIZE / WRITE_SIZE in KiB per launch, FETCH_SIZE x 2 on gfx950"; tools/pmc_kernel.sh 3 bf16 gemm_tn_bf16_256 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; echo "# wgrad_finish_kernel"; tools/pmc_kernel.sh 3 bf16 wgrad_finish "FETCH_SIZE" "WRITE_SIZE") > $O/pmc_tn256_level3.txt 2>&1
(timeout 300 python tools/tn256_probe.py 2>&1 | tail -8) > $O/tn256_probe.txt
(echo "# chain_fwd_bf16_kernel, level-3 block (tools/level_trace.py 3 bf16): FETCH_SIZE / WRITE_SIZE in KiB per launch (FETCH_SIZE x 2 on gfx950), MFMA busy, clock"; tools/pmc_kernel.sh 3 bf16 chain_fwd "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE") > $O/pmc_chain_level3.txt 2>&1
tools/fp32_power_ceiling.sh $O/fp32_power_ceiling.txt > /dev/null 2>&1
(timeout 300 python tools/ddp_probe.py 2>&1 | grep -v INFO | tail -6) > $O/ddp_probe.txt
tools/extras_all.sh gpurun_out/evidence_$TAG > $O/extras.log 2>&1
(timeout 300 python bench_extra.py --workload restormer 2>&1 | tail -1) > $O/extra_restormer_balanced.json
(timeout 300 python bench_extra.py --workload restormer --restormer-save full 2>&1 | tail -1) > $O/extra_restormer_full.json
tools/kernel_table.sh $O/infer2k_bf16_kernels_two_streams.txt 4 python $R/bench_extra.py --workload infer2k --dtype bf16 --steps 3 --warmup 1
(for d in bf16 fp32; do for n in 1 2 4; do python bench_extra.py --workload infer2k --dtype $d --tile-streams $n --steps 5 --warmup 2 2>/dev/null | tail -1; done; done) > $O/infer2k_streams.txt
(timeout 600 python tests/stream_stress.py --reps 30 2>&1 | grep "differing\|FAILED\|stable") > $O/stream_stress.txt
(timeout 900 python tests/fuzz_shapes.py --seed 6 --n 40 2>&1 | tail -5) > $O/fuzz_seed6_tail.txt
find $R/gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
tail -3 $O/pytest_gpu_full.log; cat $O/bench_default_command.json | cut -c1-300
