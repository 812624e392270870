#!/bin/bash
# Round-5 FINAL evidence on ONE GPU box (gpurun): tools/evidence_r5.sh without the unchanged fp32 power-ceiling run, plus what the last
# milestones added: per-kernel times of the network-edge convs (MFMA forms), Restormer's fabric traffic per kernel and GEMM classes by shape,
# the torch kernels left in the steps.   -> gpurun_out/evidence_<tag>/   (tools/copy_evidence.sh <tag> r5 copies the judged files)
TAG=${1:-r5b}; R=$PWD; O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
(timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6) > $O/pytest_gpu_full.log 2>&1
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
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
tools/kernel_table.sh $O/restormer_balanced_kernels.txt 5 python $R/bench_extra.py --workload restormer --steps 3 --warmup 2
tools/kernel_table.sh $O/x3_step_kernels.txt 5 python $R/bench.py --no-cpu-baseline --no-secondary --no-prof --gemm-precision bf16x3 --steps 4 --warmup 1
tools/kernel_table.sh $O/edge_convs_kernels.txt 1 python $R/tools/edge_times.py
(timeout 300 python tools/level_probe.py 2>&1 | tail -6) > $O/level_probe_fp32.txt
(timeout 300 python tools/level_probe.py bf16 2>&1 | tail -6) > $O/level_probe_bf16.txt
tools/level_kernels.sh 3 bf16 $O/bf16_block_level3_kernels.txt
tools/level_kernels.sh 0 bf16 $O/bf16_block_level0_kernels.txt
(echo "# rocprofv3 --pmc (one counter group per pass) of tools/level_trace.py 3 bf16; FETCH_SIZE / WRITE_SIZE in KiB per launch, FETCH_SIZE x 2 on gfx950"; tools/pmc_kernel.sh 3 bf16 gemm_tn_bf16_256 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum") > $O/pmc_tn256_level3.txt 2>&1
(timeout 300 python tools/ddp_probe.py 2>&1 | grep -v INFO | tail -6) > $O/ddp_probe.txt
tools/extras_all.sh gpurun_out/evidence_$TAG > $O/extras.log 2>&1
(timeout 300 python bench_extra.py --workload restormer 2>&1 | tail -1) > $O/extra_restormer_balanced.json
(timeout 300 python bench_extra.py --workload restormer --restormer-save full 2>&1 | tail -1) > $O/extra_restormer_full.json
tools/kernel_table.sh $O/infer2k_bf16_kernels_two_streams.txt 4 python $R/bench_extra.py --workload infer2k --dtype bf16 --steps 3 --warmup 1
(for d in bf16 fp32; do for n in 1 2; do python bench_extra.py --workload infer2k --dtype $d --tile-streams $n --steps 5 --warmup 2 2>/dev/null | tail -1; done; done) > $O/infer2k_streams.txt
(timeout 300 python tools/fill_trace.py --dcpt --size 256 2>&1 | grep -v "INFO\|Warn\|warn" | head -60) > $O/torch_kernels_dcpt_bf16_step.txt
(timeout 300 python tools/fill_trace.py --batch 32 2>&1 | grep -v "INFO\|Warn\|warn" | head -40) > $O/torch_kernels_fp32_step.txt
(timeout 600 python tests/stream_stress.py --reps 30 2>&1 | grep "differing\|FAILED\|stable") > $O/stream_stress.txt
(timeout 900 python tests/fuzz_shapes.py --seed 7 --n 40 2>&1 | tail -5) > $O/fuzz_seed7_tail.txt
find $R/gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
tail -3 $O/pytest_gpu_full.log; cat $O/bench_default_command.json | cut -c1-300
