#!/bin/bash
# Round evidence on ONE GPU box (gpurun): full GPU test log, rocprofv3 kernel stats + PMC passes of the headline command, step budget of the
# default two-stream command, default bench line, bf16 / x3 tables, secondary workloads.  Everything lands in gpurun_out/<tag>/.
TAG=${1:-r3}; R=$PWD; O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
(timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -6) > $O/pytest_gpu_full.log 2>&1
tools/profile_gpu.sh $TAG > $O/profile_gpu.log 2>&1
cp $R/gpurun_out/prof_$TAG/summary.txt $O/rocprofv3_summary_serialized.txt
cp $(find $R/gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv 2>/dev/null
python tools/pmc_traffic.py $R/gpurun_out/prof_$TAG > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp; export TMPDIR=/tmp; D=$(mktemp -d)
rocprofv3 --kernel-trace --output-format csv -d $D -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > $O/step_budget_bench.log 2>&1
cd $R; python tools/step_budget.py $(find $D -name "bench_kernel_trace.csv") --json $O/step_budget.json > $O/step_budget_default_command.txt 2>&1; rm -rf $D
cp $O/step_budget.json profiles/step_budget.json   # (so that the bench line below reads the budget of THIS library: step_budget.stale = false)
(timeout 600 python bench.py 2>&1 | tail -1) > $O/bench_default_command.json
tools/kernel_table.sh $O/naf_bf16_kernels.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2
tools/kernel_table.sh $O/x3_step_kernels.txt 5 python $R/bench.py --no-cpu-baseline --no-secondary --no-prof --gemm-precision bf16x3 --steps 4 --warmup 1
(timeout 300 python tools/level_probe.py 2>&1 | tail -6) > $O/level_probe_fp32.txt
(timeout 300 python tools/level_probe.py bf16 2>&1 | tail -6) > $O/level_probe_bf16.txt
tools/level_kernels.sh 3 bf16 $O/bf16_block_level3_kernels.txt
tools/level_kernels.sh 0 bf16 $O/bf16_block_level0_kernels.txt
(timeout 300 python tools/level_probe.py x3 2>&1 | tail -6) > $O/level_probe_x3.txt
(timeout 200 python tools/clock_watch.py 2>&1 | tail -8) > $O/clock_watch.txt
tools/extras_all.sh gpurun_out/evidence_$TAG > $O/extras.log 2>&1
for m in full balanced lean; do (timeout 300 python bench_extra.py --workload restormer --restormer-save $m 2>&1 | tail -1) > $O/extra_restormer_$m.json; done
find $R/gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
tail -3 $O/pytest_gpu_full.log; cat $O/bench_default_command.json | cut -c1-300
