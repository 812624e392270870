#!/bin/bash
# Round-6 evidence of the round's LAST library on ONE GPU box, most important first, every step under its own timeout:
# full GPU suite, rocprofv3 summary + PMC traffic of the headline command, step budget, the default bench line, steady-state kernel tables of the
# bf16 steps and of the head, the secondary workloads, fuzz + stream stress.   tools/evidence_r6.sh [tag]
TAG=${1:-r6}; R=$PWD; O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
(timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6) > $O/pytest_gpu_full.log 2>&1
timeout 2400 tools/profile_gpu.sh $TAG > $O/profile_gpu.log 2>&1
cp $R/gpurun_out/prof_$TAG/summary.txt $O/rocprofv3_summary_serialized.txt
cp $(find $R/gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv 2>/dev/null
timeout 300 python tools/pmc_traffic.py $R/gpurun_out/prof_$TAG > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp; export TMPDIR=/tmp; D=$(mktemp -d)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > $O/step_budget_bench.log 2>&1
cd $R; timeout 300 python tools/step_budget.py $(find $D -name "bench_kernel_trace.csv") --json $O/step_budget.json > $O/step_budget_default_command.txt 2>&1; rm -rf $D
cp $O/step_budget.json profiles/step_budget.json   # (so that the bench line below reads the budget of THIS library: step_budget.stale = false)
T0=$SECONDS; (timeout 900 python bench.py 2>/dev/null | tail -1) > $O/bench_default_command.json; echo "python bench.py (default command, every secondary + cpu baseline): $((SECONDS - T0)) s wall" > $O/bench_time.txt
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
tools/kernel_table.sh $O/naf_bf16_kernels.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2
tools/kernel_table.sh $O/head256_kernels.txt 5 python $R/tools/head_probe.py
tools/kernel_table.sh $O/restormer_balanced_kernels.txt 5 python $R/bench_extra.py --workload restormer --steps 3 --warmup 2
rm -f $O/*.trace.csv.gz
(timeout 300 python tools/head_probe.py --steps 12 2>&1 | tail -1) > $O/head_probe_time.txt
timeout 1500 tools/extras_all.sh gpurun_out/evidence_$TAG > $O/extras.log 2>&1
(timeout 400 python bench_extra.py --workload dcpt --dtype bf16_edge32 --size 256 2>&1 | tail -1) > $O/extra_dcpt256_bf16_edge32.json
(timeout 400 python bench_extra.py --workload restormer 2>&1 | tail -1) > $O/extra_restormer_balanced.json
(timeout 400 python tests/stream_stress.py --reps 20 2>&1 | grep "differing\|FAILED\|stable") > $O/stream_stress.txt
(timeout 500 python tests/fuzz_shapes.py --seed 11 --n 30 2>&1 | tail -5) > $O/fuzz_seed11_tail.txt
find $R/gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
tail -3 $O/pytest_gpu_full.log; cut -c1-400 $O/bench_default_command.json; cat $O/bench_time.txt $O/head_probe_time.txt
