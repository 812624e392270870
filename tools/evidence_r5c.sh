#!/bin/bash
# Round-5 evidence of the LAST library of the round (ABI 14: cached conv packs of the head) on ONE GPU box -- the parts of tools/evidence_r5b.sh
# that the change can move, most important first (the box budget left was ~25 minutes): full GPU suite, the default bench line, rocprofv3 summary +
# PMC traffic of the headline command, step budget, kernel tables of the bf16 steps, two-pass A/B of the DCPT step, fuzz + stream stress.
TAG=${1:-r5c}; R=$PWD; O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
(timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6) > $O/pytest_gpu_full.log 2>&1
tools/profile_gpu.sh $TAG > $O/profile_gpu.log 2>&1
cp $R/gpurun_out/prof_$TAG/summary.txt $O/rocprofv3_summary_serialized.txt
cp $(find $R/gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv 2>/dev/null
python tools/pmc_traffic.py $R/gpurun_out/prof_$TAG > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp; export TMPDIR=/tmp; D=$(mktemp -d)
rocprofv3 --kernel-trace --output-format csv -d $D -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > $O/step_budget_bench.log 2>&1
cd $R; python tools/step_budget.py $(find $D -name "bench_kernel_trace.csv") --json $O/step_budget.json > $O/step_budget_default_command.txt 2>&1; rm -rf $D
cp $O/step_budget.json profiles/step_budget.json   # (so that the bench line below reads the budget of THIS library: step_budget.stale = false)
T0=$SECONDS; (timeout 900 python bench.py 2>/dev/null | tail -1) > $O/bench_default_command.json; echo "python bench.py (default command, every secondary + cpu baseline): $((SECONDS - T0)) s wall" > $O/bench_time.txt
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
tools/kernel_table.sh $O/naf_bf16_kernels.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2
(for i in 1 2; do for tp in "" "--two-pass"; do for sz in 128 256; do
   echo -n "dcpt bf16 $sz ${tp:-batched}: "; timeout 300 python bench_extra.py --workload dcpt --dtype bf16 --size $sz $tp 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['peak_mem_gb'], 'GB')"
 done; done; done) > $O/dcpt_two_pass_ab.txt 2>&1
tools/extras_all.sh gpurun_out/evidence_$TAG > $O/extras.log 2>&1
(timeout 400 python tests/stream_stress.py --reps 20 2>&1 | grep "differing\|FAILED\|stable") > $O/stream_stress.txt
(timeout 500 python tests/fuzz_shapes.py --seed 8 --n 30 2>&1 | tail -5) > $O/fuzz_seed8_tail.txt
find $R/gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
tail -3 $O/pytest_gpu_full.log; cat $O/bench_default_command.json | cut -c1-400; cat $O/dcpt_two_pass_ab.txt
