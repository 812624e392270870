#!/bin/bash
# device idle per steady-state step of every timed workload (kernel traces + tools/idle_gaps.py)
R=$PWD; O=$R/gpurun_out/${1:-r6_idle}; mkdir -p $O
run() { name=$1; marker=$2; shift 2; tools/kernel_table.sh $O/$name.txt 6 "$@"; echo "== $name"; python tools/idle_gaps.py $O/$name.trace.csv.gz $marker | tee $O/$name.idle.txt; }
run headline_fp32 adamw_kernel python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-prof
run naf_bf16 adamw_kernel python $R/bench_extra.py --workload naf --dtype bf16 --steps 5 --warmup 2
run restormer adamw_kernel python $R/bench_extra.py --workload restormer --steps 4 --warmup 2
run infer2k_bf16 conv3x3_b2s python $R/bench_extra.py --workload infer2k --dtype bf16 --steps 4 --warmup 2
run infer2k_fp32 conv3x3_b2s python $R/bench_extra.py --workload infer2k --dtype fp32 --steps 4 --warmup 2
rm -f $O/*.trace.csv.gz
