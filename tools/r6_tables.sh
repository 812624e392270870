#!/bin/bash
# steady-state kernel tables (+ device idle) of the bf16 steps
R=$PWD; O=$R/gpurun_out/${1:-r6_tables}; mkdir -p $O
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
tools/kernel_table.sh $O/head256_kernels.txt 5 python $R/tools/head_probe.py
for f in dcpt_allbf16_256 dcpt_allbf16_128; do echo "== $f"; head -12 $O/${f}_kernels.txt | cut -c1-200; grep -n "at::native\|rocclr\|Fill" $O/${f}_kernels.txt | cut -c1-160; done
