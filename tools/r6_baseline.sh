#!/bin/bash
# Round-6 baseline on today's box: GPU suite, default bench line, kernel tables of the head alone and of the DCPT 256 step.
TAG=${1:-r6_base}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
(timeout 1200 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6) > $O/pytest_gpu.log 2>&1
(timeout 900 python bench.py 2>/dev/null | tail -1) > $O/bench_default.json
tools/kernel_table.sh $O/head256_kernels.txt 5 python $R/tools/head_probe.py
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/naf_bf16_kernels.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2
tail -3 $O/pytest_gpu.log; cut -c1-600 $O/bench_default.json; head -3 $O/head256_kernels.txt
