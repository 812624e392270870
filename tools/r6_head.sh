#!/bin/bash
# head work of round 6: the head's parity tests, then the head probe (timing + per-kernel table)
TAG=${1:-r6_head}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dchead.py tests/test_gpu_bf16.py tests/test_gpu_dcpt_step.py -q -m gpu -x 2>&1 | tail -15) > $O/pytest_head.log 2>&1
tail -15 $O/pytest_head.log
(timeout 300 python tools/head_probe.py --steps 12 2>&1 | tail -2) > $O/head_time.txt; cat $O/head_time.txt
tools/kernel_table.sh $O/head256_kernels.txt 5 python $R/tools/head_probe.py
head -30 $O/head256_kernels.txt | cut -c1-150
