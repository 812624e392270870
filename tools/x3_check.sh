#!/bin/bash
# bf16x3 kernels after a change: its own tests, the fp32 parity suite with the mode forced, the level-3 block table, the bench line in both modes
O=$PWD/gpurun_out/x3_check; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -m gpu 2>&1 | tail -4) > $O/tests_x3.log; cat $O/tests_x3.log
tools/level_kernels.sh 3 x3 $O/x3_block_level3_kernels.txt; grep "x3_kernel" $O/x3_block_level3_kernels.txt | cut -c1-120
bash tools/x3_e2e.sh gpurun_out/x3_check
