#!/bin/bash
# A/B of the head's two levers (variant build with the tuning switches): LayerNorm in the GEMM epilogues x weight gradients on the side stream
R=$PWD; O=$R/gpurun_out/${1:-r6_head_ab}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dchead.py tests/test_gpu_dcpt_step.py -q -m gpu -x 2>&1 | tail -15) > $O/pytest_head.log 2>&1
tail -5 $O/pytest_head.log
for rep in 1 2; do for epi in 1 0; do for side in 1 0; do
  echo -n "LN epilogues=$epi side stream=$side: "
  DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_headtune.so DCPT_HEAD_LN_EPI=$epi DCPT_HEAD_SIDE=$side timeout 300 python tools/head_probe.py --steps 12 2>&1 | tail -1
done; done; done | tee $O/head_ab.txt
