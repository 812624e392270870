#!/bin/bash
R=$PWD; O=$R/gpurun_out/${1:-r6_tall}; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_dchead.py "tests/test_gpu_bf16.py::test_conv_ln_bf16_oracle" tests/test_gpu_bf16.py::test_dc_head_bf16_oracle -q -m gpu 2>&1 | tail -15) | tee $O/pytest.log
for rep in 1 2; do for tall in 1 0; do
  echo -n "tall tile=$tall: "; DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_headtune.so DCPT_NT_TALL=$tall timeout 300 python tools/head_probe.py --steps 12 2>&1 | tail -1
done; done | tee $O/head_tall_ab.txt
tools/kernel_table.sh $O/head256_kernels.txt 5 python $R/tools/head_probe.py
head -24 $O/head256_kernels.txt | cut -c1-150
