#!/bin/bash
R=$PWD; O=$R/gpurun_out/${1:-r6_tall2}; mkdir -p $O
for rep in 1 2 3; do for cfg in "1 1" "1 0" "0 0"; do set -- $cfg
  echo -n "tall tile=$1 (LN-backward epilogue on it=$2): "; DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_headtune.so DCPT_NT_TALL=$1 DCPT_NT_TALL_BWD=$2 timeout 300 python tools/head_probe.py --steps 12 2>&1 | tail -1
done; done | tee $O/head_tall_ab.txt
for cfg in "1 1" "0 0"; do set -- $cfg
DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_headtune.so DCPT_NT_TALL=$1 DCPT_NT_TALL_BWD=$2 tools/kernel_table.sh $O/head256_kernels_tall$1.txt 5 python $R/tools/head_probe.py
grep "gemm_nt_bf16" $O/head256_kernels_tall$1.txt | head -12 | cut -c1-140
done
