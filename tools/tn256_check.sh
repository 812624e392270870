#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/tn256_check; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "wgrad or nafblock or block" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
DCPT_TOOL_LIB=$R/experiments/lib/libdcpt_hip_tl256.so timeout 200 python tools/tn256_timeline.py 3 2>&1 | grep -v "^conv. range\|amdgpu.ids" > $O/timeline_l3.txt
cat $O/timeline_l3.txt
timeout 400 bash tools/pmc_kernel.sh 3 bf16 gemm_tn_bf16_256 "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" > $O/pmc.txt 2>&1; cat $O/pmc.txt
for B in 32 64; do B=$B timeout 300 bash tools/level_kernels.sh 3 bf16 $O/l3_b$B.txt; grep "gemm_tn_bf16_256\|wgrad_finish\|total" $O/l3_b$B.txt; done
timeout 200 python tools/tn256_probe.py 2>&1 | grep -v amdgpu > $O/tn_probe.txt; cat $O/tn_probe.txt
