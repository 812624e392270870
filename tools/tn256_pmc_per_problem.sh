#!/bin/bash
# experiment 5: fabric reads of the grouped weight-gradient launch per problem (only one problem's blocks run; same geometry)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/tn256_pmc; mkdir -p $O; rm -f $O/pmc.txt
export DCPT_TOOL_LIB=$R/experiments/lib/libdcpt_hip_tune.so
for P in 0 1 2 3 -1; do
  echo "== only problem $P (0 conv5, 1 conv4, 2 conv3, 3 conv1; -1 all)" >> $O/pmc.txt
  DCPT_TN_ONLY=$P timeout 400 bash tools/pmc_kernel.sh 3 bf16 gemm_tn_bf16_256 "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" >> $O/pmc.txt 2>&1
done
cat $O/pmc.txt
