#!/bin/bash
# per-kernel A/B of the LayerNorm backward (bf16) inside the DCPT 256 x 256 step: this tree's library against experiments/lib/libdcpt_hip_head.so
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ln_bwd_ab; mkdir -p $O
tools/kernel_table.sh $O/dcpt256_new.txt 6 python $R/tools/bench_extra_variant.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
DCPT_TOOL_LIB=$R/experiments/lib/libdcpt_hip_head.so tools/kernel_table.sh $O/dcpt256_head.txt 6 python $R/tools/bench_extra_variant.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
for v in new head; do echo "== $v"; head -1 $O/dcpt256_$v.txt; grep "ln_bwd_bf16\|ln_fwd_bf16" $O/dcpt256_$v.txt | cut -c1-150; done
