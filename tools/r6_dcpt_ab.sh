#!/bin/bash
# same-box A/B of the head's levers on the whole DCPT bf16 step (variant build with the tuning switches)
R=$PWD; O=$R/gpurun_out/${1:-r6_dcpt_ab}; mkdir -p $O
for rep in 1 2; do for sz in 256 128; do for cfg in "1 1" "1 0" "0 1" "0 0"; do set -- $cfg
  echo -n "size $sz LN epilogues=$1 head side stream=$2: "
  DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_headtune.so DCPT_HEAD_LN_EPI=$1 DCPT_HEAD_SIDE=$2 timeout 300 python tools/bench_extra_variant.py --workload dcpt --dtype bf16 --size $sz --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms')"
done; done; done | tee $O/dcpt_head_ab.txt
