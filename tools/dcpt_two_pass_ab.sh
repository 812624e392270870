#!/bin/bash
# same-box A/B of the DCPT step's encoder passes: one pass over the stacked 2B batch (default) against the reference's two passes of B
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/two_pass_ab; mkdir -p $O
(for i in 1 2; do for tp in "" "--two-pass"; do for sz in 128 256; do
   echo -n "dcpt bf16 $sz ${tp:-batched}: "; timeout 300 python bench_extra.py --workload dcpt --dtype bf16 --size $sz $tp 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['peak_mem_gb'], 'GB')"
 done; done; done) 2>&1 | tee $O/dcpt_two_pass_ab.txt
