#!/bin/bash
# TIMING-ONLY experiment (LABNOTES 7): what a side-stream join that lags one backward call would buy.  experiments/lib/libdcpt_hip_lag.so = the
# product objects + a copy of side.hip whose join waits for the PREVIOUS call's side work (DCPT_SIDE_LAG=1) or for nothing (=2: the bound);
# the results of those runs are garbage (workspace / freed saved tensors race) -- only the step times mean anything.
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/side_lag; mkdir -p $O
(for i in 1 2; do for lag in 0 1 2; do for wl in "naf --dtype fp32" "naf --dtype bf16"; do
   echo -n "lag=$lag $wl: "; DCPT_SIDE_LAG=$lag DCPT_TOOL_LIB=$R/experiments/lib/libdcpt_hip_lag.so timeout 300 python tools/bench_extra_variant.py --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms')"
 done; done; done) 2>&1 | tee $O/side_lag.txt
