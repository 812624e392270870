#!/bin/bash
# same-box A/B: this tree's library against the last commit's (experiments/lib/libdcpt_hip_head.so) on the bf16 steps + the tests of the bf16 path
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ln_bwd_ab; mkdir -p $O
(for i in 1 2; do for v in new head; do
    L=""; [ $v = head ] && L=$R/experiments/lib/libdcpt_hip_head.so
    for wl in "dcpt --size 256" "dcpt" "naf"; do
      echo -n "$v $wl: "; DCPT_TOOL_LIB=$L timeout 300 python tools/bench_extra_variant.py --workload $wl --dtype bf16 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
    done
done; done) 2>&1 | tee $O/step_ab.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_dchead.py tests/test_gpu_dcpt_step.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
