#!/bin/bash
# same-box A/B of the head's cached conv packs (ABI 14, DCPT_CONV_PACK_CACHE=0: per-call packs as before) + the tests that cover the head
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/conv_pack_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_dchead.py tests/test_gpu_dcpt_step.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
  for v in 1 0; do
    for wl in "dcpt" "dcpt --size 256"; do
      DCPT_CONV_PACK_CACHE=$v timeout 300 python bench_extra.py --workload $wl --dtype bf16 2>/dev/null | python -c "import sys,json; print('cache=$v $wl', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
    done
  done
done 2>&1 | tee $O/step_ab.txt
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
grep -n "wpack\|total kernel" $O/dcpt_allbf16_128_kernels.txt | cut -c1-150
