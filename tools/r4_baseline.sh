#!/bin/bash
# round-4 start-of-round evidence on one box: changed tests, per-kernel tables of the bf16 steps (two-stream and serialized), Restormer
TAG=${1:-r4_base}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py tests/test_gpu_dcpt_step.py -q -x 2>&1 | tail -5) > $O/pytest_changed.log 2>&1
tools/kernel_table.sh $O/naf_bf16_kernels.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2
tools/kernel_table.sh $O/naf_bf16_kernels_serialized.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2 --side-stream 0
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels_serialized.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2 --side-stream 0
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
tools/kernel_table.sh $O/restormer_balanced_kernels.txt 5 python $R/bench_extra.py --workload restormer --steps 3 --warmup 2
for a in "naf --dtype bf16" "naf --dtype bf16 --side-stream 0" "dcpt --dtype bf16" "dcpt --dtype bf16 --size 256" "restormer"; do
  (timeout 300 python bench_extra.py --workload $a 2>&1 | tail -1 | cut -c1-600) >> $O/extras.txt
done
cat $O/pytest_changed.log; cat $O/extras.txt
