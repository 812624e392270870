#!/bin/bash
R=$PWD; O=$R/gpurun_out/${1:-r6_lazylog}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dcpt_step.py tests/test_gpu_cli.py -q -m gpu -x 2>&1 | tail -4) | tee $O/pytest.log
for sz in 256 128; do for rep in 1 2; do
  echo -n "dcpt bf16 $sz: "; timeout 300 python bench_extra.py --workload dcpt --dtype bf16 --size $sz --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['log'])"
done; done | tee $O/dcpt_times.txt
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
head -3 $O/dcpt_allbf16_256_kernels.txt | cut -c1-250; head -3 $O/dcpt_allbf16_128_kernels.txt | cut -c1-250
