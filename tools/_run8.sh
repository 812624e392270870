R=$PWD; O=$R/gpurun_out/r4_head; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_dchead.py -q -x 2>&1 | tail -8) | tee $O/t.log
tools/kernel_table.sh $O/dcpt_allbf16_256_kernels_serialized.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2 --side-stream 0
tools/kernel_table.sh $O/dcpt_allbf16_128_kernels_serialized.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2 --side-stream 0
head -36 $O/dcpt_allbf16_256_kernels_serialized.txt | cut -c1-150
head -30 $O/dcpt_allbf16_128_kernels_serialized.txt | cut -c1-150
for a in "dcpt --dtype bf16" "dcpt --dtype bf16 --size 256" "naf --dtype bf16"; do
  (timeout 300 python bench_extra.py --workload $a 2>&1 | tail -1 | cut -c1-250) | tee -a $O/extras.txt
done
