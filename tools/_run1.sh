O=gpurun_out/r4_tn1; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_bf16.py -q -x -k "conv1x1_wgrad" 2>&1 | tail -15) > $O/t1.log
cat $O/t1.log
(timeout 300 python tools/tn256_probe.py 2>&1 | tail -10) | tee $O/probe.txt
(timeout 900 python -m pytest tests/test_gpu_bf16.py -q -x 2>&1 | tail -8) | tee $O/t2.log
tools/level_kernels.sh 3 bf16 $O/bf16_block_level3_kernels.txt; cat $O/bf16_block_level3_kernels.txt | cut -c1-150
(timeout 300 python bench_extra.py --workload naf --dtype bf16 2>&1 | tail -1 | cut -c1-300) | tee $O/naf.txt
