R=$PWD; O=$R/gpurun_out/r4_t15; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_dchead.py -q -x 2>&1 | tail -3)
tools/kernel_table.sh $O/d256.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 4 --warmup 2 --side-stream 0
grep "ln_bwd_bf16\|ln_fwd_bf16" $O/d256.txt | cut -c1-90; grep -o 'ms_per_step": [0-9.]*' $O/d256.txt
tools/kernel_table.sh $O/naf.txt 8 python $R/bench_extra.py --workload naf --dtype bf16 --steps 6 --warmup 2 --side-stream 0
grep "ln_bwd_bf16\|ln_fwd_bf16" $O/naf.txt | cut -c1-90; grep -o 'ms_per_step": [0-9.]*' $O/naf.txt
