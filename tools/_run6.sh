R=$PWD; O=$R/gpurun_out/r4_full1; mkdir -p $O
(timeout 1700 python -m pytest tests/ -q -m gpu 2>&1 | tail -25) | tee $O/pytest_gpu_full.log
(timeout 900 python bench.py 2>&1 | tail -1) > $O/bench_default.json; cut -c1-1500 $O/bench_default.json
