#!/bin/bash
# fp32 NT GEMMs: epilogue operands trickled in under the main loop (product) vs loaded in the epilogue (variant built with -DDCPT_NT_PREFETCH=0)
R=$PWD; O=$R/gpurun_out/${1:-r6_pf_ab}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not bf16x3" 2>&1 | tail -4) | tee $O/pytest_parity.log
for rep in 1 2 3; do
  echo -n "prefetch on : "; timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [ (k['kernel'][8:], k['tflops']) for k in d['roofline']['by_kernel'][3:9]])"
  echo -n "prefetch off: "; DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_nopf.so timeout 300 python tools/bench_variant.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [ (k['kernel'][8:], k['tflops']) for k in d['roofline']['by_kernel'][3:9]])"
done | tee $O/pf_ab.txt
