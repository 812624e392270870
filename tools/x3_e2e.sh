#!/bin/bash
# end-to-end check of the bf16x3 mode on the GPU box: fp32 parity suite with the mode forced on every eligible launch, level probe, bench
OUT=$PWD/${1:-gpurun_out/x3}; mkdir -p $OUT
(timeout 1500 python -c "
import sys, pytest
sys.path.insert(0, '.')
from dcpt_amd import functional as DF
DF.set_gemm_precision('bf16x3', min_tiles=1)
sys.exit(pytest.main(['tests/test_gpu_parity.py', 'tests/test_gpu_dcpt_step.py', 'tests/test_gpu_dchead.py', '-x', '-q', '-m', 'gpu']))
" 2>&1 | tail -12) > $OUT/parity_forced.log 2>&1
tail -4 $OUT/parity_forced.log
for m in fp32 bf16x3; do (timeout 300 python bench.py --no-cpu-baseline --no-secondary --gemm-precision $m 2>&1 | tail -1) > $OUT/bench_$m.json; done
python - <<PY
import json
for m in ("fp32","bf16x3"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%m).read().strip().splitlines()[-1]); print(m, d["ms_per_step"], d["value"], d["config"]["loss"], d["roofline"]["all_gemm_tflops"])
    except Exception as e: print(m, "ERR", e, open("$OUT/bench_%s.json"%m).read()[-600:])
PY
