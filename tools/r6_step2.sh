#!/bin/bash
R=$PWD; O=$R/gpurun_out/${1:-r6_step2}; mkdir -p $O
(timeout 600 python tools/trace_shapes.py 2>&1 | tail -20) | tee $O/trace_shapes.txt
(timeout 900 python -m pytest tests/test_gpu_dcpt_step.py tests/test_gpu_dchead.py -q -m gpu -x 2>&1 | tail -8) | tee $O/pytest_dcpt.log
(timeout 600 python bench_extra.py --workload dcpt --dtype bf16 --size 64 --batch 4 --gpus 2 --path-check-shared-device --steps 2 --warmup 1 2>&1 | tail -3) | tee $O/extra_dcpt_2rank_pathcheck.txt
(timeout 600 python bench_extra.py --workload restormer --size 32 --batch 2 --gpus 2 --path-check-shared-device --steps 2 --warmup 1 2>&1 | tail -3) | tee $O/extra_restormer_2rank_pathcheck.txt
T0=$SECONDS; (timeout 1200 python bench.py 2>$O/bench_err.txt | tail -1) > $O/bench_default.json; echo "bench.py default: $((SECONDS - T0)) s wall" | tee $O/bench_time.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_step2/bench_default.json"))
print(d["ms_per_step"], d["value"])
for k,v in d.get("secondary",{}).items(): print(k, v.get("ms_per_step", v.get("ms_per_image")), (v.get("roofline") or {}).get("kernel"))
PY
tail -3 $O/bench_err.txt
