#!/bin/bash
# the default bench line of the round's final tree (library unchanged since the evidence run: same digest; host side: the loss log no longer synchronises)
R=$PWD; O=$R/gpurun_out/${1:-r6_final}; mkdir -p $O
T0=$SECONDS; (timeout 900 python bench.py 2>/dev/null | tail -1) > $O/bench_default_command.json; echo "python bench.py (default command, every secondary + cpu baseline): $((SECONDS - T0)) s wall" > $O/bench_time.txt
(timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_dcpt_step.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -3) > $O/pytest_models.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_final/bench_default_command.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic_stale"], d["roofline"]["step_budget"]["stale"], d["lib_digest"][:8])
for k,v in d.get("secondary",{}).items(): print(k, v.get("ms_per_step", v.get("ms_per_image")))
PY
cat $O/bench_time.txt; tail -2 $O/pytest_models.log
