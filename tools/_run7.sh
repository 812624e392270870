R=$PWD; O=$R/gpurun_out/r4_ddp; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_configs.py -q -k "ddp" 2>&1 | tail -25) | tee $O/t.log
(timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1) > $O/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_ddp/bench.json"))
print(d["ms_per_step"], d["ms_fwd_bwd"]); print(d["secondary"].get("ddp_wrap_one_rank"))
PY
