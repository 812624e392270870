import json,sys
r=json.load(open(sys.argv[1])); print(r["value"], r["ms_per_step"], r["roofline"]["all_gemm_ms_per_step"], r["roofline"]["all_gemm_tflops"], r.get("cpu_baseline",{}).get("value"))
for k in r["roofline"]["by_kernel"]: print(k)
