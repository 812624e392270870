import json, sys
r = json.load(open(sys.argv[1]))
rf = r["roofline"]
print(r["value"], r["ms_per_step"], rf.get("all_gemm_ms_per_step"), rf.get("all_gemm_tflops"), r.get("cpu_baseline", {}).get("value"),
      "| top:", rf.get("kernel"), rf.get("achieved"), rf.get("frac"), "as_run", rf.get("as_run"), "whole", rf.get("whole_step"))
for k in rf.get("by_kernel", []):
    print(k)
