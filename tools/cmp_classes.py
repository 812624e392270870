import json,sys
rows={}
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"]["all_gemm_ms_per_step"], d["roofline"]["all_gemm_tflops"])
    for k in d["roofline"]["by_kernel"]:
        rows.setdefault((k["kernel"],tuple(k["MNK"])),{})[f]=k["tflops"]
for k,v in rows.items():
    print(f"{k[0]:36s} {str(k[1]):24s}", "  ".join(f"{v.get(f,0):7.1f}" for f in sys.argv[1:]))
