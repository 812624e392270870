#!/bin/bash
# fabric traffic (FETCH_SIZE, WRITE_SIZE: KiB per launch, FETCH x 2 on gfx950) and kernel times of EVERY kernel of one NAFBlock forward + backward,
# per level and storage type:   tools/traffic_tables.sh "<level> <fp32|bf16>" ...   -> gpurun_out/traffic/
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/traffic; mkdir -p $O
for cfg in "$@"; do set -- $cfg
  timeout 400 bash tools/pmc_kernel.sh $1 $2 "" "FETCH_SIZE" "WRITE_SIZE" 2>&1 | grep -v "at::native\|rocclr" > $O/pmc_l$1_$2.txt
  timeout 300 bash tools/level_kernels.sh $1 $2 $O/kernels_l$1_$2.txt
  python - $O/pmc_l$1_$2.txt $O/kernels_l$1_$2.txt <<'PY'
import re, sys
f, w = {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"(.*) launches (\d+) (FETCH_SIZE|WRITE_SIZE)=([\d.e+]+)", line.strip())
    if m:
        (f if m.group(3) == "FETCH_SIZE" else w)[m.group(1).replace("void ", "")] = float(m.group(4)) * 1024 / 1e6
t = {}
for line in open(sys.argv[2]):
    m = re.search(r"n/step=\s*([\d.]+) avg=\s*([\d.]+)us\s+(?:void )?(.*?)\(", line)
    if m: t[m.group(3)] = (float(m.group(1)), float(m.group(2)))
print(f"== {sys.argv[1]}")
for k in sorted(f, key=lambda k: -(f[k] * 2 + w.get(k, 0))):
    n, us = t.get(k, (0, 0))
    tot = f[k] * 2 + w.get(k, 0)
    print(f"{k[:58]:58s} n={n:4.0f} {us:7.1f} us  read {f[k]*2:7.1f} MB  write {w.get(k,0):7.1f} MB" + (f"  {tot/us/1e6*1e6/1e6:5.2f} TB/s" if us else ""))
PY
done
