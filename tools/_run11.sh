R=$PWD; O=$R/gpurun_out/r4_t11; mkdir -p $O
tools/kernel_table.sh $O/dcpt128.txt 6 python $R/bench_extra.py --workload dcpt --dtype bf16 --steps 4 --warmup 2
python - <<'PY'
import re
tot=0
for ln in open("gpurun_out/r4_t11/dcpt128.txt"):
    m=re.search(r"n/step=\s*([\d.]+)", ln)
    if m and "FillFunctor" not in ln and "copyBuffer" not in ln: tot+=float(m.group(1))
print("launches/step (top 60 kernels, without init fills/copies):", tot)
PY
head -45 $O/dcpt128.txt | cut -c1-150
