#!/bin/bash
# every secondary workload of BASELINE.json on one box: JSON lines into <outdir>
OUT=$PWD/${1:-gpurun_out/extras}; mkdir -p $OUT
run() { name=$1; shift; (timeout 400 python bench_extra.py "$@" 2>&1 | tail -1) > $OUT/extra_$name.json; cat $OUT/extra_$name.json | cut -c1-400; }
run naf_fp32 --workload naf --dtype fp32
run naf_bf16 --workload naf --dtype bf16
run dcpt_fp32 --workload dcpt --dtype fp32
run dcpt_allbf16 --workload dcpt --dtype bf16
run dcpt256_fp32 --workload dcpt --dtype fp32 --size 256
run dcpt256_allbf16 --workload dcpt --dtype bf16 --size 256
run infer2k --workload infer2k --dtype fp32
run infer2k_bf16 --workload infer2k --dtype bf16
