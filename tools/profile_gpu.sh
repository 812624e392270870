#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the benchmark, summaries into gpurun_out/prof_$1
# usage: tools/profile_gpu.sh <tag>
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --side-stream 0"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -o bench -- $BENCH > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
python tools/prof_summary.py $OUT auto > $OUT/summary.txt 2>&1   # (steps = intro-conv launches in the trace)
tail -60 $OUT/summary.txt
# keep only the small files (the raw traces are big)
find $OUT -name "*.csv" -size +20M -delete
