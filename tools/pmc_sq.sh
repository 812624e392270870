#!/bin/bash
# SQ wait breakdown of the kernels a command launches (one PMC pass, kernel-trace only)
# usage: tools/pmc_sq.sh <tag> <command...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -o x -- "$@" > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -o x -- "$@" > $OUT/p2.log 2>&1
cd $ROOT; python tools/pmc_sq_summary.py $OUT
for f in $OUT/p1.log $OUT/p2.log; do tail -n 3 $f | cut -c1-300; done
