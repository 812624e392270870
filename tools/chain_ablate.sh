#!/bin/bash
# The chain kernel's time per mode and per compile-time ablation (chain_bf16.hip CHAIN_ABL_*), one box call:
#   tools/chain_ablate.sh <out.txt> [level]     (variants built beforehand: for v in base now nomfma nolds nost; do tools/build_variant.sh chain_$v ...)
R=$PWD; OUT=$1; L=${2:-3}; cd /tmp; export TMPDIR=/tmp
mkdir -p $(dirname $OUT); : > $OUT
for v in base now nomfma nolds nost nowmfma; do
  [ -f $R/experiments/lib/libdcpt_hip_chain_$v.so ] || continue
  for mode in train infer; do
    D=$(mktemp -d)
    DCPT_TOOL_LIB=$R/experiments/lib/libdcpt_hip_chain_$v.so rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/chain_trace.py $L $mode > /dev/null 2>&1
    echo "$v $mode: $(python $R/tools/kstats.py $(find $D -name "*kernel_trace.csv") 1 40 | grep chain_fwd)" >> $OUT
    rm -rf $D
  done
done
cat $OUT
