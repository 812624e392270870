#!/bin/bash
# per-kernel table (rocprofv3 --kernel-trace) of any command:  tools/kernel_table.sh <out.txt> <steps> <command...>
R=$PWD; OUT=$1; STEPS=$2; shift 2
mkdir -p $(dirname $OUT); cd /tmp; export TMPDIR=/tmp
D=$(mktemp -d)
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- "$@" > $D/log.txt 2>&1
python $R/tools/kstats.py $(find $D -name "*kernel_trace.csv") $STEPS 60 > $OUT
tail -2 $D/log.txt >> $OUT
rm -rf $D
