#!/bin/bash
# per-kernel table (rocprofv3 --kernel-trace) of any command:  tools/kernel_table.sh <out.txt> <steps> <command...>
R=$PWD; OUT=$1; STEPS=$2; shift 2
mkdir -p $(dirname $OUT); cd /tmp; export TMPDIR=/tmp
D=$(mktemp -d)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- "$@" > $D/log.txt 2>&1
python $R/tools/kstats.py $(find $D -name "*kernel_trace.csv") $STEPS 80 > $OUT
tail -2 $D/log.txt >> $OUT
gzip -c $(find $D -name "*kernel_trace.csv" | head -1) > ${OUT%.txt}.trace.csv.gz 2>/dev/null; rm -rf $D
