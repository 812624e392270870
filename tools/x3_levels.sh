#!/bin/bash
# bf16x3 mode per level: level probe + per-kernel tables of one block at levels 1..4 (one gpurun call)
O=$PWD/gpurun_out/x3_levels; mkdir -p $O
(timeout 300 python tools/level_probe.py x3 2>&1 | tail -6) > $O/level_probe_x3.txt
for l in 1 2 3 4; do tools/level_kernels.sh $l x3 $O/x3_block_level${l}_kernels.txt; done
cat $O/level_probe_x3.txt
