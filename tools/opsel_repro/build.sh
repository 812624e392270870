#!/bin/bash
# Code objects of the ending conv WITH packed fp32 (what the library shipped until round 4) and with subsets of its packed instructions
# rewritten as scalar ones in the assembly -- same registers, same order, nothing else touched:
#   p0_orig          as compiled
#   u0_all_scalar    all 38 packed-fp32 instructions of the row loop scalar
#   w1_..opsel       only the 24 with op_sel / op_sel_hi scalar        <- cures it
#   w2_..plain       only the 14 without scalar                         <- does not
#   q2_nops          s_nop 3 after EVERY instruction of the loop        <- does not (not a wait-state problem)
# then: python tools/opsel_repro/run.py   (GPU box; prints launches that differ from the quiet result next to bf16 GEMM load)
set -e
cd "$(dirname "$0")"; R=$PWD/../..; W=${TMPDIR:-/tmp}/opsel_repro; mkdir -p $W
hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Wno-unused-function -Wno-unused-variable -Wno-unused-value -I$R/dcpt_amd/csrc $R/dcpt_amd/csrc/conv3x3.hip -o $W/orig.s 2>/dev/null
python scalarize.py $W p0_orig:0:0 u0_all_scalar:1:999 w1_scalar_only_opsel_forms:opsel w2_scalar_only_plain_forms:plain q2_nop_after_every_instr:nops
ls -la *.hsaco
