#!/bin/bash
# Round-4 verdict, item 2: "spend fewer instructions per MFMA, then measure whether the clock really drops".  One box call:
#   the lean exact-fp32 loop of tools/ubench/ws_gemm_f32 (whole GEMM and k-loop alone) and the product's level-3 GEMM kernels, each with
#   MFMA-busy share, effective clock (GRBM_GUI_ACTIVE / duration) and instructions per MFMA.      tools/fp32_power_ceiling.sh <out.txt>
R=$PWD; OUT=$1; mkdir -p $(dirname $OUT)
C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
{
  echo "# un-profiled"; $R/tools/ubench/ws_gemm_f32
  for c in 0 1 2 3; do
    echo "# ws_gemm_f32 configuration $c (0: M 32768 N 512 whole GEMM, 1: N 1024, 2: N 512 k-loop alone x16, 3: N 1024 k-loop alone x8)"
    $R/tools/pmc_cmd.sh "$C1" ws_gemm $R/tools/ubench/ws_gemm_f32 $c
    $R/tools/pmc_cmd.sh "$C2" ws_gemm $R/tools/ubench/ws_gemm_f32 $c
  done
  echo "# product kernels: level-3 fp32 NAFBlock forward + backward (tools/level_trace.py 3; SQ_INSTS_VALU counts the MFMAs too)"
  $R/tools/pmc_cmd.sh "$C1" gemm_ $(which python) $R/tools/level_trace.py 3
  $R/tools/pmc_cmd.sh "$C2" gemm_ $(which python) $R/tools/level_trace.py 3
} > $OUT 2>&1
cat $OUT
