R=$PWD; O=$R/gpurun_out/r4_dw; mkdir -p $O
tools/build_variant.sh dw "dwring.hip" > $O/build.log 2>&1
L=$R/experiments/lib/libdcpt_hip_dw.so
for lv in 3 4 2; do
for cfg in "0 0" "32 0" "16 0" "16 2" "16 1" "32 2" "8 0" "8 2"; do
  set -- $cfg
  DCPT_TOOL_LIB=$L DCPT_DWR_FWD_LP=$1 DCPT_DWR_FWD_NRP=$2 tools/level_kernels.sh $lv bf16 $O/tmp.txt; echo "level $lv LP=$1 NRP=$2: $(grep dwr_gate_fwd $O/tmp.txt | cut -c1-75)"
done; done
for cfg in "0 0" "1 0" "2 0" "4 0"; do set -- $cfg
  DCPT_TOOL_LIB=$L DCPT_DWR_BWD_NRP=$1 tools/level_kernels.sh 3 bf16 $O/tmp.txt; echo "level 3 bwd NRP=$1: $(grep dwr_bwd $O/tmp.txt | cut -c1-75)"
done
for lpb in 8 32; do
  DCPT_TOOL_LIB=$L DCPT_DWR_BWD_LP=$lpb tools/level_kernels.sh 3 bf16 $O/tmp.txt; echo "level 3 bwd LP=$lpb: $(grep dwr_bwd $O/tmp.txt | cut -c1-75)"
done
