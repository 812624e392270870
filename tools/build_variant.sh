#!/bin/bash
# Diagnostic variant of the library (never shipped; experiments/ is git-ignored; delete it before the last push of a round):
#   tools/build_variant.sh <suffix> "<sources to recompile>" <extra hipcc flags...>  ->  experiments/lib/libdcpt_hip_<suffix>.so   (use: DCPT_TOOL_LIB=<that path> python tools/<tool>.py)
# the other objects are taken from the product build (python -m dcpt_amd.build).  Recompiled sources get -DDCPT_TUNING: only there do the
# DCPT_* environment switches (dcpt_common.h dcpt_tuning) exist.
set -e
cd "$(dirname "$0")/.."
SUF=$1; RE="$2"; shift 2
python -m dcpt_amd.build > /dev/null
mkdir -p experiments/build/$SUF experiments/lib
OBJS=""
for s in $(python -c "from dcpt_amd.build import SOURCES; print(' '.join(SOURCES))"); do
  if [[ " $RE " == *" $s "* ]]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-value -Xclang -target-feature -Xclang -packed-fp32-ops -DDCPT_TUNING "$@" -c dcpt_amd/csrc/$s -o experiments/build/$SUF/${s%.hip}.o &
    OBJS="$OBJS experiments/build/$SUF/${s%.hip}.o"
  else
    OBJS="$OBJS dcpt_amd/build/${s%.hip}.o"
  fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o experiments/lib/libdcpt_hip_$SUF.so
echo experiments/lib/libdcpt_hip_$SUF.so
