#!/bin/bash
# Diagnostic variant of the library (never shipped; dcpt_amd/lib is git-ignored):
#   tools/build_variant.sh <suffix> "<sources to recompile>" <extra hipcc flags...>  ->  dcpt_amd/lib/libdcpt_hip_<suffix>.so
# the other objects are taken from the product build (python -m dcpt_amd.build).
set -e
cd "$(dirname "$0")/.."
SUF=$1; RE="$2"; shift 2
python -m dcpt_amd.build > /dev/null
mkdir -p dcpt_amd/build/$SUF
OBJS=""
for s in $(python -c "from dcpt_amd.build import SOURCES; print(' '.join(SOURCES))"); do
  if [[ " $RE " == *" $s "* ]]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-value "$@" -c dcpt_amd/csrc/$s -o dcpt_amd/build/$SUF/${s%.hip}.o &
    OBJS="$OBJS dcpt_amd/build/$SUF/${s%.hip}.o"
  else
    OBJS="$OBJS dcpt_amd/build/${s%.hip}.o"
  fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o dcpt_amd/lib/libdcpt_hip_$SUF.so
echo dcpt_amd/lib/libdcpt_hip_$SUF.so
