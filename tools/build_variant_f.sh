#!/bin/bash
# Experiment build of ONE translation unit: tools/build_variant_f.sh NAME FILE.hip [-DFLAG ...]
#   -> build/lib_NAME.so = the current objects with FILE.hip recompiled with the flags.
# A/B on one GPU box: gpurun -- STAGES=substr bash tools/gpu_lease.sh TAG stage_ab:build/lib_A.so,build/lib_B.so
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../rav1e_amd/csrc"
mkdir -p ../../build
BASE=$(basename $FILE .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed "$@" -c $FILE -o ../../build/${BASE}_$NAME.o
OBJS=$(ls *.o | grep -v "^${BASE}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib_$NAME.so $OBJS ../../build/${BASE}_$NAME.o -ldl
ls -la ../../build/lib_$NAME.so
