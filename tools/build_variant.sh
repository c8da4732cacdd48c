#!/bin/bash
# Experiment builds of the headline kernel: tools/build_variant.sh NAME [-DFLAG ...]
#   -> build/lib_NAME.so = the current objects with rdo_cand.hip recompiled (headline
#      instantiations only, so that a variant builds in well under a minute) with the flags.
# A/B on one GPU box: gpurun -- bash tools/gpu_lease.sh TAG bench_ab:build/lib_A.so,build/lib_B.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rav1e_amd/csrc"
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed \
  -DR1_HEADLINE_ONLY "$@" -c rdo_cand.hip -o ../../build/rdo_cand_$NAME.o
OBJS=$(ls *.o | grep -v "^rdo_cand")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib_$NAME.so $OBJS ../../build/rdo_cand_$NAME.o -ldl
ls -la ../../build/lib_$NAME.so
