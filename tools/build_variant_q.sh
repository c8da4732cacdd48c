#!/bin/bash
# Experiment builds of the chain kernels: tools/build_variant_q.sh NAME QM [-DFLAG ...]
#   -> build/lib_NAME.so = the current objects with the 8- and 10-bit QM slices of rdo_cand.hip
#      recompiled with the flags (the two compile in parallel; QM = 2 takes a few minutes).
set -e
NAME=$1; QM=$2; shift 2
cd "$(dirname "$0")/../rav1e_amd/csrc"
mkdir -p ../../build
for b in 8 10; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed \
    -DR1_RDO_TU_BD=$b -DR1_RDO_TU_QM=$QM "$@" -c rdo_cand.hip -o ../../build/rdo_cand_b${b}_q${QM}_$NAME.o &
done
wait
OBJS=$(ls *.o | grep -v "^rdo_cand_b8_q${QM}.o\|^rdo_cand_b10_q${QM}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib_$NAME.so $OBJS ../../build/rdo_cand_b8_q${QM}_$NAME.o ../../build/rdo_cand_b10_q${QM}_$NAME.o -ldl
ls -la ../../build/lib_$NAME.so
