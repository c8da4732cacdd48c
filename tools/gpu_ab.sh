#!/bin/bash
# A/B the headline bench between prebuilt library variants on ONE GPU box
# (box-to-box clock variance is +-3 %, larger than most single optimisations).
# usage (via gpurun): tools/gpu_ab.sh a.so b.so [bench args]   -- runs a b a b
A=$1; B=$2; shift 2
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for v in $A $B $A $B; do
  cp $v rav1e_amd/librav1e_hip.so
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms'])"
done
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
