#!/bin/bash
# A/B of prebuilt library variants on a chain (--chain full | pixel), 8- and 10-bit, two timing passes:
#   tools/gpu_ab_chain.sh TAG CHAIN "lib1.so lib2.so ..."
TAG=$1; CH=$2; LIBS=$3
cd /root/repo || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for pass in 1 2; do
  for bd in 8 10; do
    for v in $LIBS; do
      cp $v rav1e_amd/librav1e_hip.so
      python bench.py --no-extra --cpu-seconds 0 --bit-depth $bd --chain $CH 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-18s bd %2d %9.0f Mpx/s kernel_ms %s' % ('$v'.split('/')[-1], $bd, d['value'], d['kernel_ms']))"
    done
  done
done 2>&1 | tee $OUT/res.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
