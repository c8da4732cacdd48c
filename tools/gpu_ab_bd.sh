#!/bin/bash
# A/B of prebuilt library variants at ONE bit depth on one box:
#   tools/gpu_ab_bd.sh TAG BD "lib1.so lib2.so ..." [bench args]
# pass 1 with a short CPU parity leg (parity_ok must be true), passes 2 and 3 timing only
TAG=$1; BD=$2; LIBS=$3; shift 3
cd /root/repo || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for pass in 2 0 0; do
  for v in $LIBS; do
    cp $v rav1e_amd/librav1e_hip.so
    python bench.py --no-extra --cpu-seconds $pass --bit-depth $BD "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-22s bd $BD %9.0f Mpx/s kernel_ms %s parity %s' % ('$v'.split('/')[-1], d['value'], d['kernel_ms'], d.get('parity_ok')))"
  done
done 2>&1 | tee $OUT/res.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
