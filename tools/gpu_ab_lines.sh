#!/bin/bash
# A/B of library variants on the bench's extra / config lines: tools/gpu_ab_lines.sh TAG "libs" name-substring
TAG=$1; LIBS=$2; PAT=$3
cd /root/repo || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for pass in 1 2; do
  for v in $LIBS; do
    cp $v rav1e_amd/librav1e_hip.so
    python bench.py --cpu-seconds 0 --steps 40 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for e in d['extra_lines']:
    if '$PAT' in e['name']: print('%-18s %-28s %10.0f %s parity %s' % ('$v'.split('/')[-1], e['name'], e['value'], e.get('kernel_ms'), e.get('parity_ok')))"
  done
done 2>&1 | tee $OUT/res.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
