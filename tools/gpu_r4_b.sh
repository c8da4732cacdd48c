#!/bin/bash
# round 4, call B: quantizer rewrite -- parity (whole GPU suite) + same-box A/B of the full and pixel chains
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 | tee $OUT/pytest_gpu.log
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in build/lib_prevq.so build/lib_newq.so; do
  cp $v rav1e_amd/librav1e_hip.so
  for chain in full pixel; do for bd in 8 10; do
    python bench.py --steps 20 --warmup 3 --no-extra --cpu-seconds 0 --chain $chain --bit-depth $bd 2>&1 | tail -1 > /tmp/l.json
    python3 -c "
import json; d=json.loads(open('/tmp/l.json').read()); print('$v $chain $bd', d['value'], d.get('kernel_ms'))"
  done; done
done; done 2>&1 | tee $OUT/ab.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
