#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_j
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "pixel or pred_cand or rdo" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/pytest.log
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do
for v in build/lib_nokeep.so /tmp/lib_orig.so; do
  cp $v rav1e_amd/librav1e_hip.so
  python bench.py --steps 20 --warmup 3 --no-extra --cpu-seconds 0 --chain pixel --bit-depth 8 2>&1 | tail -1 > /tmp/l.json
  python3 -c "
import json; d=json.loads(open('/tmp/l.json').read()); print('$v pixel', d['value'], d.get('kernel_ms'), d['rdo_only']['value'])"
done; done 2>&1 | tee $OUT/ab.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
