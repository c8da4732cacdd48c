#!/bin/bash
# timing-only breakdown of the pixel-domain chain: variants with phases stubbed out (results wrong by design)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_h
mkdir -p $OUT
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in /tmp/lib_orig.so build/lib_px_d.so build/lib_px_dc.so build/lib_px_dcr.so; do
  cp $v rav1e_amd/librav1e_hip.so
  python bench.py --steps 20 --warmup 3 --no-extra --cpu-seconds 0 --chain pixel --bit-depth 8 2>&1 | tail -1 > /tmp/l.json
  python3 -c "
import json; d=json.loads(open('/tmp/l.json').read()); print('$v pixel', d['value'], d.get('kernel_ms'))"
done
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
python bench.py --steps 20 --warmup 3 --no-extra --cpu-seconds 0 --chain full --bit-depth 8 2>&1 | tail -1 > /tmp/l.json
python3 -c "
import json; d=json.loads(open('/tmp/l.json').read()); print('full', d['value'], d.get('kernel_ms'))"
python bench.py --steps 20 --warmup 3 --no-extra --cpu-seconds 0 --bit-depth 8 2>&1 | tail -1 > /tmp/l.json
python3 -c "
import json; d=json.loads(open('/tmp/l.json').read()); print('cand', d['value'], d.get('kernel_ms'))"
done 2>&1 | tee $OUT/ab.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
