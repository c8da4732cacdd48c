#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_l
mkdir -p $OUT
for rep in 1 2; do for st in 1 2 4; do for bd in 8 10; do
  python bench.py --no-extra --cpu-seconds 0 --streams $st --bit-depth $bd 2>/dev/null | grep "^{" > /tmp/l.json
  python3 -c "
import json; d=json.loads(open('/tmp/l.json').read()); print('streams $st bd $bd', d['value'], d['ms_per_step'], d['kernel_ms'])"
done; done; done | tee $OUT/streams.txt
