#!/bin/bash
# the per-kernel rows, the frame pipeline and the ME rows re-measured at sustained clocks
cd /root/repo || exit 1
OUT=gpurun_out/r04_v4; mkdir -p $OUT
for bd in 8 10; do
  timeout 900 python tools/bench_kernels.py --bit-depth $bd 2>/dev/null | grep "^{" > $OUT/kernels_${bd}bit.jsonl
  timeout 600 python tools/frame_pipeline.py --bit-depth $bd 2>/dev/null | grep "^{" > $OUT/frame_pipeline_${bd}bit.json
done
timeout 900 python tools/bench_me.py 2>/dev/null | grep "^{" > $OUT/me_4k.jsonl
for f in $OUT/frame_pipeline_*.json; do python3 -c "
import json
d=json.loads(open('$f').read()); print('$f'.split('/')[-1], d['sum_ms'], d['two_stream_ms (ME of the next frame beside the other stages)'], d['stage_ms'])"; done
cut -c1-150 $OUT/me_4k.jsonl | head -6
