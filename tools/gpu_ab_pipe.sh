#!/bin/bash
# same-box A/B over library variants under build/ab/: the tile-ME parity tests, then the ME stage of
# tools/frame_pipeline.py (8 tiles x 3 references) and tools/bench_me.py's 16 / 64 job lines
# usage (via gpurun): VARIANTS="base new" BD=8 tools/gpu_ab_pipe.sh      (the last variant stays installed)
VARIANTS=${VARIANTS:-"base new"}
mkdir -p gpurun_out/abpipe
for rep in 1 2; do for v in $VARIANTS; do
  cp build/ab/$v.so rav1e_amd/librav1e_hip.so
  timeout 200 python tools/frame_pipeline.py --bit-depth ${BD:-8} 2>&1 | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v pipeline', d['stage_ms']['estimate_tile_motion_8tiles_x_3refs'], d['sum_ms'], d['two_stream_ms (ME of the next frame beside the other stages)'])" | tee -a gpurun_out/abpipe/ab.log
  timeout 200 python tools/bench_me.py --tile-only --reps 10 --bit-depth ${BD:-8} 2>&1 | grep "^{" | python3 -c "
import sys,json
print('$v', [(d['jobs'], d['ms']) for d in map(json.loads, sys.stdin) if 'jobs' in d])" | tee -a gpurun_out/abpipe/ab.log
done; done
timeout 900 python -m pytest tests -m gpu -x -q -k "estimate_tile_motion or me_ref or motion_estimation" 2>&1 | tail -3 | tee gpurun_out/abpipe/pytest.log
