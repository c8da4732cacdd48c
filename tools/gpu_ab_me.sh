#!/bin/bash
# parity of the tile ME, then same-box A/B of tools/bench_me.py --tile-only over the library
# variants under build/ab/ (old.so new.so ...; the last one stays installed)
VARIANTS=${VARIANTS:-"old new"}
mkdir -p gpurun_out/abme
last=""
for rep in 1 2; do for v in $VARIANTS; do
  cp build/ab/$v.so rav1e_amd/librav1e_hip.so
  timeout 300 python tools/bench_me.py --tile-only --reps 10 --bit-depth ${BD:-8} 2>&1 | grep "^{" | python3 -c "
import sys,json
print('$v', [(d['jobs'], d['ms']) for d in map(json.loads, sys.stdin) if 'jobs' in d])" | tee -a gpurun_out/abme/ab.log
  last=$v
done; done
timeout 900 python -m pytest tests -m gpu -x -q -k "estimate_tile_motion or me_ref or estimate_motion_blocks" 2>&1 | tail -4 | tee gpurun_out/abme/pytest.log
