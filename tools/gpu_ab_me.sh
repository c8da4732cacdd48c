#!/bin/bash
# parity of the tile ME, then same-box A/B (build/ab/old.so vs new.so) of tools/bench_me.py --tile-only
mkdir -p gpurun_out/abme
timeout 900 python -m pytest tests -m gpu -x -q -k "estimate_tile_motion" 2>&1 | tail -4 | tee gpurun_out/abme/pytest.log
for rep in 1 2; do for v in old new; do
  cp build/ab/$v.so rav1e_amd/librav1e_hip.so
  timeout 300 python tools/bench_me.py --tile-only --reps 10 2>&1 | grep "^{" | python3 -c "
import sys,json
print('$v', [(d['jobs'], d['ms']) for d in map(json.loads, sys.stdin)])" | tee -a gpurun_out/abme/ab.log
done; done
cp build/ab/new.so rav1e_amd/librav1e_hip.so
timeout 300 python tools/bench_me.py --reps 3 --cpu 2>&1 | grep "^{" | tail -3 | tee gpurun_out/abme/cpu_check.log
