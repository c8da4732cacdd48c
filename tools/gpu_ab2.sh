#!/bin/bash
# parity subset, then same-box A/B of build/ab/old.so vs build/ab/new.so on the headline and
# the pixel chain.  usage (via gpurun): tools/gpu_ab2.sh [pytest -k expression]
mkdir -p gpurun_out/ab
K=${1:-"rdo or fused or small_input"}
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -4 | tee gpurun_out/ab/pytest.log
for rep in 1 2; do for v in old new; do
  cp build/ab/$v.so rav1e_amd/librav1e_hip.so
  for cfg in "8 cand" "10 cand" "8 pixel"; do
    set -- $cfg
    timeout 300 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --bit-depth $1 --chain $2 2>&1 | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', '$1', '$2', d['value'], d['kernel_ms'])" | tee -a gpurun_out/ab/ab.log
  done
done; done
cp build/ab/new.so rav1e_amd/librav1e_hip.so
