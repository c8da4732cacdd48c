#!/bin/bash
# like gpu_ab2.sh with the chain list in $CFGS ("bd chain" pairs separated by ;)
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests -m gpu -x -q -k "${1:-rdo}" 2>&1 | tail -3 | tee gpurun_out/ab/pytest.log
IFS=';' read -ra CF <<< "${CFGS:-8 pixel;10 pixel;8 full}"
for rep in 1 2; do for v in old new; do
  cp build/ab/$v.so rav1e_amd/librav1e_hip.so
  for cfg in "${CF[@]}"; do
    set -- $cfg
    timeout 300 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --bit-depth $1 --chain $2 2>&1 | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', '$1', '$2', d['value'], d['kernel_ms'])" | tee -a gpurun_out/ab/ab.log
  done
done; done
cp build/ab/new.so rav1e_amd/librav1e_hip.so
