#!/bin/bash
# iteration loop for the full-candidate chain: parity (quantize + full cand) and both bench chains
TAG=${1:-f}; shift
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${K:-rdo_full or quantize}" 2>&1 | tail -4 | tee gpurun_out/$TAG/pytest.log
for bd in ${BDS:-8 10}; do
  for chain in ${CHAINS:-pixel full cand}; do
    timeout 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --chain $chain --bit-depth $bd "$@" 2>&1 | grep "^{" | tee gpurun_out/$TAG/bench_${chain}_$bd.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$chain', $bd, d['value'], d['kernel_ms'], d['roofline']['frac'])"
  done
done
