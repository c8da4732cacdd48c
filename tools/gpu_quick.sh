#!/bin/bash
# quick iteration loop on the GPU box: fused/mc parity + bench (no CPU baseline)
TAG=${1:-q}; shift
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${K:-rdo_cand or mc_put or smoke}" 2>&1 | tail -4 | tee gpurun_out/$TAG/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 "$@" 2>&1 | tail -1 | tee gpurun_out/$TAG/bench.log | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l); continue
    print('value',d['value'],'ms/step',d['ms_per_step'],'kernel_ms',d['kernel_ms'],'frac',d['roofline']['frac'])
"
