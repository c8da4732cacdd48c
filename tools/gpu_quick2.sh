#!/bin/bash
# quick A/B after a kernel change: fused parity tests, then the bench at 1 and 4 streams
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-quick2}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plane_ops.py -x -q -k "rdo or fused or cand or plane or pad or downsample" 2>&1 | tail -3
for st in 1 4 1 4; do
  timeout 300 python bench.py --cpu-seconds 0 --no-extra --streams $st --steps 40 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $st', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
timeout 300 python bench.py --cpu-seconds 0 --no-extra --bit-depth 10 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('10-bit', d['value'], d['kernel_ms'])"
python tools/cpu_scaling.py 2>&1 | tail -12
