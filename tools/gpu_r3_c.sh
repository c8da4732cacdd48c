#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_c
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
timeout 600 python bench.py --cpu-seconds 4 2>&1 | tail -1 > $OUT/bench.json
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read()); print('bench', d['value'], d['kernel_ms'], d.get('parity_ok'), [ (e['name'], e['value'], e['parity_ok']) for e in d.get('extra_lines',[])])"
timeout 300 python bench.py --gpus 2 --single-device --backend gloo --steps 4 --warmup 1 --cpu-seconds 0 --no-extra 2>&1 | tail -3 | cut -c1-1500
