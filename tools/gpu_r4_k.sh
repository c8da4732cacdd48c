#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_k
mkdir -p $OUT
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/pytest.log
for c in cand full pixel; do
  python bench.py --no-extra --cpu-seconds 0 --chain $c 2>/dev/null | grep "^{" > /tmp/l.json
  python3 -c "
import json; d=json.loads(open('/tmp/l.json').read()); print('$c', d['value'], d['kernel_ms'], d.get('rdo_only',{}).get('value'))"
done | tee $OUT/chains.txt
