#!/bin/bash
# round 4, call A: the whole GPU suite on the new build + bench with the exchange self-check
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 600 python bench.py --no-extra --verify-exchange 2>&1 | grep "^{" | tee $OUT/bench.json | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_a/bench.json").read())
print("value", d["value"], "exchange_ok", d["config"].get("exchange_ok"), "parity", d.get("parity"))
PY
