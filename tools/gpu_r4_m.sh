#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_m
mkdir -p $OUT
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/pytest.log
timeout 600 python bench.py --cpu-seconds 4 2>/dev/null | grep "^{" > $OUT/bench.json
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_m/bench.json").read())
print("value", d["value"], d["kernel_ms"], d.get("parity_ok"))
for e in d["extra_lines"]: print(e["name"], e["value"], e["ms_per_step"], e["kernel_ms"], e["parity_ok"])
PY
