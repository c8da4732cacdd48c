#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "cdef or loop_filter or frame" 2>&1 | tail -15 | tee $OUT/pytest_cdef.log
timeout 300 python tools/cdef_time.py 2>&1 | tee $OUT/cdef_time.jsonl
s=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - s )) s" | tee $OUT/bench_default.time
tail -3 $OUT/bench_default.err
python - <<'PY' 2>&1 | tee $OUT/bench_summary.txt
import json, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_e/bench_default.json")
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "parity", d.get("parity_ok"))
    print("roofline", d["roofline"])
    for e in d.get("extra_lines", []):
        print(e.get("name"), e.get("value"), e.get("unit"), e.get("ms_per_step"), "parity", e.get("parity_ok"), e.get("parity_bad"), e.get("kernel_ms"))
except Exception as ex:
    print("summary failed", ex)
PY
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_ref_vectors.py -q -m gpu -k compound 2>&1 | grep -E "^E  |passed|failed" | head -5; done | tee $OUT/compound_repeat.log
