#!/bin/bash
# round 4: the peer-store exchange on one GPU (two processes sharing it), the bench's world-1
# self-check, and the bench's N = 2 control flow with both ranks on cuda:0 (dry run: the numbers mean nothing)
cd /root/repo || exit 1
mkdir -p gpurun_out/r04_push
timeout 400 python -m pytest tests/test_distributed.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r04_push/pytest.log
( timeout 300 python bench.py --verify-exchange --steps 20 --warmup 5 --cpu-seconds 0 --no-extra ; echo "bench rc=$?" ) > gpurun_out/r04_push/bench_verify.log 2>&1
for ex in auto p2p; do
( timeout 300 python bench.py --gpus 2 --single-device --backend gloo --exchange $ex --steps 12 --warmup 3 --cpu-seconds 0 --no-extra ; echo "bench rc=$?" ) > gpurun_out/r04_push/bench_dry2_$ex.log 2>&1
done
( timeout 300 python bench.py --gpus 4 --single-device --backend gloo --steps 12 --warmup 3 --cpu-seconds 0 --no-extra ; echo "bench rc=$?" ) > gpurun_out/r04_push/bench_dry4_auto.log 2>&1
tail -4 gpurun_out/r04_push/pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_push/bench_*.log")):
    js = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if not js:
        print(f, "NO JSON"); print(open(f).read()[-1500:]); continue
    d = json.loads(js[-1]); c = d["config"]
    print(f, d["value"], d["ms_per_step"], c.get("exchange"), c.get("exchange_ok"), c.get("compute_ms"), c.get("exchange_ms"))
PY
