#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, bench, rocprof summary.
# usage: tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 gfx > $OUT/device.txt
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 | grep "^{" | tee $OUT/bench.log
echo "== rocprof"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > /tmp/prof_$TAG.log 2>&1; tail -2 /tmp/prof_$TAG.log)
find /tmp/prof_$TAG -name "*kernel_stats*" -exec cp {} $OUT/ \; 2>/dev/null
find /tmp/prof_$TAG -type f | head -8
ls $OUT
