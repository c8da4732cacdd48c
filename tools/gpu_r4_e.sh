#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 | tee $OUT/pytest_gpu.log
timeout 900 python tools/bench_kernels.py --bit-depth 8 2>&1 | grep "^{" > $OUT/kernels_8bit.jsonl
grep -E "intra|deblock|edges" $OUT/kernels_8bit.jsonl | cut -c1-160
