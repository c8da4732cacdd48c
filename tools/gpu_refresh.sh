#!/bin/bash
# re-measure the per-kernel table and the frame pipeline after a kernel change (via gpurun)
TAG=${1:-refresh}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
for bd in 8 10; do
  timeout 900 python tools/bench_kernels.py --bit-depth $bd 2>&1 | grep "^{" > $OUT/kernels_${bd}bit.jsonl
  timeout 600 python tools/frame_pipeline.py --bit-depth $bd 2>&1 | grep "^{" | tee $OUT/frame_pipeline_${bd}bit.json | cut -c1-700
done
grep "pre-screen" $OUT/kernels_8bit.jsonl | cut -c1-200
