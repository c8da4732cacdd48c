#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "intra or prescreen or small_input or predict or edges or cfl" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee $OUT/pytest_gpu.log
for sh in 0 1 2 3; do
  echo "== R1_EDGES_LPC_SHIFT=$sh"
  R1_EDGES_LPC_SHIFT=$sh timeout 600 python tools/bench_kernels.py --bit-depth 8 2>&1 | grep "^{" | grep -E "get_intra_edges|predict_intra" | cut -c1-110
done 2>&1 | tee $OUT/ab.txt
