#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "intra or prescreen or small_input or predict" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee $OUT/pytest_gpu.log
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for v in /tmp/lib_orig.so build/lib_pred_noloop.so /tmp/lib_orig.so build/lib_pred_noloop.so; do
  cp $v rav1e_amd/librav1e_hip.so
  echo "== $v"
  for bd in 8 10; do timeout 600 python tools/bench_kernels.py --bit-depth $bd 2>&1 | grep "^{" | grep -E "pre-screen|deblock" | cut -c1-150; done
done 2>&1 | tee $OUT/ab.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
