#!/bin/bash
# round 4, call C: k_me_persist with the hand-over spelled as release / acquire fences (R1_ME_FORMAL) vs the product
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_c
mkdir -p $OUT
export TMPDIR=/tmp
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
for v in /tmp/lib_orig.so build/lib_me_formal.so; do
  cp $v rav1e_amd/librav1e_hip.so
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_ref_vectors.py -x -q -k "me_ref" 2>&1 | grep -E "passed|failed|error" | tail -3
  timeout 600 python tools/bench_me.py --tile-only --reps 5 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print({k: d[k] for k in d if k in ('config','jobs','ms','launch','ms_per_frame','mode','label','gpu_ms','n_jobs')})"
done 2>&1 | tee $OUT/me_fence_ab.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
