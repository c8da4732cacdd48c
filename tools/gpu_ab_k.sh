#!/bin/bash
# parity subset ($1 = pytest -k), then same-box A/B (build/ab/old.so vs new.so) of the
# tools/bench_kernels.py rows whose name matches $2 (grep pattern)
mkdir -p gpurun_out/abk
timeout 900 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -4 | tee gpurun_out/abk/pytest.log
for v in old new old new; do
  cp build/ab/$v.so rav1e_amd/librav1e_hip.so
  timeout 600 python tools/bench_kernels.py --bit-depth ${BD:-8} 2>&1 | grep "^{" | grep "$2" | python3 -c "
import sys,json
print('$v', [(d['kernel'][:28], d['ms']) for d in map(json.loads, sys.stdin)])" | tee -a gpurun_out/abk/ab.log
done
cp build/ab/new.so rav1e_amd/librav1e_hip.so
