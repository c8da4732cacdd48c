#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ref_vectors.py -m gpu -q -x -k "cdef" 2>&1 | tail -30 | cut -c1-600 | tee $OUT/pytest_cdef.log
timeout 300 python tools/cdef_time.py 2>&1 | tee $OUT/cdef_time.jsonl
