#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_d
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
