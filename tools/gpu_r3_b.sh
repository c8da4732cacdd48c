#!/bin/bash
# parity of the full build on the kernels this round touches, then A/B of variants
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "rdo_cand or mc_ or fused_candidate or pixel_candidate or dist_and_mc" 2>&1 | tail -5
bash tools/gpu_ab_multi.sh ab2 "build/lib_r2base.so build/lib_v3.so build/lib_v3p.so"
