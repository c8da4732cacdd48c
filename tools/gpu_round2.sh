bash tools/gpu_round.sh r02_v3 2>&1 | tail -40
bash tools/gpu_mfma.sh r02_mfma 2>&1 | tail -14
