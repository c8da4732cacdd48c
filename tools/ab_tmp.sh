mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rdo or satd or fused" 2>&1 | tail -4
for rep in 1 2; do for v in old new; do
  cp build/ab/$v.so rav1e_amd/librav1e_hip.so
  for bd in 8 10; do
    timeout 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --bit-depth $bd 2>&1 | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', $bd, d['value'], d['kernel_ms'])"
  done
done; done
cp build/ab/new.so rav1e_amd/librav1e_hip.so
