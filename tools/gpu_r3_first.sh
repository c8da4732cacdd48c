#!/bin/bash
# round 3, first call: micro-benchmarks + baseline of the round-2 build on this box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_first
mkdir -p $OUT
timeout 300 tools/ubench/valu_rate2 2>&1 | tee $OUT/valu_rate2.txt | tail -30
(cd tools/ubench && timeout 120 /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate 2>/dev/null; timeout 300 ./valu_rate > $OUT/valu_rate.txt 2>&1)
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extra 2>&1 | tail -1 > $OUT/bench_base.json
python3 -c "
import json
d=json.loads(open('$OUT/bench_base.json').read()); print('base 8-bit', d['value'], d['kernel_ms'])"
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extra --bit-depth 10 2>&1 | tail -1 > $OUT/bench_base_10.json
python3 -c "
import json
d=json.loads(open('$OUT/bench_base_10.json').read()); print('base 10-bit', d['value'], d['kernel_ms'])"
