#!/bin/bash
# Round 4: everything profiles/r04_* is made of, in one gpurun call.
# usage (via gpurun): tools/gpu_round4.sh <tag>      -> gpurun_out/<tag>/ (+ <tag>_pmc*, <tag>_pmc_lines)
TAG=${1:-r04_v1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -6 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== bench (default flags)"
timeout 900 python bench.py 2>/dev/null | grep "^{" | tee $OUT/bench.json | cut -c1-300
for bd in 8 10; do
  [ $bd = 10 ] && timeout 600 python bench.py --cpu-seconds 0 --no-extra --bit-depth 10 2>/dev/null | grep "^{" > $OUT/bench_10bit.json
  for chain in full pixel; do
    timeout 600 python bench.py --cpu-seconds 0 --chain $chain --bit-depth $bd 2>/dev/null | grep "^{" > $OUT/${chain}_chain_${bd}bit.json
  done
  timeout 900 python tools/bench_kernels.py --bit-depth $bd 2>/dev/null | grep "^{" > $OUT/kernels_${bd}bit.jsonl
  timeout 600 python tools/frame_pipeline.py --bit-depth $bd 2>/dev/null | grep "^{" > $OUT/frame_pipeline_${bd}bit.json
done
timeout 900 python tools/bench_me.py --cpu 2>/dev/null | grep "^{" > $OUT/me_4k.jsonl
echo "== rocprof stats"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-extra > /tmp/prof_$TAG.log 2>&1; tail -1 /tmp/prof_$TAG.log | cut -c1-200)
find /tmp/prof_$TAG -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \; 2>/dev/null
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profp_$TAG -o prof -- python $GRAFT_REPO_ROOT/tools/frame_pipeline.py --reps 3 > /tmp/profp_$TAG.log 2>&1; tail -1 /tmp/profp_$TAG.log | cut -c1-120)
find /tmp/profp_$TAG -name "*kernel_stats*" -exec cp {} $OUT/frame_pipeline_kernel_stats.csv \; 2>/dev/null
echo "== pmc: headline 8-bit, 10-bit, pixel chain; per-launch FETCH / WRITE of the lines"
bash tools/gpu_pmc.sh ${TAG}_pmc > $OUT/pmc.log 2>&1
bash tools/gpu_pmc.sh ${TAG}_pmc10 --bit-depth 10 > $OUT/pmc10.log 2>&1
bash tools/gpu_pmc.sh ${TAG}_pmcpx --chain pixel > $OUT/pmcpx.log 2>&1
bash tools/gpu_pmc_lines.sh ${TAG}_pmc_lines > $OUT/pmc_lines.log 2>&1
ls $OUT
for f in $OUT/*_chain_*.json $OUT/bench_10bit.json $OUT/frame_pipeline_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read()); print('$f'.split('/')[-1], d.get('value'), d.get('kernel_ms'), d.get('rdo_only',{}).get('value'), d.get('stage_ms'))"; done
cut -c1-160 $OUT/me_4k.jsonl
