#!/bin/bash
# Everything the round's profiles/ are made of, in one gpurun call.
# usage (via gpurun): tools/gpu_round.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-r01_v9}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -6 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench (default flags)"
timeout 900 python bench.py 2>&1 | grep "^{" | tee $OUT/bench.json | cut -c1-400
for bd in 8 10; do
  [ $bd = 10 ] && timeout 600 python bench.py --cpu-seconds 0 --bit-depth 10 2>&1 | grep "^{" > $OUT/bench_10bit.json
  for chain in full pixel; do
    timeout 600 python bench.py --cpu-seconds 0 --chain $chain --bit-depth $bd 2>&1 | grep "^{" > $OUT/${chain}_chain_${bd}bit.json
  done
  timeout 900 python tools/bench_kernels.py --bit-depth $bd 2>&1 | grep "^{" > $OUT/kernels_${bd}bit.jsonl
  timeout 600 python tools/frame_pipeline.py --bit-depth $bd 2>&1 | grep "^{" > $OUT/frame_pipeline_${bd}bit.json
done
timeout 900 python tools/bench_me.py --cpu 2>&1 | grep "^{" > $OUT/me_4k.jsonl
echo "== rocprof stats"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --no-extra > /tmp/prof_$TAG.log 2>&1; tail -1 /tmp/prof_$TAG.log | cut -c1-200)
find /tmp/prof_$TAG -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \; 2>/dev/null
echo "== rocprof stats of the frame pipeline"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profp_$TAG -o prof -- python $GRAFT_REPO_ROOT/tools/frame_pipeline.py --reps 3 > /tmp/profp_$TAG.log 2>&1; tail -1 /tmp/profp_$TAG.log | cut -c1-120)
find /tmp/profp_$TAG -name "*kernel_stats*" -exec cp {} $OUT/frame_pipeline_kernel_stats.csv \; 2>/dev/null
echo "== pmc"
bash tools/gpu_pmc.sh ${TAG}_pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc $OUT/pmc_summary.json 2>&1 | tail -2
ls $OUT; for f in $OUT/*_chain_*.json $OUT/bench_10bit.json $OUT/frame_pipeline_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read()); print('$f'.split('/')[-1], d.get('value'), d.get('kernel_ms'), d.get('stage_ms'), d.get('two_stream_ms (ME of the next frame beside the other stages)'))"; done
cat $OUT/me_4k.jsonl | cut -c1-160
