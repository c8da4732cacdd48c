#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every launch of bench.py INCLUDING its extra / config lines (run via gpurun).
# usage: tools/gpu_pmc_lines.sh tag   -> gpurun_out/<tag>/{tcc1,tcc2}.csv ; summarise with
#   python tools/pmc_summary.py gpurun_out/<tag> /tmp/x.json profiles/rNN_pmc_launches.json
TAG=${1:-pmc_lines}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pass in "tcc1 FETCH_SIZE" "tcc2 WRITE_SIZE"; do
  set -- $pass
  timeout 900 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmcl_$TAG/$1 -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --prewarm-ms 0 --cpu-seconds 0 --no-events > /tmp/pmcl_$TAG.$1.log 2>&1
  f=$(find /tmp/pmcl_$TAG/$1 -name "*counter_collection.csv" | head -1)
  # keep only our kernels (the csv of a whole bench run is tens of MB)
  [ -n "$f" ] && (head -1 $f; grep '"void (anonymous namespace)::k_' $f) > $OUT/$1.csv || (echo "no counters for $1"; tail -5 /tmp/pmcl_$TAG.$1.log)
done
ls -la $OUT
