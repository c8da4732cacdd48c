#!/bin/bash
# texture-addresser (vector-memory issue) counters of the bench's kernels. usage: tools/gpu_pmc_ta.sh tag [bench args]
TAG=${1:-pmc_ta}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcta_$TAG/$n -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-events --no-extra $BENCH_ARGS > /tmp/pmcta_$TAG.$n.log 2>&1
  f=$(find /tmp/pmcta_$TAG/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep 'k_rdo_cand' $f) > $OUT/$n.csv || (echo "no counters for $n"; tail -5 /tmp/pmcta_$TAG.$n.log)
}
BENCH_ARGS="$@"
run ta1 TA_TA_BUSY TA_FLAT_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TCP_PENDING_STALL_CYCLES
run ta3 GRBM_GUI_ACTIVE
python3 - $OUT <<'PY'
import csv,glob,collections,sys,os
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1]+"/*.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]; k=k[k.index("k_rdo_cand"):k.index(">")+1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    d={c:sum(x)/len(x) for c,x in v.items()}
    g=d.get("GRBM_GUI_ACTIVE",0)/8
    print(k, {c:round(x) for c,x in d.items()}, "TA_BUSY/(256 CU x cycles) = %.3f"%(d.get("TA_TA_BUSY",0)/(256*g) if g else 0))
PY
