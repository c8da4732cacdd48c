#!/bin/bash
# PMC counter passes for the bench (run via gpurun). usage: tools/gpu_pmc.sh tag [bench args]
TAG=${1:-pmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$TAG/$n -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-events --no-extra $BENCH_ARGS > /tmp/pmc_$TAG.$n.log 2>&1
  f=$(find /tmp/pmc_$TAG/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/$n.csv || (echo "no counters for $n"; tail -5 /tmp/pmc_$TAG.$n.log)
}
BENCH_ARGS="$@"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 - <<PY
import csv,glob,collections,os
out="$OUT"
for f in sorted(glob.glob(out+"/*.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")[:60]
        if "k_rdo" not in k and "k_dist" not in k and "k_fwd" not in k and "k_mc" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        cnt[(k,r["Counter_Name"])]+=1
    for k,v in agg.items():
        print(os.path.basename(f),k,{c:round(x/cnt[(k,c)]) for c,x in v.items()})
PY
