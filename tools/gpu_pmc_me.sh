#!/bin/bash
# PMC counters of k_me_persist / k_me_diag for one tile-ME configuration (tools/bench_me.py --only N). usage: tools/gpu_pmc_me.sh tag [cfg]
TAG=${1:-pmcme}; CFG=${2:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$TAG/$n -o p -- \
    python $GRAFT_REPO_ROOT/tools/bench_me.py --only $CFG --reps 1 > /tmp/pmc_$TAG.$n.log 2>&1
  f=$(find /tmp/pmc_$TAG/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/$n.csv || (echo "no counters for $n"; tail -5 /tmp/pmc_$TAG.$n.log)
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
python3 - <<PY
import csv,glob,collections
out="$OUT"
for f in sorted(glob.glob(out+"/*.csv")):
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_me_diag" not in r.get("Kernel_Name","") and "k_me_persist" not in r.get("Kernel_Name",""): continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print(f.split("/")[-1], {k:(round(v), n[k]) for k,v in agg.items()})
PY
