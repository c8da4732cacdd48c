#!/bin/bash
# PMC pass (instruction counts) + kernel stats of the CDEF frame path: tools/cdef_time.py under rocprofv3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_cdef_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_cdef -o p -- python $GRAFT_REPO_ROOT/tools/cdef_time.py --reps 3 > /tmp/pmc_cdef.log 2>&1
f=$(find /tmp/pmc_cdef -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/sq.csv || tail -5 /tmp/pmc_cdef.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_cdef -o p -- python $GRAFT_REPO_ROOT/tools/cdef_time.py --reps 20 > /tmp/stats_cdef.log 2>&1
find /tmp/stats_cdef -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/sq.csv")):
    k = r["Kernel_Name"]
    if "cdef" not in k: continue
    k = k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: (min(v), max(v), len(v)) for c, v in cs.items() if c in ("SQ_INSTS_VALU", "SQ_WAVES")})
PY
head -12 $OUT/kernel_stats.csv | cut -c1-160
