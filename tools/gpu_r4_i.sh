#!/bin/bash
bash tools/gpu_ab_multi.sh r04_ab2 "build/lib_nopad.so build/lib_pad.so"
export TMPDIR=/tmp
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
cp build/lib_pad.so rav1e_amd/librav1e_hip.so
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_ab2
for bd in 8 10; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_pad$bd -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-events --no-extra --bit-depth $bd > /tmp/pmc_pad$bd.log 2>&1)
f=$(find /tmp/pmc_pad$bd -name "*counter_collection.csv" | head -1)
python3 - $f $bd <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if "k_rdo_cand" in k: agg[k[31:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    d={c:sum(x)/len(x) for c,x in v.items()}
    print("pad", sys.argv[2], k, "lds conflict %.3f"%(d["SQ_LDS_BANK_CONFLICT"]/max(1,d["SQ_LDS_IDX_ACTIVE"])), "wait_any %.2f"%(d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"]))
PY
done | tee $OUT/pmc_pad.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
