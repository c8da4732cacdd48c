#!/bin/bash
# counters of the MFMA put_8tap experiment (csrc/mc_mfma.hip) beside the
# dot4/dot2 VALU path, one variant per rocprofv3 pass.  usage (via gpurun): tools/gpu_mfma.sh <tag>
TAG=${1:-mfma}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 python $GRAFT_REPO_ROOT/tools/bench_mc_mfma.py --steps 30 | grep "^{" > $OUT/mc_ab.jsonl
pass() { # name variant counters...
  n=$1; v=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/mf_$TAG/$n -o p -- \
    python $GRAFT_REPO_ROOT/tools/bench_mc_mfma.py --steps 3 --only $v > /tmp/mf_$TAG.$n.log 2>&1
  f=$(find /tmp/mf_$TAG/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/$n.csv || (echo "no counters for $n"; tail -3 /tmp/mf_$TAG.$n.log)
}
for v in mfma dot4; do
  pass ${v}_insts $v SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES
  pass ${v}_busy $v SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVE_CYCLES
done
python3 - <<PY
import csv, glob, collections, json, os, re
out = "$OUT"
res = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_mc" not in k:
            continue
        m = re.search(r"(k_mc\w*)<([^>]*)>", k)
        agg["%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        res[k].update({c: sum(v) / len(v) for c, v in cs.items()})
json.dump(res, open(out + "/mfma_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print(k, {c: round(x) for c, x in v.items()})
PY
