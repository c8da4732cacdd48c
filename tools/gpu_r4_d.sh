#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py 2>$OUT/bench.err | grep "^{" > $OUT/bench.json; tail -3 $OUT/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_d/bench.json").read())
print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity_ok"), d.get("parity_checked"))
r=d["roofline"]; print({k:r[k] for k in ("bound","achieved","peak","frac","traffic","binding_roof")}); print(r["valu"]); print(r["hbm"])
cb=d["cpu_baseline"]; print({k:cb[k] for k in cb if k!="reference_probe"})
for e in d.get("extra_lines",[]): print(e["name"], e["value"], e["roofline"]["binding_roof"], e["roofline"]["frac"], e["roofline"]["hbm"].get("frac_unique"), e["roofline"]["hbm"].get("reads_cache_resident"), e["parity_ok"])
PY
bash tools/gpu_pmc_lines.sh r04_pmc_lines
