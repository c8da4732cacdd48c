#!/usr/bin/env python3
"""One rocprofv3 counter_collection.csv -> JSON on stdout: per kernel (template arguments kept), the average
of every counter per dispatch, the dispatch count, the average dispatch duration of that pass and, where the
pass has them, the derived figures (VALU instructions per wave, LDS bank-conflict share, HBM bytes with the
gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md "HBM").  Used by tools/gpu_lease.sh (pmc_* steps)."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(k_\w+)<([^>]*)>", name)
    return "%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else name[:80]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        k = r.get("Kernel_Name", "")
        if "k_" not in k:
            continue
        key = "%s@%s" % (short(k), r.get("Grid_Size", "?"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        try:
            dur[key][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            pass
    out = {}
    for k, cs in agg.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["dispatches"] = min(len(v) for v in cs.values())
        if dur.get(k):
            d["pass_duration_us"] = sum(dur[k].values()) / len(dur[k]) / 1e3
        if "SQ_INSTS_VALU" in d and d.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        if d.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in d:
            d["wait_any_share"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_traffic_bytes"] = d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024
        out[k] = {c: round(v, 4) for c, v in d.items()}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
