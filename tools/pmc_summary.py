#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/gpu_pmc.sh into one JSON
(per kernel: average counter values per dispatch, HBM traffic per launch).

usage: python tools/pmc_summary.py gpurun_out/<tag> profiles/<name>.json [profiles/<name>_launches.json]

HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE
are reported in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by
exactly 2x (calibrated for 16 B/lane streams; our loads are unaligned dwords
and bytes, so the doubled figure is an upper estimate and the raw one a lower
bound -- both are kept).  Counters come from separate --pmc passes (FETCH_SIZE
and WRITE_SIZE do not fit one pass)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(k_\w+)<([^>]*)>", name)
    return "%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else name[:60]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)     # dispatch durations in the GRBM pass (ns), per kernel
    for f in sorted(glob.glob(os.path.join(src, "*.csv"))):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_" not in k:
                continue
            agg[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur[short(k)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, cs in agg.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        if dur.get(k) and "GRBM_GUI_ACTIVE" in d:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / the dispatch's own duration in
            # that pass = the clock the chip held while this kernel ran
            d["grbm_pass_duration_ns"] = sum(dur[k]) / len(dur[k])
            d["clock_ghz_by_counters"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["grbm_pass_duration_ns"]
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
            d["hbm_read_bytes_x2"] = d["FETCH_SIZE"] * 2048
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
            d["hbm_traffic_bytes"] = d["hbm_read_bytes_x2"] + d["hbm_write_bytes"]
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
            d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        out[k] = {c: round(v, 4 if c == "clock_ghz_by_counters" else 1) for c, v in d.items()}
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    # per (kernel, grid size): the launches of bench.py's extra / config lines share kernels with the
    # headline but not their grids (bench.py launch_pmc)
    if len(sys.argv) > 3:
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in sorted(glob.glob(os.path.join(src, "*.csv"))):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "")
                if "k_" not in k:
                    continue
                by["%s@%s" % (short(k), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        lo = {}
        for k, cs in by.items():
            d = {c: sum(v) / len(v) for c, v in cs.items()}
            d["n"] = min(len(v) for v in cs.values())
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                d["hbm_traffic_bytes"] = d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024
            lo[k] = {c: round(v, 1) for c, v in d.items()}
        json.dump(lo, open(sys.argv[3], "w"), indent=1, sort_keys=True)
        print(len(lo), "launch groups ->", sys.argv[3])
    for k, d in out.items():
        print(k, {c: d[c] for c in ("hbm_traffic_bytes", "valu_insts_per_wave") if c in d})


if __name__ == "__main__":
    main()
