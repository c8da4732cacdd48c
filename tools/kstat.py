#!/usr/bin/env python3
"""Static instruction statistics of the fused-candidate kernels in a device assembly file
(hipcc -S --cuda-device-only): per kernel VALU total / slow class, LDS, VMEM, VGPRs, LDS bytes.
usage: tools/kstat.py file.s [substring]"""
import re, subprocess, sys
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_lshrrev_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_mov_b32", "v_not_b32", "v_min_i16", "v_add_f32")
def main():
    f = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "k_rdo_cand"
    kern, rows, meta = None, {}, {}
    for line in open(f):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1); rows[kern] = {}; continue
        if kern is None: continue
        t = line.strip()
        m2 = re.search(r"; (NumVgprs|ScratchSize|LDSByteSize|Occupancy): (\d+)", line)
        if m2: meta.setdefault(kern, {})[m2.group(1)] = int(m2.group(2))
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"): continue
        op = t.split()[0]
        if "dpp" in t or "sdwa" in t: op += "(x)"
        rows[kern][op] = rows[kern].get(op, 0) + 1
    for k, c in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        if pat not in name or not c: continue
        valu = {o: n for o, n in c.items() if o.startswith("v_")}
        fast = sum(n for o, n in valu.items() if o.replace("_e32", "").replace("_e64", "") in FAST)
        tot = sum(valu.values())
        short = re.sub(r"^void \(anonymous namespace\)::", "", name).split("(")[0]
        print("%-44s VALU %5d (slow %5d) LDS %4d VMEM %3d SALU %4d | %s" % (
            short, tot, tot - fast, sum(n for o, n in c.items() if o.startswith("ds_")),
            sum(n for o, n in c.items() if o.startswith(("global_", "buffer_"))),
            sum(n for o, n in c.items() if o.startswith("s_")), meta.get(k)))
main()
