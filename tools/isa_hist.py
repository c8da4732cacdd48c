#!/usr/bin/env python3
"""Instruction histogram of the gfx950 code of one .hip file, per kernel.

  python tools/isa_hist.py rav1e_amd/csrc/rdo_cand.hip [kernel-name-substring ...]
  python tools/isa_hist.py --json profiles/r03_isa_mix.json -DR1_HEADLINE_ONLY rav1e_amd/csrc/rdo_cand.hip k_rdo_cand
      (the per-kernel VALU mix bench.py's `roofline` prices the VALU-issue roof with)

Compiles the file to device assembly (hipcc -S --cuda-device-only) and counts, per kernel,
VALU / SALU / LDS / VMEM instructions.  VALU is weighted with the issue costs measured by
tools/ubench/valu_rate{,2}.hip on MI355X: add / sub / right shifts /
and / or / xor / mov issue in ~2.5 cycles per wave64, everything else (mul24, mad24, min /
max, med3, dot2 / dot4, v_pk_*, v_perm, v_alignb*, DPP, SDWA, v_sad, v_bfe, cvt_pk ...) in
~4.5 (v_lshlrev_b32 too).  Straight-line code only: loops count once (the fused kernels are fully unrolled)."""
import collections
import os
import re
import subprocess
import sys

# round 3 (tools/ubench/valu_rate2.hip, profiles/r03_ab_notes.md): v_lshlrev_b32 is NOT in the fast
# class (4.3 cycles; the right shifts are), fast = 2.5, slow = 4.3 .. 4.8
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32",
        "v_ashrrev_i32", "v_lshrrev_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mov_b32", "v_not_b32", "v_add_f32", "v_sub_f32", "v_accvgpr", "v_subb_co_u32", "v_nop", "v_min_i16")
COST_FAST, COST_SLOW = 2.5, 4.5


def main():
    argv = sys.argv[1:]
    jpath = None
    if argv and argv[0] == "--json":
        jpath, argv = argv[1], argv[2:]
    defs = [a for a in argv if a.startswith("-D")]      # e.g. -DR1_HEADLINE_ONLY for rdo_cand.hip
    argv = [a for a in argv if not a.startswith("-D")]
    src = argv[0]
    pats = argv[1:]
    jout = {"_model": {"fast_cycles": COST_FAST, "slow_cycles": COST_SLOW,
                       "source": "tools/ubench/valu_rate.hip on MI355X; cycles of the 2.4 GHz nominal clock "
                                 "per wave64 instruction per SIMD; static counts of the unrolled kernel"}}
    here = os.path.dirname(os.path.abspath(src))
    out = "/tmp/isa_hist_%d.s" % os.getpid()
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                           "--cuda-device-only", "-Wno-unused-function", "-Wno-pass-failed", "-I", here] + defs +
                          [src, "-o", out], stderr=subprocess.DEVNULL)
    kern, rows = None, collections.OrderedDict()
    meta = {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
            rows[kern] = collections.Counter()
            continue
        if kern is None:
            continue
        m = re.match(r"\s+\.(vgpr_count|sgpr_count|lds_size|private_segment_fixed_size)\s*:?\s*(\d+)", line)
        t = line.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            m2 = re.search(r"; (NumVgprs|ScratchSize|LDSByteSize|Occupancy): (\d+)", line)
            if m2:
                meta.setdefault(kern, {})[m2.group(1)] = int(m2.group(2))
            continue
        op = t.split()[0]
        if "dpp" in t or "sdwa" in t:
            op += "(dpp/sdwa)"
        rows[kern][op] += 1
    os.unlink(out)
    for k, c in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        if pats and not any(p in name for p in pats):
            continue
        if not c:
            continue
        valu = {o: n for o, n in c.items() if o.startswith("v_") and not o.startswith(("v_mfma", "v_readl", "v_readf"))}
        fast = sum(n for o, n in valu.items() if o.split("(")[0].replace("_e32", "").replace("_e64", "") in FAST and "(" not in o)
        slow = sum(valu.values()) - fast
        mfma = sum(n for o, n in c.items() if o.startswith("v_mfma"))
        lds = sum(n for o, n in c.items() if o.startswith("ds_"))
        vmem = sum(n for o, n in c.items() if o.startswith(("global_", "buffer_", "flat_", "scratch_")))
        salu = sum(n for o, n in c.items() if o.startswith("s_"))
        short = re.sub(r"^void \(anonymous namespace\)::", "", name).split("(")[0].replace(" ", "")
        jout[short] = {"valu": fast + slow, "fast": fast, "slow": slow, "mfma": mfma, "lds": lds, "vmem": vmem,
                       "salu": salu, "issue_cycles_per_valu": round((fast * COST_FAST + slow * COST_SLOW) /
                                                                    max(1, fast + slow), 4)}
        print("%s\n  VALU %d (fast %d, slow %d) -> %.0f issue cycles; MFMA %d, LDS %d, VMEM %d, SALU %d; %s"
              % (name[:150], fast + slow, fast, slow, fast * COST_FAST + slow * COST_SLOW, mfma, lds, vmem, salu,
                 meta.get(k, {})))
        top = sorted(valu.items(), key=lambda kv: -kv[1])[:14]
        print("  top VALU: " + ", ".join("%s %d" % kv for kv in top))
    if jpath:
        import json
        json.dump(jout, open(jpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
