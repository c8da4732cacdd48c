#!/usr/bin/env python3
"""Per-kernel resources of a built library: tools/kres.py rav1e_amd/librav1e_hip.so [name-substring]
Walks the clang offload bundles inside the .so, pulls the gfx950 code objects out and prints, per kernel,
VGPRs / AGPRs / LDS bytes / scratch bytes and the waves per SIMD the three allow (512 VGPRs per SIMD lane in
granules of 8, 160 KB of LDS per CU, one-wave workgroups assumed when a kernel's block size is 64)."""
import os, re, struct, subprocess, sys, tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    blob = open(path, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + 24)
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += 24


def main():
    lib = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
            name = g("name")
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"^void \(anonymous namespace\)::", "", dem).split("(")[0]
            vg, ag = int(g("vgpr_count")), int(blk.split("\n")[0].strip() or 0)
            lds, scr, wg = int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size")), int(g("max_flat_workgroup_size"))
            regs = max(vg + ag, 1)
            w_reg = min(8, 512 // ((regs + 7) // 8 * 8))
            waves_per_wg = max(1, (wg + 63) // 64)
            w_lds = 8 if lds == 0 else min(8, (160 * 1024 // lds) * waves_per_wg // 4)
            rows.append((dem, vg, ag, lds, scr, wg, w_reg, w_lds))
    for r in sorted(set(rows)):
        if pat in r[0]:
            print("%-58s vgpr %3d agpr %3d lds %6d scratch %4d block %4d | waves/SIMD by regs %d, by LDS %d" % r)


main()
