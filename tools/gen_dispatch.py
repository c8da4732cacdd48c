#!/usr/bin/env python3
"""Emit the per-table-entry dispatch symbols of the HIP backend:

  include/rav1e_amd_dispatch.h        declarations, one `extern "C"` symbol per entry of the
                                      reference's dispatch tables, with the reference's exact
                                      argument lists (so that a `CpuFeatureLevel::HIP` row is a
                                      table fill: `out[idx] = Some(rav1e_..._hip)`)
  rav1e_amd/csrc/dispatch_gen.inc     their definitions (thin calls into the generic shims of
                                      ctx.hip, which stage host blocks and launch the batch kernels)
  tests/c/dispatch_list.h             X-macro list of the same entries for tests/c/test_dispatch.c

Reference tables (file:line under /root/reference/src/asm/x86):
  SAD_FNS / SATD_FNS / *_HBD_FNS          dist/mod.rs:21-43,184-330,483-729   (22 BlockSizes)
  WEIGHTED_SSE_FNS / _HBD_FNS             dist/sse.rs:18-34,56-86
  CDEF_DIST_KERNEL_FNS                    dist/cdef_dist.rs:18-48
  PUT_FNS / PREP_FNS / AVG_FNS (+HBD)     mc.rs:17-84,371-620             (index get_2d_mode_idx)
  INV_TXFM_FNS / INV_TXFM_HBD_FNS         transform/inverse.rs:69-330     ([TxSize][TxType])
  CDEF_FILTER_FNS, CDEF_DIR_*_FNS         cdef.rs:16-37,160-191
  DEQUANTIZE_FNS                          quantize.rs:22-37
  rav1e_ipred_* / ipred_cfl* / cfl_ac_*   predict.rs:21-236               (no table: a match on mode / variant)
Symbol names follow the reference's pattern with the ISA suffix `hip`.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64),
          (64, 32), (64, 64), (64, 128), (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64),
          (64, 16)]
FILTERS = [("REGULAR", 0, "regular"), ("SMOOTH", 1, "smooth"), ("SHARP", 2, "sharp")]
TX_DIMS = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16),
           (32, 64), (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
# TxType -> (TYPE1, TYPE2) as in impl_itx_fns! (transform/inverse.rs:215-262); the symbol is
# rav1e_inv_txfm_add_<TYPE2>_<TYPE1>_<W>x<H>_<bpc>bpc_<isa>
TXT = {0: ("dct", "dct"), 1: ("adst", "dct"), 2: ("dct", "adst"), 3: ("adst", "adst"),
       4: ("flipadst", "dct"), 5: ("dct", "flipadst"), 6: ("flipadst", "flipadst"),
       7: ("adst", "flipadst"), 8: ("flipadst", "adst"), 9: ("identity", "identity"),
       10: ("dct", "identity"), 11: ("identity", "dct"), 12: ("adst", "identity"),
       13: ("identity", "adst"), 14: ("flipadst", "identity"), 15: ("identity", "flipadst"),
       16: ("wht", "wht")}


def valid_tx(ts, tt):
    w, h = TX_DIMS[ts]
    m = max(w, h)
    if tt == 16:
        return (w, h) == (4, 4)
    if m == 64:
        return tt == 0
    if m == 32:
        return tt in (0, 9)
    return True


def main(out_root=None):
    """writes the three files under `out_root` (default: the repository)"""
    out_root = out_root or ROOT
    decl, defs, xl = [], [], []

    def add(kind, name, cdecl, body, xargs):
        decl.append("%s;" % cdecl)
        defs.append('extern "C" %s { %s }' % (cdecl, body))
        xl.append("X_%s(%s, %s)" % (kind, name, xargs))

    # ---- dist
    for (w, h) in BLOCKS:
        for (fam, kind, sym) in (("SAD", "R1_DIST_SAD", "sad"), ("SATD", "R1_DIST_SATD", "satd")):
            n8 = "rav1e_%s%s%dx%d_hip" % (sym, "" if sym == "sad" else "_", w, h)
            add(fam, n8, "uint32_t %s(const uint8_t *src, ptrdiff_t src_stride, const uint8_t *dst, "
                "ptrdiff_t dst_stride)" % n8,
                "return dist_shim(%s, src, src_stride, dst, dst_stride, %d, %d, 1, 8);" % (kind, w, h),
                "%d, %d" % (w, h))
            n16 = "rav1e_%s_%dx%d_hbd_hip" % (sym, w, h)
            if sym == "sad":
                add("SAD_HBD", n16, "uint32_t %s(const uint16_t *src, ptrdiff_t src_stride, const uint16_t *dst, "
                    "ptrdiff_t dst_stride)" % n16,
                    "return dist_shim(%s, src, src_stride, dst, dst_stride, %d, %d, 2, 10);" % (kind, w, h),
                    "%d, %d" % (w, h))
            else:
                add("SATD_HBD", n16, "uint32_t %s(const uint16_t *src, ptrdiff_t src_stride, const uint16_t *dst, "
                    "ptrdiff_t dst_stride, uint32_t bdmax)" % n16,
                    "return dist_shim(%s, src, src_stride, dst, dst_stride, %d, %d, 2, bd_from_max((int)bdmax));"
                    % (kind, w, h), "%d, %d" % (w, h))
        n = "rav1e_weighted_sse_%dx%d_hip" % (w, h)
        add("WSSE", n, "uint64_t %s(const uint8_t *src, ptrdiff_t src_stride, const uint8_t *dst, ptrdiff_t "
            "dst_stride, const uint32_t *scale, ptrdiff_t scale_stride)" % n,
            "return wsse_shim(src, src_stride, dst, dst_stride, scale, scale_stride, %d, %d, 1);" % (w, h),
            "%d, %d" % (w, h))
        n = "rav1e_weighted_sse_%dx%d_hbd_hip" % (w, h)
        add("WSSE_HBD", n, "uint64_t %s(const uint16_t *src, ptrdiff_t src_stride, const uint16_t *dst, "
            "ptrdiff_t dst_stride, const uint32_t *scale, ptrdiff_t scale_stride)" % n,
            "return wsse_shim(src, src_stride, dst, dst_stride, scale, scale_stride, %d, %d, 2);" % (w, h),
            "%d, %d" % (w, h))
    for (w, h) in ((4, 4), (4, 8), (8, 4), (8, 8)):
        n = "rav1e_cdef_dist_kernel_%dx%d_hip" % (w, h)
        add("CDK", n, "void %s(const uint8_t *src, ptrdiff_t src_stride, const uint8_t *dst, ptrdiff_t dst_stride, "
            "uint32_t *ret_ptr)" % n,
            "cdef_dist_kernel_shim(src, src_stride, dst, dst_stride, %d, %d, 1, ret_ptr);" % (w, h), "%d, %d" % (w, h))
        n = "rav1e_cdef_dist_kernel_%dx%d_hbd_hip" % (w, h)
        add("CDK_HBD", n, "void %s(const uint16_t *src, ptrdiff_t src_stride, const uint16_t *dst, ptrdiff_t "
            "dst_stride, uint32_t *ret_ptr)" % n,
            "cdef_dist_kernel_shim(src, src_stride, dst, dst_stride, %d, %d, 2, ret_ptr);" % (w, h), "%d, %d" % (w, h))
    # ---- mc
    pairs = [(fx, fy) for fx in FILTERS for fy in FILTERS] + [(("BILINEAR", 3, "bilin"), ("BILINEAR", 3, "bilin"))]
    for (fx, fy) in pairs:
        if fx[1] == 3:
            stem = "bilin"
        else:
            stem = "8tap_" + (fx[2] if fx[1] == fy[1] else "%s_%s" % (fx[2], fy[2]))
        for bpc, T in ((8, "uint8_t"), (16, "uint16_t")):
            hb = bpc == 16
            n = "rav1e_put_%s_%dbpc_hip" % (stem, bpc)
            add("PUT_HBD" if hb else "PUT", n,
                "void %s(%s *dst, ptrdiff_t dst_stride, const %s *src, ptrdiff_t src_stride, int32_t w, int32_t h, "
                "int32_t mx, int32_t my%s)" % (n, T, T, ", int32_t bitdepth_max" if hb else ""),
                "put_shim(dst, dst_stride, src, src_stride, w, h, mx, my, %d, %d, %d, %s);"
                % (fx[1], fy[1], 2 if hb else 1, "bd_from_max(bitdepth_max)" if hb else "8"),
                "%d, %d" % (fx[1], fy[1]))
            n = "rav1e_prep_%s_%dbpc_hip" % (stem, bpc)
            add("PREP_HBD" if hb else "PREP", n,
                "void %s(int16_t *tmp, const %s *src, ptrdiff_t src_stride, int32_t w, int32_t h, int32_t mx, "
                "int32_t my%s)" % (n, T, ", int32_t bitdepth_max" if hb else ""),
                "prep_shim(tmp, src, src_stride, w, h, mx, my, %d, %d, %d, %s);"
                % (fx[1], fy[1], 2 if hb else 1, "bd_from_max(bitdepth_max)" if hb else "8"),
                "%d, %d" % (fx[1], fy[1]))
    for bpc, T in ((8, "uint8_t"), (16, "uint16_t")):
        hb = bpc == 16
        n = "rav1e_avg_%dbpc_hip" % bpc
        add("AVG_HBD" if hb else "AVG", n,
            "void %s(%s *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2, int32_t w, int32_t h%s)"
            % (n, T, ", int32_t bitdepth_max" if hb else ""),
            "avg_shim(dst, dst_stride, tmp1, tmp2, w, h, %d, %s);" % (2 if hb else 1,
                                                                     "bd_from_max(bitdepth_max)" if hb else "8"), "0, 0")
    # ---- inverse transforms
    for ts, (w, h) in enumerate(TX_DIMS):
        for tt in range(17):
            if not valid_tx(ts, tt):
                continue
            t1, t2 = TXT[tt]
            n = "rav1e_inv_txfm_add_%s_%s_%dx%d_8bpc_hip" % (t2, t1, w, h)
            add("ITX", n, "void %s(uint8_t *dst, ptrdiff_t dst_stride, int16_t *coeff, int32_t eob)" % n,
                "(void)eob; inv_shim_abort(dst, dst_stride, coeff, %d, %d, 1, 8);" % (ts, tt), "%d, %d" % (ts, tt))
            for bpc in (10, 12):
                n = "rav1e_inv_txfm_add_%s_%s_%dx%d_%dbpc_hip" % (t2, t1, w, h, bpc)
                add("ITX_HBD", n, "void %s(uint16_t *dst, ptrdiff_t dst_stride, int16_t *coeff, int32_t eob, "
                    "int32_t bitdepth_max)" % n,
                    "(void)eob; inv_shim_abort(dst, dst_stride, coeff, %d, %d, 2, bd_from_max(bitdepth_max));"
                    % (ts, tt), "%d, %d, %d" % (ts, tt, bpc))
    # ---- CDEF
    for (w, h, xd, yd) in ((4, 4, 1, 1), (4, 8, 1, 0), (8, 8, 0, 0)):
        n = "rav1e_cdef_filter_%dx%d_hip" % (w, h)
        add("CDEFF", n, "void %s(uint8_t *dst, ptrdiff_t dst_stride, const uint16_t *tmp, ptrdiff_t tmp_stride, "
            "int32_t pri_strength, int32_t sec_strength, int32_t dir, int32_t damping)" % n,
            "cdef_filter_shim(dst, dst_stride, tmp, tmp_stride, pri_strength, sec_strength, dir, damping, %d, %d, 8, 1);"
            % (xd, yd), "%d, %d" % (xd, yd))
        n = "rav1e_cdef_filter_%dx%d_16bpc_hip" % (w, h)
        add("CDEFF_HBD", n, "void %s(uint16_t *dst, ptrdiff_t dst_stride, const uint16_t *tmp, ptrdiff_t tmp_stride, "
            "int32_t pri_strength, int32_t sec_strength, int32_t dir, int32_t damping, int32_t bitdepth_max)" % n,
            "cdef_filter_shim(dst, dst_stride, tmp, tmp_stride, pri_strength, sec_strength, dir, damping, %d, %d, "
            "bd_from_max(bitdepth_max), 2);" % (xd, yd), "%d, %d" % (xd, yd))
    add("CDEFD", "rav1e_cdef_dir_8bpc_hip",
        "int32_t rav1e_cdef_dir_8bpc_hip(const uint8_t *tmp, ptrdiff_t tmp_stride, uint32_t *var)",
        "return cdef_dir_shim(tmp, tmp_stride, var, 1, 8);", "0, 0")
    add("CDEFD_HBD", "rav1e_cdef_dir_16bpc_hip",
        "int32_t rav1e_cdef_dir_16bpc_hip(const uint16_t *tmp, ptrdiff_t tmp_stride, uint32_t *var, int32_t bitdepth_max)",
        "return cdef_dir_shim(tmp, tmp_stride, var, 2, bd_from_max(bitdepth_max));", "0, 0")
    # ---- predict:: (src/asm/x86/predict.rs:21-236; consumed by dispatch_predict_intra :239-857 and
    # pred_cfl_ac :873-927).  `angle` of z1/z2/z3 carries the edge-filter flags in its unused bits
    # (bit 10 enable_ief, bit 9 smooth neighbour, :301-303); z2 receives dx / dy = distance of the
    # block to the frame edge in a frame rounded up to 8 px (:306-316); the HBD forms receive
    # (max_width, max_height, bit_depth_max), z2 dx / dy in the max_* slots (:584-625).
    ANG = [("dc", 0, 3), ("dc_128", 0, 0), ("dc_left", 0, 1), ("dc_top", 0, 2), ("v", 1, 0), ("h", 2, 0),
           ("smooth", 9, 0), ("smooth_v", 10, 0), ("smooth_h", 11, 0), ("paeth", 12, 0),
           ("z1", 3, 0), ("z3", 7, 0)]
    for (nm, mode, variant) in ANG:
        n = "rav1e_ipred_%s_8bpc_hip" % nm
        add("IPRED", n, "void %s(uint8_t *dst, ptrdiff_t stride, const uint8_t *topleft, int width, int height, "
            "int angle)" % n,
            "ipred_shim(dst, stride, topleft, width, height, angle, %d, %d, 0, 0, nullptr, 1, 8);" % (mode, variant),
            "%d, %d" % (mode, variant))
        n = "rav1e_ipred_%s_16bpc_hip" % nm
        add("IPRED_HBD", n, "void %s(uint16_t *dst, ptrdiff_t stride, const uint16_t *topleft, int width, "
            "int height, int angle, int max_width, int max_height, int bit_depth_max)" % n,
            "(void)max_width; (void)max_height; ipred_shim(dst, stride, topleft, width, height, angle, %d, %d, 0, 0, "
            "nullptr, 2, bd_from_max(bit_depth_max));" % (mode, variant), "%d, %d" % (mode, variant))
    add("IPRED_Z2", "rav1e_ipred_z2_8bpc_hip",
        "void rav1e_ipred_z2_8bpc_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *topleft, int width, int height, "
        "int angle, int dx, int dy)",
        "ipred_shim(dst, stride, topleft, width, height, angle, 4, 0, dx, dy, nullptr, 1, 8);", "4, 0")
    add("IPRED_Z2_HBD", "rav1e_ipred_z2_16bpc_hip",
        "void rav1e_ipred_z2_16bpc_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *topleft, int width, "
        "int height, int angle, int dx, int dy, int bit_depth_max)",
        "ipred_shim(dst, stride, topleft, width, height, angle, 4, 0, dx, dy, nullptr, 2, bd_from_max(bit_depth_max));",
        "4, 0")
    for (nm, variant) in (("cfl", 3), ("cfl_128", 0), ("cfl_left", 1), ("cfl_top", 2)):
        n = "rav1e_ipred_%s_8bpc_hip" % nm
        add("CFL", n, "void %s(uint8_t *dst, ptrdiff_t stride, const uint8_t *topleft, int width, int height, "
            "const int16_t *ac, int alpha)" % n,
            "ipred_shim(dst, stride, topleft, width, height, alpha, 13, %d, 0, 0, ac, 1, 8);" % variant,
            "13, %d" % variant)
        n = "rav1e_ipred_%s_16bpc_hip" % nm
        add("CFL_HBD", n, "void %s(uint16_t *dst, ptrdiff_t stride, const uint16_t *topleft, int width, int height, "
            "const int16_t *ac, int alpha, int bit_depth_max)" % n,
            "ipred_shim(dst, stride, topleft, width, height, alpha, 13, %d, 0, 0, ac, 2, bd_from_max(bit_depth_max));"
            % variant, "13, %d" % variant)
    for (nm, xd, yd) in (("420", 1, 1), ("422", 1, 0), ("444", 0, 0)):
        n = "rav1e_ipred_cfl_ac_%s_8bpc_hip" % nm
        add("CFLAC", n, "void %s(int16_t *ac, const uint8_t *src, ptrdiff_t stride, int w_pad, int h_pad, int width, "
            "int height)" % n,
            "cfl_ac_shim(ac, src, stride, w_pad, h_pad, width, height, %d, %d, 1, 8);" % (xd, yd), "%d, %d" % (xd, yd))
        n = "rav1e_ipred_cfl_ac_%s_16bpc_hip" % nm
        add("CFLAC_HBD", n, "void %s(int16_t *ac, const uint16_t *src, ptrdiff_t stride, int w_pad, int h_pad, "
            "int width, int height)" % n,
            "cfl_ac_shim(ac, src, stride, w_pad, h_pad, width, height, %d, %d, 2, 10);" % (xd, yd), "%d, %d" % (xd, yd))
    # ---- dequantize (Rust-internal fn type in the reference; C spelling of the same arguments)
    add("DEQ", "rav1e_dequantize_hip",
        "void rav1e_dequantize_hip(uint8_t qindex, const int16_t *coeffs_ptr, uint16_t eob, int16_t *rcoeffs_ptr, "
        "uint8_t tx_size, size_t bit_depth, int8_t dc_delta_q, int8_t ac_delta_q)",
        "(void)eob; dequant_shim(qindex, coeffs_ptr, rcoeffs_ptr, tx_size, (int)bit_depth, dc_delta_q, ac_delta_q);",
        "0, 0")

    hdr = ["/* GENERATED by tools/gen_dispatch.py -- do not edit.",
           " * One symbol per entry of the reference's x86 dispatch tables, with the reference's exact C ABI",
           " * (host pointers, strides in BYTES exactly as the asm receives them, values returned the way the",
           " * asm returns them).  %d symbols.  See the generator's docstring for the table locations." % len(decl),
           " * Two families have no asm counterpart in the reference and are extensions in the same style:",
           " * rav1e_cdef_dist_kernel_*_hbd_hip (the reference's HBD entry is a Rust fn returning a tuple,",
           " * src/asm/x86/dist/cdef_dist.rs:26-48; here the LBD asm's ret_ptr convention) and rav1e_dequantize_hip",
           " * (a Rust-internal fn type, src/asm/x86/quantize.rs:22-31, in C spelling). */",
           "#ifndef RAV1E_AMD_DISPATCH_H", "#define RAV1E_AMD_DISPATCH_H", "#include <stddef.h>", "#include <stdint.h>",
           "#ifdef __cplusplus", 'extern "C" {', "#endif"] + decl + ["#ifdef __cplusplus", "}", "#endif", "#endif"]
    for d in ("include", os.path.join("rav1e_amd", "csrc"), os.path.join("tests", "c")):
        os.makedirs(os.path.join(out_root, d), exist_ok=True)
    open(os.path.join(out_root, "include", "rav1e_amd_dispatch.h"), "w").write("\n".join(hdr) + "\n")
    open(os.path.join(out_root, "rav1e_amd", "csrc", "dispatch_gen.inc"), "w").write(
        "/* GENERATED by tools/gen_dispatch.py -- do not edit. */\n" + "\n".join(defs) + "\n")
    open(os.path.join(out_root, "tests", "c", "dispatch_list.h"), "w").write(
        "/* GENERATED by tools/gen_dispatch.py -- do not edit. */\n" + "\n".join(xl) + "\n")
    print(len(decl), "symbols")


if __name__ == "__main__":
    import sys
    main(sys.argv[1] if len(sys.argv) > 1 else None)
