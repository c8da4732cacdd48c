"""ctypes loader for the CPU oracle (oracle/libr1oracle.so) -- test infrastructure.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "libr1oracle.so")


class Plane(C.Structure):
    """Layout shared by R1Plane (include/rav1e_amd.h) and r1o_plane."""
    _fields_ = [("data", C.c_void_p), ("stride", C.c_int32), ("alloc_height", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("xorigin", C.c_int32),
                ("yorigin", C.c_int32), ("bytes_per_px", C.c_int32), ("bit_depth", C.c_int32)]


DIST_CAND = np.dtype([("ox", "<i2"), ("oy", "<i2"), ("rx", "<i2"), ("ry", "<i2")])
MC_CAND = np.dtype([("rx", "<i2"), ("ry", "<i2"), ("col_frac", "u1"), ("row_frac", "u1"),
                    ("mode_x", "u1"), ("mode_y", "u1")])
RDO_CAND = np.dtype([("ox", "<i2"), ("oy", "<i2"), ("rx", "<i2"), ("ry", "<i2"),
                     ("col_frac", "u1"), ("row_frac", "u1"), ("mode_x", "u1"), ("mode_y", "u1"),
                     ("tx_type", "u1"), ("reserved", "u1", (3,))])
assert DIST_CAND.itemsize == 8 and MC_CAND.itemsize == 8 and RDO_CAND.itemsize == 16


def build(force=False):
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h", ".inc"))]
    if force or not os.path.exists(SO) or any(
            os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["make", "-C", ODIR, "-s", "-B"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        build()
    L = C.CDLL(SO)
    vp, i, sz, pd = C.c_void_p, C.c_int, C.c_size_t, C.c_ssize_t
    sigs = {
        "r1o_get_sad": (C.c_uint32, [vp, pd, vp, pd, i, i, i]),
        "r1o_get_satd": (C.c_uint32, [vp, pd, vp, pd, i, i, i]),
        "r1o_get_weighted_sse": (C.c_uint64, [vp, pd, vp, pd, vp, sz, i, i, i]),
        "r1o_cdef_dist_kernel": (C.c_uint32, [vp, pd, vp, pd, i, i, i, i]),
        "r1o_apply_ssim_boost": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, i]),
        "r1o_cdef_dist_wxh": (C.c_uint64, [vp, pd, vp, pd, i, i, i, i, vp, sz]),
        "r1o_put_8tap": (None, [vp, pd, vp, pd, i, i, i, i, i, i, i, i]),
        "r1o_prep_8tap": (None, [vp, vp, pd, i, i, i, i, i, i, i, i]),
        "r1o_mc_avg": (None, [vp, pd, vp, vp, i, i, i, i]),
        "r1o_forward_transform": (i, [vp, vp, sz, i, i, i, i]),
        "r1o_tx_width": (i, [i]), "r1o_tx_height": (i, [i]),
        "r1o_valid_av1_transform": (i, [i, i]),
        "r1o_fwd_txfm_1d": (None, [vp, i]),
        "r1o_inv_txfm_1d": (i, [vp, i, i, i]),
        "r1o_inverse_transform_add": (i, [vp, vp, pd, i, i, i, i, i]),
        "r1o_inv_txfm_add_batch": (i, [vp, i, vp, vp, i, i, i, i, i, i]),
        "r1o_gen_scan": (None, [i, i, i, vp]),
        "r1o_scan_kind": (i, [i]),
        "r1o_get_scan": (i, [i, i, vp, vp]),
        "r1o_get_log_tx_scale": (i, [i]),
        "r1o_dc_q": (C.c_uint16, [i, i, i]),
        "r1o_ac_q": (C.c_uint16, [i, i, i]),
        "r1o_divu": (C.c_uint32, [C.c_uint32, C.c_uint32]),
        "r1o_quantize": (i, [vp, vp, i, i, i, i, i, i, i, i]),
        "r1o_dequantize": (None, [vp, vp, i, i, i, i, i, i]),
        "r1o_quantize_batch": (i, [vp, i, i, i, i, i, i, i, i, i, i, vp, vp, vp]),
        "r1o_intra_mode_to_angle": (i, [i]),
        "r1o_select_ief_strength": (i, [i, i, i, i]),
        "r1o_select_ief_upsample": (i, [i, i, i, i]),
        "r1o_dispatch_predict_intra": (i, [i, i, vp, pd, i, i, vp, i, i, vp, i, i, i, i, i]),
        "r1o_predict_intra": (i, [i, i, i, vp, pd, i, i, vp, i, i, i, vp, i, i, i, i, i]),
        "r1o_pred_cfl_ac": (None, [vp, vp, pd, i, i, i, i, i, i, i]),
        "r1o_get_intra_edges": (None, [vp, vp, vp, pd, i, i, i, i, i, i, i, i, i, i, i, i]),
        "r1o_cdef_find_dir": (i, [vp, pd, vp, i, i]),
        "r1o_cdef_filter_block": (None, [vp, pd, vp, pd, i, i, i, i, i, i, i, i, i]),
        "r1o_cdef_adjust_strength": (i, [i, i]),
        "r1o_cdef_filter_tile_plane": (None, [vp, vp, vp, i, i, i, i, i, vp, i, i, i, vp, i, vp, vp,
                                              i, i]),
        "r1o_estimate_intra_costs": (None, [vp, i, vp]),
        "r1o_importance_block_difference": (C.c_uint64, [vp, vp]),
        "r1o_estimate_inter_costs": (None, [vp, vp, vp, vp]),
        "r1o_update_block_importances": (None, [vp, vp, vp, vp, i, i, i, vp]),
        "r1o_tx_domain_distortion": (C.c_uint64, [vp, vp, i, i]),
        "r1o_estimate_rate": (C.c_uint64, [i, i, C.c_uint64]),
        "r1o_quantize_rdo_batch": (i, [vp, i, i, i, i, i, i, i, i, i, i, vp, vp, vp, vp, vp]),
        "r1o_diff": (None, [vp, vp, pd, vp, pd, i, i, i]),
        "r1o_set_threads": (None, [i]),
        "r1o_dist_batch": (i, [i, vp, vp, i, i, vp, i, vp]),
        "r1o_dist_scaled_batch": (i, [i, vp, vp, i, i, vp, i, vp, i, i, i, vp]),
        "r1o_fwd_txfm_batch": (i, [vp, vp, i, i, i, i, i]),
        "r1o_mc_put_batch": (i, [vp, i, i, vp, i, vp]),
        "r1o_mc_prep_batch": (i, [vp, i, i, vp, i, vp]),
        "r1o_mc_avg_batch": (i, [vp, vp, i, i, i, i, i, vp]),
        "r1o_rdo_cand_batch": (i, [vp, vp, i, i, i, vp, i, vp, vp, vp, vp]),
        "r1o_plane_pad": (None, [vp, i, i, i, i]),
        "r1o_plane_downsample": (i, [vp, vp, i, i, i, i]),
        "r1o_fast_rdo_cand_batch": (i, [vp, vp, i, i, vp, i, i, vp, vp, vp]),
        "r1o_fast512_rdo_cand_batch": (i, [vp, vp, i, i, vp, i, i, vp, vp, vp]),
        "r1o_fast512_available": (i, []),
        "r1o_estimate_tile_motion": (i, [vp, vp, vp, vp, vp]),
        "r1o_rdo_pixel_cand_batch": (i, [vp, vp, i, i, i, vp, i, i, i, i, i, i, vp, i, i, i,
                                         vp, vp, vp, vp, vp, vp, vp]),
        "r1o_tx_type_mask": (C.c_uint32, [i, i, i, i]),
        "r1o_rdo_txsearch_batch": (i, [vp, vp, vp, i, i, i, vp, i, C.c_uint32, i, i, i, i, i, vp, i, i, i,
                                       vp, vp, vp, vp, vp, vp, vp]),
        "r1o_deblock_plane": (i, [vp, vp, i, i, i, vp, i, i, i, i, i, i]),
        "r1o_deblock_sse_plane": (i, [vp, vp, i, i, i, vp, i, i, i, i, i, i, vp, vp]),
        "r1o_deblock_pick_levels": (None, [vp, vp, i, vp]),
        "r1o_activity_scales": (None, [vp, vp, vp]),
        "r1o_lrf_filter_plane": (i, [vp, vp, vp, i, i, i, i, i, i, i, i, vp, i]),
        "r1o_sgrproj_solve": (None, [vp, vp, i, i, i, i, i, i, i, vp]),
        "r1o_lrf_search_unit": (i, [vp, vp, i, i, i, i, i, i, i, i, i, vp, i, C.c_uint32, i, vp, vp]),
        "r1o_estimate_motion_batch": (i, [vp, vp, vp, vp, vp, vp, i, i, i, vp]),
        "r1o_rdo_full_cand_batch": (i, [vp, vp, i, i, i, vp, i, i, i, i, i, vp, vp, vp, vp, vp, vp]),
        "r1o_cdef_strength_search": (i, [vp, vp, vp, i, i, i, vp, i, vp, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class HostPlane:
    """v_frame 0.3.9 Plane<T> layout (PlaneConfig::new): xorigin and stride are
    rounded up so that row starts are 64-byte aligned; element (x, y) lives at
    data[(yorigin + y) * stride + xorigin + x] (src/tiling/plane_region.rs:185).
    rav1e frames use xpad = ypad = 88 for luma (src/frame/mod.rs:22-23)."""

    def __init__(self, width, height, bit_depth=8, xpad=88, ypad=88, fill=None, rng=None):
        self.bpp = 1 if bit_depth == 8 else 2
        al = 64 // self.bpp
        self.width, self.height, self.bit_depth = width, height, bit_depth
        self.xpad, self.ypad = xpad, ypad
        self.xorigin = (xpad + al - 1) // al * al
        self.yorigin = ypad
        self.stride = (self.xorigin + width + xpad + al - 1) // al * al
        self.alloc_height = self.yorigin + height + ypad
        dt = np.uint8 if self.bpp == 1 else np.uint16
        if rng is not None:
            self.data = rng.integers(0, 1 << bit_depth, size=(self.alloc_height, self.stride),
                                     dtype=dt)
        else:
            self.data = np.full((self.alloc_height, self.stride), fill or 0, dtype=dt)

    def view(self):
        """visible area as an array view"""
        return self.data[self.yorigin:self.yorigin + self.height,
                         self.xorigin:self.xorigin + self.width]

    def block_ptr(self, x, y):
        off = ((self.yorigin + y) * self.stride + self.xorigin + x) * self.bpp
        return C.c_void_p(self.data.ctypes.data + off)

    def cstruct(self, data_ptr=None):
        return Plane(data_ptr if data_ptr is not None else self.data.ctypes.data, self.stride,
                     self.alloc_height, self.width, self.height, self.xorigin, self.yorigin,
                     self.bpp, self.bit_depth)


# ---- motion estimation (oracle/me.c) ----
class MeParams(C.Structure):
    """r1o_me_params"""
    _fields_ = [("w_in_b", C.c_int32), ("h_in_b", C.c_int32), ("stats_cols", C.c_int32),
                ("stats_rows", C.c_int32), ("tile_x", C.c_int32), ("tile_y", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("bit_depth", C.c_int32),
                ("allow_hp", C.c_int32), ("allow_full_search", C.c_int32),
                ("me_range_scale", C.c_int32), ("lambda_", C.c_uint32 * 3)]


ME_BLOCK_CAND = np.dtype([("bx", "<i2"), ("by", "<i2"), ("w", "u1"), ("h", "u1"), ("corner", "u1"),
                          ("reserved", "u1"), ("pmv", "<i2", (2, 2))])
ME_RESULT = np.dtype([("row", "<i2"), ("col", "<i2"), ("sad", "<u4"), ("cost", "<u8")])
LRF_UNIT = np.dtype([("filter", "u1"), ("set", "u1"), ("xqd", "i1", (2,))])
ME_STATS = np.dtype([("row", "<i2"), ("col", "<i2"), ("normalized_sad", "<u4")])
assert ME_STATS.itemsize == 8


def plane_from_image(img, bit_depth, xpad, ypad):
    """HostPlane whose padding replicates the edge pixels (Plane::pad)."""
    h, w = img.shape
    p = HostPlane(w, h, bit_depth, xpad, ypad)
    p.data[...] = np.pad(img.astype(p.data.dtype),
                         ((p.yorigin, p.alloc_height - p.yorigin - h),
                          (p.xorigin, p.stride - p.xorigin - w)), mode="edge")
    return p


def box_down2(img):
    """2x2 box filter with rounding (what Plane::downsampled computes inside the frame)."""
    h, w = img.shape
    e = np.pad(img.astype(np.int64), ((0, h & 1), (0, w & 1)), mode="edge")
    return ((e[0::2, 0::2] + e[0::2, 1::2] + e[1::2, 0::2] + e[1::2, 1::2] + 2) >> 2)


def me_pyramid(img, bit_depth):
    """[full, half, quarter] HostPlanes of one luma image, padding 88 / 44 / 22."""
    h1 = box_down2(img)
    h2 = box_down2(h1)
    return [plane_from_image(img, bit_depth, 88, 88), plane_from_image(h1, bit_depth, 44, 44),
            plane_from_image(h2, bit_depth, 22, 22)]


def me_oracle(L, org3, ref3, w_in_b, h_in_b, tile, bit_depth, lambdas, stats, prev=None,
              allow_hp=1, allow_full_search=0, me_range_scale=1):
    """r1o_estimate_tile_motion on HostPlane pyramids; stats: ME_STATS array (rows, cols), in place."""
    po = (Plane * 3)(*[p.cstruct() for p in org3])
    pr = (Plane * 3)(*[p.cstruct() for p in ref3])
    rows, cols = stats.shape
    prm = MeParams(w_in_b, h_in_b, cols, rows, tile[0], tile[1], tile[2], tile[3], bit_depth,
                   allow_hp, allow_full_search, me_range_scale, (C.c_uint32 * 3)(*lambdas))
    rc = L.r1o_estimate_tile_motion(po, pr, C.byref(prm), stats.ctypes.data,
                                    prev.ctypes.data if prev is not None else None)
    assert rc == 0
    return stats


def me_block_oracle(L, org3, ref3, w_in_b, h_in_b, tile, bit_depth, lambdas, stats, prev, cands,
                    use_satd=1, filter_mode=0, allow_hp=1):
    po = (Plane * 3)(*[p.cstruct() for p in org3])
    pr = (Plane * 3)(*[p.cstruct() for p in ref3])
    rows, cols = stats.shape
    prm = MeParams(w_in_b, h_in_b, cols, rows, tile[0], tile[1], tile[2], tile[3], bit_depth,
                   allow_hp, 0, 1, (C.c_uint32 * 3)(*lambdas))
    out = np.zeros(len(cands), ME_RESULT)
    rc = L.r1o_estimate_motion_batch(po, pr, C.byref(prm), stats.ctypes.data,
                                     prev.ctypes.data if prev is not None else None,
                                     cands.ctypes.data, len(cands), use_satd, filter_mode,
                                     out.ctypes.data)
    assert rc == 0
    return out


class CdefSearchParams(C.Structure):
    """r1o_cdef_search_params == R1CdefSearchParams"""
    _fields_ = [("y_strengths", C.c_uint8 * 8), ("uv_strengths", C.c_uint8 * 8),
                ("damping", C.c_int32), ("bit_depth", C.c_int32), ("n_idx", C.c_int32),
                ("planes", C.c_int32), ("xdec", C.c_int32), ("ydec", C.c_int32),
                ("crop_w", C.c_int32), ("crop_h", C.c_int32), ("area_sb_w", C.c_int32),
                ("area_sb_h", C.c_int32), ("dist_scale", C.c_uint32 * 3)]


def cdef_search_case(G, name, pad=16):
    """planes + parameters of one cdef_search_ref.npz case -> (rec HostPlanes, src HostPlanes, skip,
    scales, CdefSearchParams, want_err, want_best)"""
    W, H, xdec, ydec, bd, damping, n_idx, asw, ash, planes = [int(v) for v in G[name + "_meta"]]
    rec = [plane_from_image(G["%s_rec%d" % (name, p)].astype(np.int64), bd, pad, pad) for p in range(3)]
    src = [plane_from_image(G["%s_src%d" % (name, p)].astype(np.int64), bd, pad, pad) for p in range(3)]
    prm = CdefSearchParams()
    prm.y_strengths[:] = [int(v) for v in G[name + "_ystr"]]
    prm.uv_strengths[:] = [int(v) for v in G[name + "_uvstr"]]
    prm.damping, prm.bit_depth, prm.n_idx, prm.planes = damping, bd, n_idx, planes
    prm.xdec, prm.ydec, prm.crop_w, prm.crop_h, prm.area_sb_w, prm.area_sb_h = xdec, ydec, W, H, asw, ash
    prm.dist_scale[:] = [int(v) for v in G[name + "_dscale"]]
    return (rec, src, np.ascontiguousarray(G[name + "_skip"]), np.ascontiguousarray(G[name + "_scales"]), prm,
            G[name + "_err"], G[name + "_best"])


def cdef_search_oracle(L, G, name):
    rec, src, skip, scales, prm, want_err, want_best = cdef_search_case(G, name)
    pr = (Plane * 3)(*[p.cstruct() for p in rec])
    ps = (Plane * 3)(*[p.cstruct() for p in src])
    err = np.zeros_like(want_err)
    best = np.zeros_like(want_best)
    rc = L.r1o_cdef_strength_search(pr, ps, skip.ctypes.data, skip.shape[1], skip.shape[1], skip.shape[0],
                                    scales.ctypes.data, scales.shape[1], C.byref(prm), err.ctypes.data,
                                    best.ctypes.data)
    assert rc == 0
    return err, best, want_err, want_best
