"""Host-side mirror of the reference's kernel interface over the batch C ABI.

Function names and argument meaning follow the reference (src/dist.rs,
src/mc.rs, src/transform/forward.rs); the difference is that every call takes a
whole candidate list.  All tensors live on the GPU (torch is the allocator and
stream provider only); descriptors are packed as the C structs of
include/rav1e_amd.h.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .types import TX_DIMS, valid_av1_transform

DIST_CAND = np.dtype([("ox", "<i2"), ("oy", "<i2"), ("rx", "<i2"), ("ry", "<i2")])
MC_CAND = np.dtype([("rx", "<i2"), ("ry", "<i2"), ("col_frac", "u1"), ("row_frac", "u1"),
                    ("mode_x", "u1"), ("mode_y", "u1")])
INTRA_EDGE_CAND = np.dtype([("x", "<i2"), ("y", "<i2"), ("mode", "i1"), ("angle_delta", "i1"),
                            ("flags", "u1"), ("reserved", "u1")])
INTRA_CAND = np.dtype([("mode", "u1"), ("variant", "u1"), ("angle", "<i2"), ("ief", "u1"),
                       ("avail_w", "u1"), ("avail_h", "u1"), ("reserved", "u1")])
CFL_AC_CAND = np.dtype([("x", "<i2"), ("y", "<i2"), ("w_pad", "u1"), ("h_pad", "u1"),
                        ("reserved", "u1", (2,))])
CDEF_DIR_CAND = np.dtype([("x", "<i2"), ("y", "<i2")])
CDEF_BLOCK_CAND = np.dtype([("x", "<i2"), ("y", "<i2"), ("pri_strength", "<i2"),
                            ("sec_strength", "<i2"), ("dir", "u1"), ("damping", "u1"),
                            ("edges", "u1"), ("reserved", "u1")])
EDGE_LEN = 257
RDO_CAND = np.dtype([("ox", "<i2"), ("oy", "<i2"), ("rx", "<i2"), ("ry", "<i2"),
                     ("col_frac", "u1"), ("row_frac", "u1"), ("mode_x", "u1"), ("mode_y", "u1"),
                     ("tx_type", "u1"), ("reserved", "u1", (3,))])


class R1Error(RuntimeError):
    pass


class Plane:
    """Plane<T> with v_frame 0.3.9's PlaneConfig layout, resident in HBM.

    xorigin / stride are rounded so row starts are 64-byte aligned; element
    (x, y) is data[(yorigin + y) * stride + xorigin + x]
    (src/tiling/plane_region.rs:185).  rav1e luma planes use 88 px of padding
    (src/frame/mod.rs:22-23)."""

    def __init__(self, width, height, bit_depth=8, xpad=88, ypad=88, device="cuda"):
        self.bpp = 1 if bit_depth == 8 else 2
        al = 64 // self.bpp
        self.width, self.height, self.bit_depth = width, height, bit_depth
        self.xpad, self.ypad = xpad, ypad
        self.xorigin = (xpad + al - 1) // al * al
        self.yorigin = ypad
        self.stride = (self.xorigin + width + xpad + al - 1) // al * al
        self.alloc_height = self.yorigin + height + ypad
        dt = torch.uint8 if self.bpp == 1 else torch.int16  # raw 16-bit storage
        self.data = torch.zeros((self.alloc_height, self.stride), dtype=dt, device=device)

    @classmethod
    def from_numpy(cls, arr, width, height, bit_depth, xpad, ypad, device="cuda"):
        """arr: the full (alloc_height, stride) host array of a HostPlane-style layout."""
        p = cls(width, height, bit_depth, xpad, ypad, device)
        assert arr.shape == (p.alloc_height, p.stride), (arr.shape, p.alloc_height, p.stride)
        src = arr if p.bpp == 1 else arr.view(np.int16)
        p.data.copy_(torch.from_numpy(np.ascontiguousarray(src)))
        return p

    def cstruct(self):
        return _lib.R1Plane(self.data.data_ptr(), self.stride, self.alloc_height, self.width,
                            self.height, self.xorigin, self.yorigin, self.bpp, self.bit_depth)


ME_BLOCK_CAND = np.dtype([("bx", "<i2"), ("by", "<i2"), ("w", "u1"), ("h", "u1"), ("corner", "u1"),
                          ("reserved", "u1"), ("pmv", "<i2", (2, 2))])
ME_RESULT = np.dtype([("row", "<i2"), ("col", "<i2"), ("sad", "<u4"), ("cost", "<u8")])
assert ME_BLOCK_CAND.itemsize == 16 and ME_RESULT.itemsize == 16


CFL_ALPHA_CAND = np.dtype([("x", "<i2"), ("y", "<i2"), ("variant", "u1"), ("vis_w", "u1"), ("vis_h", "u1"),
                           ("reserved", "u1")])
SGR_SOLVE_UNIT = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("set", "u1"), ("edges", "u1"),
                           ("reserved", "u1", (2,))])
SGR_EDGE_LEFT, SGR_EDGE_ABOVE = 1, 2   # R1_SGR_EDGE_*: see include/rav1e_amd.h, R1SgrSolveUnit
assert SGR_SOLVE_UNIT.itemsize == 12
# R1TrialUnit: a superblock whose restoration unit holds a self-guided choice (r1_cdef_lrf_trial_batch)
TRIAL_UNIT = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("set", "u1"), ("edges", "u1"),
                       ("xqd", "i1", (2,)), ("sb", "<i4")])
assert TRIAL_UNIT.itemsize == 16


def me_lambdas(me_lambda):
    """lambda of the three ME passes by ssdec (src/me.rs:175-177): fi.me_lambda = sqrt(fi.lambda)."""
    return [int(me_lambda * 256.0 / (1 << (2 * ss)) * (0.5 if ss == 0 else 0.125)) for ss in range(3)]


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_cands(cands, dtype):
    """numpy structured array (or device uint8 tensor) -> device byte tensor"""
    if isinstance(cands, torch.Tensor):
        return cands
    a = np.ascontiguousarray(cands, dtype=dtype)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()


class Context:
    """r1_ctx handle (one per process/GPU).  Thread-safe in the library."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise R1Error("rav1e_amd needs a HIP device; there is no CPU fallback")
        h = C.c_void_p()
        rc = self.lib.r1_ctx_create(device, C.byref(h))
        if rc != 0:
            raise R1Error("r1_ctx_create: %s" % self.lib.r1_last_error().decode())
        self.h = h

    def close(self):
        if self.h:
            self.lib.r1_ctx_destroy(self.h)
            self.h = None

    def _check(self, rc, what):
        if rc != 0:
            raise R1Error("%s failed (%d): %s" % (what, rc, self.lib.r1_last_error().decode()))

    # ---- dist:: ----
    def dist_batch(self, kind, org, ref, w, h, cands, n=None, out=None):
        """get_sad / get_satd (src/dist.rs:31,156) over a candidate list."""
        dc = _dev_cands(cands, DIST_CAND)
        n = dc.numel() // DIST_CAND.itemsize if n is None else n
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device="cuda")
        po, pr = org.cstruct(), ref.cstruct()
        self._check(self.lib.r1_dist_batch(self.h, int(kind), C.byref(po), C.byref(pr), w, h,
                                           dc.data_ptr(), n, out.data_ptr(), _stream_ptr()),
                    "r1_dist_batch")
        return out

    def dist_scaled_batch(self, kind, org, ref, w, h, cands, scales=None, xdec=0, ydec=0,
                          n=None, out=None):
        """sse_wxh (kind 2) / cdef_dist_wxh (kind 3) of src/rdo.rs:142-224 over a
        candidate list; scales: (rows, stride) int32 device tensor of Q14
        DistortionScale per 8x8 luma importance block, or None."""
        dc = _dev_cands(cands, DIST_CAND)
        n = dc.numel() // DIST_CAND.itemsize if n is None else n
        if out is None:
            out = torch.empty(n, dtype=torch.int64, device="cuda")
        po, pr = org.cstruct(), ref.cstruct()
        self._check(self.lib.r1_dist_scaled_batch(
            self.h, int(kind), C.byref(po), C.byref(pr), w, h, dc.data_ptr(), n,
            scales.data_ptr() if scales is not None else None,
            scales.stride(0) if scales is not None else 0, xdec, ydec, out.data_ptr(),
            _stream_ptr()), "r1_dist_scaled_batch")
        return out

    # ---- transform::forward ----
    def forward_transform_batch(self, residual, tx_size, tx_type, bit_depth, coeff_bytes=None,
                                out=None):
        """forward_transform (src/transform/forward.rs:71) over n dense blocks.
        residual: int16 device tensor (n, h, w)."""
        w, h = TX_DIMS[int(tx_size)]
        n = residual.numel() // (w * h)
        if coeff_bytes is None:
            coeff_bytes = 2 if bit_depth == 8 else 4
        if out is None:
            out = torch.empty((n, w * h), dtype=torch.int16 if coeff_bytes == 2 else torch.int32,
                              device="cuda")
        self._check(self.lib.r1_fwd_txfm_batch(self.h, residual.data_ptr(), out.data_ptr(), n,
                                               int(tx_size), int(tx_type), bit_depth, coeff_bytes,
                                               _stream_ptr()), "r1_fwd_txfm_batch")
        return out

    # ---- transform::inverse ----
    def inverse_transform_add_batch(self, coeffs, pred, tx_size, tx_type, bit_depth, out=None):
        """inverse_transform_add (src/transform/inverse.rs:1633) over n blocks.
        coeffs: (n, stride) int16/int32 device tensor (stride >= coded area);
        pred: (n, h, w) pixels; returns the reconstruction (pass out=pred for
        the reference's in-place behaviour)."""
        w, h = TX_DIMS[int(tx_size)]
        n = pred.numel() // (w * h)
        bpp = 1 if bit_depth == 8 else 2
        if out is None:
            out = torch.empty_like(pred)
        self._check(self.lib.r1_inv_txfm_add_batch(
            self.h, coeffs.data_ptr(), coeffs.stride(0) if coeffs.dim() > 1 else coeffs.numel() // n,
            pred.data_ptr(), out.data_ptr(), n, int(tx_size), int(tx_type), bit_depth, bpp,
            _stream_ptr()), "r1_inv_txfm_add_batch")
        return out

    # ---- quantize:: ----
    @staticmethod
    def _qparams(qindex, bit_depth, is_intra, dc_delta_q=0, ac_delta_q=0):
        return _lib.R1QuantParams(int(qindex), int(bit_depth), int(bool(is_intra)), int(dc_delta_q),
                                  int(ac_delta_q))

    def quantize_batch(self, coeffs, tx_size, tx_type, qindex, bit_depth, is_intra,
                       dc_delta_q=0, ac_delta_q=0, want_rcoeffs=True):
        """QuantizationContext::quantize (+ dequantize) over n blocks
        (src/quantize/mod.rs:268-384).  coeffs: (n, stride) device tensor.
        -> dict(qcoeffs, eobs, rcoeffs) over the coded area."""
        w, h = TX_DIMS[int(tx_size)]
        area = min(w, 32) * min(h, 32)
        n = coeffs.shape[0]
        cb = coeffs.element_size()
        q = torch.empty((n, area), dtype=coeffs.dtype, device="cuda")
        r = torch.empty((n, area), dtype=coeffs.dtype, device="cuda") if want_rcoeffs else None
        eobs = torch.empty(n, dtype=torch.int16, device="cuda")
        qp = self._qparams(qindex, bit_depth, is_intra, dc_delta_q, ac_delta_q)
        self._check(self.lib.r1_quantize_batch(
            self.h, coeffs.data_ptr(), coeffs.stride(0), n, int(tx_size), int(tx_type), C.byref(qp),
            cb, q.data_ptr(), eobs.data_ptr(), r.data_ptr() if r is not None else None,
            _stream_ptr()), "r1_quantize_batch")
        return {"qcoeffs": q, "eobs": eobs, "rcoeffs": r}

    def quantize_rdo_batch(self, coeffs, tx_size, tx_type, qindex, bit_depth, is_intra,
                           dc_delta_q=0, ac_delta_q=0, want_rate=True):
        """quantize + dequantize + transform-domain distortion + estimate_rate
        (src/encoder.rs:1556-1650, src/rdo.rs:127-139) over n blocks; coeffs must
        hold the full w*h forward-transform output per block."""
        w, h = TX_DIMS[int(tx_size)]
        area = min(w, 32) * min(h, 32)
        n = coeffs.shape[0]
        q = torch.empty((n, area), dtype=coeffs.dtype, device="cuda")
        r = torch.empty((n, area), dtype=coeffs.dtype, device="cuda")
        eobs = torch.empty(n, dtype=torch.int16, device="cuda")
        dist = torch.empty(n, dtype=torch.int64, device="cuda")
        rate = torch.empty(n, dtype=torch.int64, device="cuda") if want_rate else None
        qp = self._qparams(qindex, bit_depth, is_intra, dc_delta_q, ac_delta_q)
        self._check(self.lib.r1_quantize_rdo_batch(
            self.h, coeffs.data_ptr(), coeffs.stride(0), n, int(tx_size), int(tx_type), C.byref(qp),
            coeffs.element_size(), q.data_ptr(), eobs.data_ptr(), r.data_ptr(), dist.data_ptr(),
            rate.data_ptr() if rate is not None else None, _stream_ptr()), "r1_quantize_rdo_batch")
        return {"qcoeffs": q, "eobs": eobs, "rcoeffs": r, "tx_dist": dist, "est_rate": rate}

    def dequantize_batch(self, qcoeffs, tx_size, qindex, bit_depth, dc_delta_q=0, ac_delta_q=0):
        """rust::dequantize (src/quantize/mod.rs:363-384) over n coded-area blocks."""
        n = qcoeffs.shape[0]
        r = torch.empty_like(qcoeffs)
        qp = self._qparams(qindex, bit_depth, 0, dc_delta_q, ac_delta_q)
        self._check(self.lib.r1_dequantize_batch(self.h, qcoeffs.data_ptr(), n, int(tx_size),
                                                 C.byref(qp), qcoeffs.element_size(),
                                                 r.data_ptr(), _stream_ptr()),
                    "r1_dequantize_batch")
        return r

    # ---- predict:: ----
    def intra_edges_batch(self, rec, tile, tx_size, cands, n=None):
        """get_intra_edges (src/partition.rs:639-898) for n transform blocks of
        one tile.  tile = (x, y, w, h) in plane pixels.  -> (edges (n, 257), lens (n, 2))"""
        dc = _dev_cands(cands, INTRA_EDGE_CAND)
        n = dc.numel() // INTRA_EDGE_CAND.itemsize if n is None else n
        edges = torch.empty((n, EDGE_LEN), dtype=torch.uint8 if rec.bpp == 1 else torch.int16,
                            device="cuda")
        lens = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
        pr = rec.cstruct()
        self._check(self.lib.r1_intra_edges_batch(self.h, C.byref(pr), tile[0], tile[1], tile[2],
                                                  tile[3], int(tx_size), dc.data_ptr(), n,
                                                  edges.data_ptr(), EDGE_LEN, lens.data_ptr(),
                                                  _stream_ptr()), "r1_intra_edges_batch")
        return edges, lens

    def predict_intra_batch(self, tx_size, cands, edges, lens, bit_depth, ac=None, n=None):
        """dispatch_predict_intra (src/predict.rs:705-784) for n blocks -> (n, h, w) pixels."""
        w, h = TX_DIMS[int(tx_size)]
        dc = _dev_cands(cands, INTRA_CAND)
        n = dc.numel() // INTRA_CAND.itemsize if n is None else n
        bpp = 1 if bit_depth == 8 else 2
        out = torch.empty((n, h, w), dtype=torch.uint8 if bpp == 1 else torch.int16, device="cuda")
        self._check(self.lib.r1_predict_intra_batch(
            self.h, int(tx_size), dc.data_ptr(), n, edges.data_ptr(), edges.stride(0),
            lens.data_ptr(), ac.data_ptr() if ac is not None else None, bit_depth, bpp,
            out.data_ptr(), _stream_ptr()), "r1_predict_intra_batch")
        return out

    def intra_satd_batch(self, src, tx_size, cands, group, pos_xy, edges, lens, ac=None, n=None):
        """the intra mode pre-screen (src/rdo.rs:1434-1506): predict `group` modes per block from
        one edge set, get_satd against the source block at pos_xy -> (n,) int32"""
        dc = _dev_cands(cands, INTRA_CAND)
        n = dc.numel() // INTRA_CAND.itemsize if n is None else n
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        ps = src.cstruct()
        self._check(self.lib.r1_intra_satd_batch(
            self.h, C.byref(ps), int(tx_size), dc.data_ptr(), n, group, pos_xy.data_ptr(),
            edges.data_ptr(), edges.stride(0), lens.data_ptr(),
            ac.data_ptr() if ac is not None else None, out.data_ptr(), _stream_ptr()),
            "r1_intra_satd_batch")
        return out

    def update_block_importances(self, intra_costs, future_importances, inter_costs, mvs, w, h, length,
                                 ref_importances):
        """update_block_importances after its SATD map (src/api/internal.rs:911-1068); device
        tensors: int32 (w*h) costs, float32 importances, int16 (w*h, 2) (row, col) motion vectors;
        ref_importances (float32) is updated in place"""
        need = self.lib.r1_update_block_importances_scratch_bytes(w, h) if w * h else 0
        if need < 0:
            raise R1Error("r1_update_block_importances_scratch_bytes failed")
        scratch = torch.empty(max(int(need), 256), dtype=torch.uint8, device="cuda")
        self._check(self.lib.r1_update_block_importances(
            self.h, intra_costs.data_ptr(), future_importances.data_ptr(), inter_costs.data_ptr(),
            mvs.data_ptr(), w, h, length, ref_importances.data_ptr(), scratch.data_ptr(),
            scratch.numel(), _stream_ptr()), "r1_update_block_importances")
        return ref_importances

    def prescreen_select_batch(self, keys, group, keep_head, k):
        """the selection step of the mode pre-screens (src/rdo.rs:1352-1357, 1504-1509): keys
        (n_groups * group,) int32 SATDs in the caller's candidate order -> (n_groups, k) uint8
        indices: the first keep_head candidates stay, the rest stably sorted by key, take k"""
        n_groups = keys.numel() // group
        out = torch.empty((n_groups, k), dtype=torch.uint8, device="cuda")
        self._check(self.lib.r1_prescreen_select_batch(self.h, keys.data_ptr(), n_groups, group,
                                                       keep_head, k, out.data_ptr(), _stream_ptr()),
                    "r1_prescreen_select_batch")
        return out

    def cfl_alpha_search_batch(self, src, tx_size, cands, edges, lens, ac, n=None):
        """rdo_cfl_alpha (src/rdo.rs:1593-1688) for one chroma plane -> (alpha int16, sse int64)"""
        dc = _dev_cands(cands, CFL_ALPHA_CAND)
        n = dc.numel() // CFL_ALPHA_CAND.itemsize if n is None else n
        alpha = torch.empty(n, dtype=torch.int16, device="cuda")
        cost = torch.empty(n, dtype=torch.int64, device="cuda")
        ps = src.cstruct()
        self._check(self.lib.r1_cfl_alpha_search_batch(
            self.h, C.byref(ps), int(tx_size), dc.data_ptr(), n, edges.data_ptr(), edges.stride(0),
            lens.data_ptr(), ac.data_ptr(), alpha.data_ptr(), cost.data_ptr(), _stream_ptr()),
            "r1_cfl_alpha_search_batch")
        return alpha, cost

    def cfl_ac_batch(self, luma, bw, bh, xdec, ydec, cands, n=None):
        """pred_cfl_ac (src/predict.rs:1020-1063) -> (n, bh*bw) int16"""
        dc = _dev_cands(cands, CFL_AC_CAND)
        n = dc.numel() // CFL_AC_CAND.itemsize if n is None else n
        ac = torch.empty((n, bw * bh), dtype=torch.int16, device="cuda")
        pl = luma.cstruct()
        self._check(self.lib.r1_cfl_ac_batch(self.h, C.byref(pl), bw, bh, xdec, ydec, dc.data_ptr(),
                                             n, ac.data_ptr(), _stream_ptr()), "r1_cfl_ac_batch")
        return ac

    # ---- cdef:: ----
    def cdef_find_dir_batch(self, luma, cands, n=None):
        """cdef_find_dir (src/cdef.rs:84-143) -> (dir uint8, var int32)"""
        dc = _dev_cands(cands, CDEF_DIR_CAND)
        n = dc.numel() // CDEF_DIR_CAND.itemsize if n is None else n
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        v = torch.empty(n, dtype=torch.int32, device="cuda")
        pl = luma.cstruct()
        self._check(self.lib.r1_cdef_find_dir_batch(self.h, C.byref(pl), dc.data_ptr(), n,
                                                    d.data_ptr(), v.data_ptr(), _stream_ptr()),
                    "r1_cdef_find_dir_batch")
        return d, v

    def cdef_filter_block_batch(self, src, dst, xdec, ydec, cands, n=None):
        """cdef_filter_block (src/cdef.rs:198-298) for n blocks, src plane -> dst plane"""
        dc = _dev_cands(cands, CDEF_BLOCK_CAND)
        n = dc.numel() // CDEF_BLOCK_CAND.itemsize if n is None else n
        a, b = src.cstruct(), dst.cstruct()
        self._check(self.lib.r1_cdef_filter_block_batch(self.h, C.byref(a), C.byref(b), xdec, ydec,
                                                        dc.data_ptr(), n, _stream_ptr()),
                    "r1_cdef_filter_block_batch")

    @staticmethod
    def _cdef_params(y_strengths, uv_strengths, damping, bit_depth):
        prm = _lib.R1CdefParams()
        for i in range(8):
            prm.y_strengths[i] = int(y_strengths[i])
            prm.uv_strengths[i] = int(uv_strengths[i])
        prm.damping, prm.bit_depth = int(damping), int(bit_depth)
        return prm

    def cdef_filter_frame_plane(self, luma, src, dst, p, xdec, ydec, tile_w, tile_h, skip_mi,
                                cdef_index_sb, y_strengths, uv_strengths, damping, bit_depth):
        """cdef_filter_tile (src/cdef.rs:600-625) for plane p of the whole frame.
        skip_mi: (mi_rows, mi_cols) uint8 device tensor; cdef_index_sb: (sb_rows, sb_cols)."""
        prm = self._cdef_params(y_strengths, uv_strengths, damping, bit_depth)
        l, a, b = luma.cstruct(), src.cstruct(), dst.cstruct()
        self._check(self.lib.r1_cdef_filter_frame_plane(
            self.h, C.byref(l), C.byref(a), C.byref(b), p, xdec, ydec, tile_w, tile_h,
            skip_mi.data_ptr(), skip_mi.stride(0), skip_mi.shape[1], skip_mi.shape[0],
            cdef_index_sb.data_ptr(), cdef_index_sb.stride(0), C.byref(prm), _stream_ptr()),
            "r1_cdef_filter_frame_plane")

    def cdef_analyze_frame(self, luma, tile_w, tile_h, mi_cols, mi_rows):
        """cdef_analyze_superblock (src/cdef.rs:340-373) for every 8x8 luma block of the frame ->
        (dir uint8 [nby, nbx], var int32 [nby, nbx]); blocks outside mi_cols x mi_rows stay 0"""
        nbx, nby = ((tile_w + 63) // 64) * 8, ((tile_h + 63) // 64) * 8
        assert self.lib.r1_cdef_analyze_blocks(tile_w, tile_h) == nbx * nby
        d = torch.zeros((nby, nbx), dtype=torch.uint8, device="cuda")
        v = torch.zeros((nby, nbx), dtype=torch.int32, device="cuda")
        l = luma.cstruct()
        self._check(self.lib.r1_cdef_analyze_frame(self.h, C.byref(l), tile_w, tile_h, mi_cols, mi_rows,
                                                   d.data_ptr(), v.data_ptr(), _stream_ptr()),
                    "r1_cdef_analyze_frame")
        return d, v

    def cdef_filter_frame_plane_dirs(self, dirs, variances, src, dst, p, xdec, ydec, tile_w, tile_h,
                                     skip_mi, cdef_index_sb, y_strengths, uv_strengths, damping, bit_depth):
        """cdef_filter_superblock over the frame for plane p with the analysis of cdef_analyze_frame"""
        prm = self._cdef_params(y_strengths, uv_strengths, damping, bit_depth)
        a, b = src.cstruct(), dst.cstruct()
        self._check(self.lib.r1_cdef_filter_frame_plane_dirs(
            self.h, dirs.data_ptr(), variances.data_ptr(), C.byref(a), C.byref(b), p, xdec, ydec,
            tile_w, tile_h, skip_mi.data_ptr(), skip_mi.stride(0), skip_mi.shape[1], skip_mi.shape[0],
            cdef_index_sb.data_ptr(), cdef_index_sb.stride(0), C.byref(prm), _stream_ptr()),
            "r1_cdef_filter_frame_plane_dirs")

    def cdef_strength_search(self, rec, src, skip_mi, y_strengths, uv_strengths, damping, bit_depth,
                             n_idx, xdec, ydec, crop_w, crop_h, area_sb=(1, 1), scales=None,
                             dist_scale=(1 << 14, 1 << 14, 1 << 14)):
        """the CDEF leg of rdo_loop_decision (src/rdo.rs:2104-2560, no restoration filter): for
        every superblock and cdef_index < n_idx the ScaledDistortion of the filtered superblock
        against the source, and the first index of smallest cost.
        rec / src: lists of 1 or 3 Planes (whole frame; rec deblocked); skip_mi: (mi_rows, mi_cols)
        uint8 device tensor; scales: (h/8, w/8) int32 device tensor (Q14) or None.
        -> (err (n_sby, n_sbx, 8) int64 holding u64, best (n_sby, n_sbx) int8, -1 = skipped)"""
        prm = _lib.R1CdefSearchParams()
        for i in range(8):
            prm.y_strengths[i] = int(y_strengths[i])
            prm.uv_strengths[i] = int(uv_strengths[i])
        prm.damping, prm.bit_depth, prm.n_idx, prm.planes = int(damping), int(bit_depth), int(n_idx), len(rec)
        prm.xdec, prm.ydec, prm.crop_w, prm.crop_h = int(xdec), int(ydec), int(crop_w), int(crop_h)
        prm.area_sb_w, prm.area_sb_h = int(area_sb[0]), int(area_sb[1])
        for i in range(3):
            prm.dist_scale[i] = int(dist_scale[i])
        mi_rows, mi_cols = skip_mi.shape
        n_sbx, n_sby = (mi_cols + 15) // 16, (mi_rows + 15) // 16
        pr = (_lib.R1Plane * 3)(*[(rec[k] if k < len(rec) else rec[0]).cstruct() for k in range(3)])
        ps = (_lib.R1Plane * 3)(*[(src[k] if k < len(src) else src[0]).cstruct() for k in range(3)])
        err = torch.empty((n_sby, n_sbx, 8), dtype=torch.int64, device="cuda")
        best = torch.empty((n_sby, n_sbx), dtype=torch.int8, device="cuda")
        scratch = torch.empty(self.lib.r1_cdef_strength_search_scratch_bytes(mi_cols, mi_rows),
                              dtype=torch.uint8, device="cuda")
        self._check(self.lib.r1_cdef_strength_search(
            self.h, pr, ps, skip_mi.data_ptr(), skip_mi.stride(0), mi_cols, mi_rows,
            scales.data_ptr() if scales is not None else None,
            scales.stride(0) if scales is not None else 0, C.byref(prm), err.data_ptr(), best.data_ptr(),
            scratch.data_ptr(), _stream_ptr()), "r1_cdef_strength_search")
        return err, best

    def _cdef_search_params(self, n_planes, y_strengths, uv_strengths, damping, bit_depth, n_idx, xdec, ydec, crop_w,
                            crop_h, area_sb, dist_scale):
        prm = _lib.R1CdefSearchParams()
        for i in range(8):
            prm.y_strengths[i] = int(y_strengths[i])
            prm.uv_strengths[i] = int(uv_strengths[i])
        prm.damping, prm.bit_depth, prm.n_idx, prm.planes = int(damping), int(bit_depth), int(n_idx), n_planes
        prm.xdec, prm.ydec, prm.crop_w, prm.crop_h = int(xdec), int(ydec), int(crop_w), int(crop_h)
        prm.area_sb_w, prm.area_sb_h = int(area_sb[0]), int(area_sb[1])
        for i in range(3):
            prm.dist_scale[i] = int(dist_scale[i])
        return prm

    def cdef_lrf_trial_scratch(self, rec, skip_mi, n_idx, xdec, ydec):
        """the scratch of cdef_lrf_trial_batch (it holds every index' trial output as whole planes), to allocate once"""
        mi_rows, mi_cols = skip_mi.shape
        nb = self.lib.r1_cdef_lrf_trial_scratch_bytes(mi_cols, mi_rows, xdec if len(rec) == 3 else 0,
                                                      ydec if len(rec) == 3 else 0, rec[0].bpp, n_idx, len(rec))
        return torch.empty(nb, dtype=torch.uint8, device="cuda")

    def cdef_lrf_trial_batch(self, rec, cdef_cur, src, skip_mi, units, y_strengths, uv_strengths, damping, bit_depth,
                             n_idx, xdec, ydec, crop_w, crop_h, area_sb=(1, 1), scales=None,
                             dist_scale=(1 << 14, 1 << 14, 1 << 14), sb_sel=None, scratch=None, outs=None):
        """a later pass of rdo_loop_decision's CDEF leg (src/rdo.rs:2377-2560): every (superblock, cdef_index) trial
        with the restoration units' CURRENT choices applied to the trial's output before the error is taken.
        units: three TRIAL_UNIT arrays (Y, U, V; may be empty) -- the superblocks under a self-guided choice;
        cdef_cur: the working copy (cdef_apply_area) or None when no unit carries an edge flag.
        -> (err (n_sby, n_sbx, 8) int64, err_planes (n_sby, n_sbx, 8, 3) int64, best (n_sby, n_sbx) int8)"""
        prm = self._cdef_search_params(len(rec), y_strengths, uv_strengths, damping, bit_depth, n_idx, xdec, ydec, crop_w,
                                       crop_h, area_sb, dist_scale)
        mi_rows, mi_cols = skip_mi.shape
        n_sbx, n_sby = (mi_cols + 15) // 16, (mi_rows + 15) // 16
        cur = cdef_cur if cdef_cur is not None else rec
        pr = (_lib.R1Plane * 3)(*[(rec[k] if k < len(rec) else rec[0]).cstruct() for k in range(3)])
        pc = (_lib.R1Plane * 3)(*[(cur[k] if k < len(cur) else cur[0]).cstruct() for k in range(3)])
        ps = (_lib.R1Plane * 3)(*[(src[k] if k < len(src) else src[0]).cstruct() for k in range(3)])
        if isinstance(units, tuple):          # (device byte tensor or None, (n_y, n_u, n_v)): uploaded once by the caller
            du, counts = units
            n_units = (C.c_int32 * 3)(*[int(v) for v in counts])
        else:
            us = [np.ascontiguousarray(u, TRIAL_UNIT) for u in units] + [np.zeros(0, TRIAL_UNIT)] * (3 - len(units))
            n_units = (C.c_int32 * 3)(*[len(u) for u in us])
            allu = np.concatenate(us)
            du = _dev_cands(allu, TRIAL_UNIT) if len(allu) else None
        o = outs if outs is not None else {}
        if "err" not in o:
            o["err"] = torch.empty((n_sby, n_sbx, 8), dtype=torch.int64, device="cuda")
        if "err_planes" not in o:
            o["err_planes"] = torch.empty((n_sby, n_sbx, 8, 3), dtype=torch.int64, device="cuda")
        if "best" not in o:
            o["best"] = torch.empty((n_sby, n_sbx), dtype=torch.int8, device="cuda")
        err, errp, best = o["err"], o["err_planes"], o["best"]
        nb = self.lib.r1_cdef_lrf_trial_scratch_bytes(mi_cols, mi_rows, xdec if len(rec) == 3 else 0,
                                                      ydec if len(rec) == 3 else 0, rec[0].bpp, n_idx, len(rec))
        if scratch is None or scratch.numel() < nb:
            scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
        self._check(self.lib.r1_cdef_lrf_trial_batch(
            self.h, pr, pc, ps, skip_mi.data_ptr(), skip_mi.stride(0), mi_cols, mi_rows,
            scales.data_ptr() if scales is not None else None, scales.stride(0) if scales is not None else 0,
            C.byref(prm), du.data_ptr() if du is not None else None, n_units,
            sb_sel.data_ptr() if sb_sel is not None else None, err.data_ptr(), errp.data_ptr(), best.data_ptr(),
            scratch.data_ptr(), _stream_ptr()), "r1_cdef_lrf_trial_batch")
        return err, errp, best

    def cdef_apply_area(self, rec, out, skip_mi, index_sb, y_strengths, uv_strengths, damping, bit_depth, n_idx, xdec,
                        ydec, crop_w, crop_h, area_sb=(1, 1)):
        """the CDEF working copy of every analysis area (src/rdo.rs:2546-2560): cdef_filter_superblock with
        index_sb[sby, sbx] (int8 device tensor; < 0 = unfiltered) from rec into out (lists of 1 or 3 Planes)"""
        prm = self._cdef_search_params(len(rec), y_strengths, uv_strengths, damping, bit_depth, n_idx, xdec, ydec, crop_w,
                                       crop_h, area_sb, (1 << 14,) * 3)
        mi_rows, mi_cols = skip_mi.shape
        pr = (_lib.R1Plane * 3)(*[(rec[k] if k < len(rec) else rec[0]).cstruct() for k in range(3)])
        po = (_lib.R1Plane * 3)(*[(out[k] if k < len(out) else out[0]).cstruct() for k in range(3)])
        scratch = torch.empty(self.lib.r1_cdef_strength_search_scratch_bytes(mi_cols, mi_rows),
                              dtype=torch.uint8, device="cuda")
        assert index_sb.dtype == torch.int8 and index_sb.is_contiguous()
        self._check(self.lib.r1_cdef_apply_area(
            self.h, pr, po, skip_mi.data_ptr(), skip_mi.stride(0), mi_cols, mi_rows, C.byref(prm),
            index_sb.data_ptr(), scratch.data_ptr(), _stream_ptr()), "r1_cdef_apply_area")
        return out

    # ---- frame glue ----
    def plane_pad(self, plane, w, h, xdec=0, ydec=0):
        """Plane::pad(w, h) in place (FramePad::pad, src/frame/mod.rs:76-86); w, h: frame size"""
        pl = plane.cstruct()
        self._check(self.lib.r1_plane_pad(self.h, C.byref(pl), w, h, xdec, ydec, _stream_ptr()),
                    "r1_plane_pad")
        return plane

    def plane_downsample(self, src, frame_w, frame_h, dec):
        """Plane::downsampled(frame_w, frame_h) (src/encoder.rs:476-477) -> new padded Plane of
        half the size and half the padding; dec: the NEW plane's decimation (1 half, 2 quarter)"""
        dst = Plane((src.width + 1) // 2, (src.height + 1) // 2, src.bit_depth, src.xpad // 2, src.ypad // 2)
        a, b = src.cstruct(), dst.cstruct()
        self._check(self.lib.r1_plane_downsample(self.h, C.byref(a), C.byref(b), frame_w, frame_h, dec, dec,
                                                 _stream_ptr()), "r1_plane_downsample")
        return dst

    # ---- lookahead cost maps ----
    def estimate_intra_costs(self, luma):
        """estimate_intra_costs (src/api/lookahead.rs:30-123) -> (h/8, w/8) int32"""
        hb, wb = luma.height // 8, luma.width // 8
        out = torch.empty((hb, wb), dtype=torch.int32, device="cuda")
        pl = luma.cstruct()
        self._check(self.lib.r1_estimate_intra_costs(self.h, C.byref(pl), out.data_ptr(),
                                                     _stream_ptr()), "r1_estimate_intra_costs")
        return out

    def estimate_inter_costs(self, org, ref, mvs):
        """SATD map of estimate_inter_costs (lookahead.rs:226-268); mvs: (h/8, w/8, 2)
        int16 device tensor of (row, col) in 1/8 pel"""
        hb, wb = org.height // 8, org.width // 8
        out = torch.empty((hb, wb), dtype=torch.int32, device="cuda")
        a, b = org.cstruct(), ref.cstruct()
        self._check(self.lib.r1_estimate_inter_costs(self.h, C.byref(a), C.byref(b), mvs.data_ptr(),
                                                     out.data_ptr(), _stream_ptr()),
                    "r1_estimate_inter_costs")
        return out

    def activity_scales(self, luma):
        """ActivityMask::from_plane + fill_scales (src/activity.rs:21-66) -> (variances, scales),
        uint32-valued int32 tensors (ceil(h/8), ceil(w/8))"""
        hb, wb = (luma.height + 7) // 8, (luma.width + 7) // 8
        var = torch.empty((hb, wb), dtype=torch.int32, device="cuda")
        sc = torch.empty((hb, wb), dtype=torch.int32, device="cuda")
        pl = luma.cstruct()
        self._check(self.lib.r1_activity_scales(self.h, C.byref(pl), var.data_ptr(), sc.data_ptr(),
                                                _stream_ptr()), "r1_activity_scales")
        return var, sc

    def importance_block_difference(self, org, ref):
        """estimate_importance_block_difference (lookahead.rs:125-180) -> f64"""
        out = torch.zeros(1, dtype=torch.int64, device="cuda")
        a, b = org.cstruct(), ref.cstruct()
        self._check(self.lib.r1_importance_block_difference(self.h, C.byref(a), C.byref(b),
                                                            out.data_ptr(), _stream_ptr()),
                    "r1_importance_block_difference")
        n = (org.height // 8) * (org.width // 8)
        return float(out.item()) / n

    # ---- mc:: ----
    def put_8tap_batch(self, ref, w, h, cands, n=None, out=None):
        dc = _dev_cands(cands, MC_CAND)
        n = dc.numel() // MC_CAND.itemsize if n is None else n
        if out is None:
            out = torch.empty((n, h, w), dtype=torch.uint8 if ref.bpp == 1 else torch.int16,
                              device="cuda")
        pr = ref.cstruct()
        self._check(self.lib.r1_mc_put_batch(self.h, C.byref(pr), w, h, dc.data_ptr(), n,
                                             out.data_ptr(), _stream_ptr()), "r1_mc_put_batch")
        return out

    def prep_8tap_batch(self, ref, w, h, cands, n=None, out=None):
        dc = _dev_cands(cands, MC_CAND)
        n = dc.numel() // MC_CAND.itemsize if n is None else n
        if out is None:
            out = torch.empty((n, h, w), dtype=torch.int16, device="cuda")
        pr = ref.cstruct()
        self._check(self.lib.r1_mc_prep_batch(self.h, C.byref(pr), w, h, dc.data_ptr(), n,
                                              out.data_ptr(), _stream_ptr()), "r1_mc_prep_batch")
        return out

    def mc_batch_mfma(self, ref, w, h, cands, prep=False, n=None, out=None):
        """put_8tap / prep_8tap with the horizontal pass on the matrix cores (csrc/mc_mfma.hip)"""
        dc = _dev_cands(cands, MC_CAND)
        n = dc.numel() // MC_CAND.itemsize if n is None else n
        if out is None:
            out = torch.empty((n, h, w), dtype=torch.int16 if prep else torch.uint8, device="cuda")
        pr = ref.cstruct()
        self._check(self.lib.r1_mc_batch_mfma(self.h, int(prep), C.byref(pr), w, h, dc.data_ptr(), n,
                                              out.data_ptr(), _stream_ptr()), "r1_mc_batch_mfma")
        return out

    def mc_avg_batch(self, tmp1, tmp2, w, h, bit_depth, out=None):
        n = tmp1.numel() // (w * h)
        bpp = 1 if bit_depth == 8 else 2
        if out is None:
            out = torch.empty((n, h, w), dtype=torch.uint8 if bpp == 1 else torch.int16,
                              device="cuda")
        self._check(self.lib.r1_mc_avg_batch(self.h, tmp1.data_ptr(), tmp2.data_ptr(), w, h, n,
                                             bit_depth, bpp, out.data_ptr(), _stream_ptr()),
                    "r1_mc_avg_batch")
        return out


    def prepare_rdo_cand(self, org, ref, w, h, dcands, n, outs):
        """Bind one fused-candidate launch once (descriptor and output tensors
        stay alive in the returned closure): the per-step host cost is then one
        ctypes call -- what a native caller of the C ABI pays -- instead of
        rebuilding the argument structs in Python every step."""
        from .types import TxSize
        tx_size = int(TxSize.by_dims(w, h))
        po, pr = org.cstruct(), ref.cstruct()
        f = self.lib.r1_rdo_cand_batch
        args = [self.h, C.byref(po), C.byref(pr), w, h, tx_size, dcands.data_ptr(), n,
                outs["sad"].data_ptr(), outs["satd"].data_ptr(), outs["coeffs"].data_ptr(), None]
        keep = (po, pr, dcands, outs)

        def launch():
            rc = f(*args, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                self._check(rc, "r1_rdo_cand_batch")
            return keep
        return launch

    def prepare_rdo_full_cand(self, org, ref, w, h, dcands, n, qindex, outs, is_intra=0):
        """prepare_rdo_cand for r1_rdo_full_cand_batch (outs: sad, satd, eob, tx_dist, est_rate)."""
        from .types import TxSize
        tx_size = int(TxSize.by_dims(w, h))
        po, pr = org.cstruct(), ref.cstruct()
        qp = self._qparams(qindex, org.bit_depth, is_intra, 0, 0)
        f = self.lib.r1_rdo_full_cand_batch
        args = [self.h, C.byref(po), C.byref(pr), w, h, tx_size, dcands.data_ptr(), n, C.byref(qp),
                outs["sad"].data_ptr(), outs["satd"].data_ptr(), outs["eob"].data_ptr(),
                outs["tx_dist"].data_ptr(), outs["est_rate"].data_ptr(), None, None]
        keep = (po, pr, qp, dcands, outs)

        def launch():
            rc = f(*args, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                self._check(rc, "r1_rdo_full_cand_batch")
            return keep
        return launch

    def rdo_full_cand_batch(self, org, ref, w, h, cands, qindex, is_intra=0, dc_delta_q=0,
                            ac_delta_q=0, n=None, want_sad=True, want_satd=True, want_rate=True,
                            want_qcoeffs=False, want_coeffs=False, outs=None):
        """mc -> sad/satd -> diff -> forward_transform -> quantize -> dequantize ->
        tx-domain distortion -> estimate_rate for every candidate, one launch."""
        from .types import TxSize
        tx_size = int(TxSize.by_dims(w, h))
        dc = _dev_cands(cands, RDO_CAND)
        n = dc.numel() // RDO_CAND.itemsize if n is None else n
        ct = torch.int16 if org.bpp == 1 else torch.int32
        o = outs if outs is not None else {}
        if "eob" not in o:
            o["eob"] = torch.empty(n, dtype=torch.int16, device="cuda")
        if "tx_dist" not in o:
            o["tx_dist"] = torch.empty(n, dtype=torch.int64, device="cuda")
        if want_sad:
            if "sad" not in o:
                o["sad"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_satd:
            if "satd" not in o:
                o["satd"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_rate:
            if "est_rate" not in o:
                o["est_rate"] = torch.empty(n, dtype=torch.int64, device="cuda")
        if want_qcoeffs:
            if "qcoeffs" not in o:
                o["qcoeffs"] = torch.empty((n, min(w, 32) * min(h, 32)), dtype=ct, device="cuda")
        if want_coeffs:
            if "coeffs" not in o:
                o["coeffs"] = torch.empty((n, w * h), dtype=ct, device="cuda")
        po, pr = org.cstruct(), ref.cstruct()
        qp = self._qparams(qindex, org.bit_depth, is_intra, dc_delta_q, ac_delta_q)

        def p(k):
            return o[k].data_ptr() if k in o else None
        self._check(self.lib.r1_rdo_full_cand_batch(
            self.h, C.byref(po), C.byref(pr), w, h, tx_size, dc.data_ptr(), n, C.byref(qp),
            p("sad"), p("satd"), p("eob"), p("tx_dist"), p("est_rate"), p("qcoeffs"), p("coeffs"),
            _stream_ptr()), "r1_rdo_full_cand_batch")
        return o

    # ---- me:: ----
    def estimate_tile_motion(self, jobs, w_in_b, h_in_b, bit_depth, lambdas, allow_hp=True,
                             allow_full_search=False, me_range_scale=1, launch_mode=0):
        """estimate_tile_motion (src/me.rs:153-218) for a list of independent
        (tile, reference frame) jobs.  Each job: dict(org=[Plane x3], ref=[Plane x3]
        (full, half, quarter resolution), stats=int32 tensor (rows, cols, 2) viewing the
        FrameMEStats array [(row | col << 16), normalized_sad], prev=tensor or None,
        tile=(x, y, w, h) luma px).  lambdas: per ssdec (see me_lambdas)."""
        n = len(jobs)
        arr = (_lib.R1MeJob * n)()
        rows, cols = jobs[0]["stats"].shape[:2]
        for j, job in enumerate(jobs):
            st = job["stats"]
            assert st.dtype == torch.int32 and st.shape == (rows, cols, 2) and st.is_contiguous()
            for l in range(3):
                arr[j].org[l] = job["org"][l].cstruct()
                arr[j].ref[l] = job["ref"][l].cstruct()
            arr[j].stats = st.data_ptr()
            pv = job.get("prev")
            arr[j].prev = pv.data_ptr() if pv is not None else None
            arr[j].tile_x, arr[j].tile_y, arr[j].tile_w, arr[j].tile_h = job["tile"]
        p = _lib.R1MeParams(w_in_b, h_in_b, cols, rows, bit_depth, int(allow_hp),
                            int(allow_full_search), me_range_scale,
                            (C.c_uint32 * 3)(*[int(v) for v in lambdas]), int(launch_mode))
        self._check(self.lib.r1_estimate_tile_motion_batch(self.h, arr, n, C.byref(p), _stream_ptr()),
                    "r1_estimate_tile_motion_batch")

    def estimate_frame_motion(self, jobs, w_in_b, h_in_b, bit_depth, lambdas, **kw):
        """The frame-level caller of estimate_tile_motion: enqueue, wait, and acknowledge.  If the
        persistent launch flagged a timed-out dependency wait (r1_me_status -> R1_ETIMEDOUT) the
        statistics are recomputed with launch_mode 1 (launch boundaries, no waits) -- the
        automatic fallback include/rav1e_amd.h describes.  `prev` of the jobs is untouched by a
        search, and a search overwrites every entry of its tile, so the re-issue is idempotent.
        -> the launch mode that produced the statistics."""
        mode = int(kw.pop("launch_mode", 0))
        self.estimate_tile_motion(jobs, w_in_b, h_in_b, bit_depth, lambdas, launch_mode=mode, **kw)
        ok, _, _ = self.me_status(wait=True)
        if ok:
            return mode
        if mode == 1:
            raise R1Error("r1_me_status flagged a launch_mode 1 call: " + self.lib.r1_last_error().decode())
        self.estimate_tile_motion(jobs, w_in_b, h_in_b, bit_depth, lambdas, launch_mode=1, **kw)
        ok, _, _ = self.me_status(wait=True)
        assert ok
        return 1

    def me_status(self, wait=True):
        """r1_me_status: (ok, first_failed_call, calls) -- the statistics of the persistent tile-ME
        launches are valid once this has reported ok after them (include/rav1e_amd.h)."""
        first, calls = C.c_ulonglong(0), C.c_ulonglong(0)
        rc = self.lib.r1_me_status(self.h, int(wait), C.byref(first), C.byref(calls))
        if rc not in (0, -5):
            self._check(rc, "r1_me_status")
        return rc == 0, int(first.value), int(calls.value)

    def estimate_motion_batch(self, job, cands, w_in_b, h_in_b, bit_depth, lambdas, max_w=64,
                              max_h=64, use_satd=True, filter_mode=0, allow_hp=True, n=None):
        """estimate_motion(.., Some(pmv), ..) of src/rdo.rs:1183-1196 for independent
        blocks (ME_BLOCK_CAND array) of one (tile, reference) job -> ME_RESULT tensor bytes."""
        arr = (_lib.R1MeJob * 1)()
        st = job["stats"]
        rows, cols = st.shape[:2]
        for l in range(3):
            arr[0].org[l] = job["org"][l].cstruct()
            arr[0].ref[l] = job["ref"][l].cstruct()
        arr[0].stats = st.data_ptr()
        pv = job.get("prev")
        arr[0].prev = pv.data_ptr() if pv is not None else None
        arr[0].tile_x, arr[0].tile_y, arr[0].tile_w, arr[0].tile_h = job["tile"]
        p = _lib.R1MeParams(w_in_b, h_in_b, cols, rows, bit_depth, int(allow_hp), 0, 1,
                            (C.c_uint32 * 3)(*[int(v) for v in lambdas]), 0)
        dc = _dev_cands(cands, ME_BLOCK_CAND)
        n = dc.numel() // ME_BLOCK_CAND.itemsize if n is None else n
        out = torch.empty(n * ME_RESULT.itemsize, dtype=torch.uint8, device="cuda")
        self._check(self.lib.r1_estimate_motion_batch(self.h, arr, C.byref(p), dc.data_ptr(), n, max_w,
                                                      max_h, int(use_satd), filter_mode,
                                                      out.data_ptr(), _stream_ptr()),
                    "r1_estimate_motion_batch")
        return out

    # ---- deblock:: ----
    def deblock_plane(self, state, plane, pli, xdec, ydec, blocks, crop_w, crop_h):
        """deblock_plane (src/deblock.rs:1294-1459), in place.  state: 24-byte R1DeblockState
        (numpy, host); blocks: uint8 device tensor (mi_rows, mi_cols, 8) of R1DeblockBlock."""
        pc = plane.cstruct()
        st = np.ascontiguousarray(state).view(np.uint8)
        assert st.size == 24 and blocks.dtype == torch.uint8 and blocks.shape[2] == 8
        self._check(self.lib.r1_deblock_plane(self.h, st.ctypes.data, C.byref(pc), pli, xdec, ydec,
                                              blocks.data_ptr(), blocks.shape[1], blocks.shape[1],
                                              blocks.shape[0], crop_w, crop_h, _stream_ptr()),
                    "r1_deblock_plane")

    def deblock_sse_plane(self, rec, src, pli, xdec, ydec, blocks, crop_w, crop_h, tallies=None):
        """sse_plane (src/deblock.rs:1461-1542) -> int64 device tensor (2, 65): vertical, horizontal"""
        if tallies is None:
            tallies = torch.zeros((2, 65), dtype=torch.int64, device="cuda")
        pr, ps = rec.cstruct(), src.cstruct()
        self._check(self.lib.r1_deblock_sse_plane(self.h, C.byref(pr), C.byref(ps), pli, xdec, ydec,
                                                  blocks.data_ptr(), blocks.shape[1], blocks.shape[1],
                                                  blocks.shape[0], crop_w, crop_h,
                                                  tallies[0].data_ptr(), tallies[1].data_ptr(),
                                                  _stream_ptr()), "r1_deblock_sse_plane")
        return tallies

    def deblock_frame(self, state, planes, xdec, ydec, blocks, crop_w, crop_h):
        """deblock_filter_frame (src/deblock.rs:1544-1551): the three planes in place, two launches"""
        arr = (_lib.R1Plane * 3)(*[p.cstruct() for p in planes])
        st = np.ascontiguousarray(state).view(np.uint8)
        assert st.size == 24 and blocks.dtype == torch.uint8 and blocks.shape[2] == 8
        self._check(self.lib.r1_deblock_frame(self.h, st.ctypes.data, arr, xdec, ydec, blocks.data_ptr(),
                                              blocks.shape[1], blocks.shape[1], blocks.shape[0], crop_w,
                                              crop_h, _stream_ptr()), "r1_deblock_frame")

    def deblock_sse_frame(self, rec, src, xdec, ydec, blocks, crop_w, crop_h, tallies=None):
        """the level search of all planes and directions in one launch -> int64 (3, 2, 65)"""
        if tallies is None:
            tallies = torch.zeros((3, 2, 65), dtype=torch.int64, device="cuda")
        ra = (_lib.R1Plane * 3)(*[p.cstruct() for p in rec])
        sa = (_lib.R1Plane * 3)(*[p.cstruct() for p in src])
        self._check(self.lib.r1_deblock_sse_frame(self.h, ra, sa, xdec, ydec, blocks.data_ptr(),
                                                  blocks.shape[1], blocks.shape[1], blocks.shape[0],
                                                  crop_w, crop_h, tallies.data_ptr(), _stream_ptr()),
                    "r1_deblock_sse_frame")
        return tallies

    def deblock_pick_levels(self, tallies, pli):
        """sse_optimize's tail on host copies of the tallies -> levels (2 for luma, 1 for chroma)"""
        t = np.ascontiguousarray(tallies.cpu().numpy())
        out = np.zeros(2, np.uint8)
        self._check(self.lib.r1_deblock_pick_levels(t[0].ctypes.data, t[1].ctypes.data, pli,
                                                    out.ctypes.data), "r1_deblock_pick_levels")
        return out[:2] if pli == 0 else out[:1]

    def rdo_pixel_cand_batch(self, org, ref, w, h, cands, qindex, dist_kind, scales=None, xdec=0,
                             ydec=0, is_intra=0, dc_delta_q=0, ac_delta_q=0, n=None, want_sad=True,
                             want_satd=True, want_qcoeffs=False, want_rec=False, outs=None, pred=None):
        """mc -> sad/satd -> diff -> forward_transform -> quantize -> dequantize -> inverse
        transform -> reconstruction -> weighted SSE / cdef_dist against the source, one launch.
        pred (dense (n, h, w) device tensor): r1_rdo_pred_cand_batch -- the prediction comes
        from that buffer instead of put_8tap(ref); dist_kind 0 = transform-domain distortion."""
        from .types import TxSize
        tx_size = int(TxSize.by_dims(w, h))
        dc = _dev_cands(cands, RDO_CAND)
        n = dc.numel() // RDO_CAND.itemsize if n is None else n
        ct = torch.int16 if org.bpp == 1 else torch.int32
        o = outs if outs is not None else {}
        if "eob" not in o:
            o["eob"] = torch.empty(n, dtype=torch.int16, device="cuda")
        if "dist" not in o:
            o["dist"] = torch.empty(n, dtype=torch.int64, device="cuda")
        if want_sad:
            if "sad" not in o:
                o["sad"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_satd:
            if "satd" not in o:
                o["satd"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_qcoeffs:
            if "qcoeffs" not in o:
                o["qcoeffs"] = torch.empty((n, min(w, 32) * min(h, 32)), dtype=ct, device="cuda")
        if want_rec:
            if "rec" not in o:
                o["rec"] = torch.empty((n, h, w), dtype=torch.uint8 if org.bpp == 1 else torch.int16,
                                            device="cuda")
        po = org.cstruct()
        qp = self._qparams(qindex, org.bit_depth, is_intra, dc_delta_q, ac_delta_q)

        def p(k):
            return o[k].data_ptr() if k in o else None
        if pred is not None:
            self._check(self.lib.r1_rdo_pred_cand_batch(
                self.h, C.byref(po), pred.data_ptr(), w, h, tx_size, dc.data_ptr(), n, C.byref(qp),
                dist_kind, scales.data_ptr() if scales is not None else None,
                scales.stride(0) if scales is not None else 0, xdec, ydec, p("sad"), p("satd"),
                p("eob"), p("dist"), p("qcoeffs"), p("rec"), _stream_ptr()), "r1_rdo_pred_cand_batch")
            return o
        pr = ref.cstruct()
        self._check(self.lib.r1_rdo_pixel_cand_batch(
            self.h, C.byref(po), C.byref(pr), w, h, tx_size, dc.data_ptr(), n, C.byref(qp), dist_kind,
            scales.data_ptr() if scales is not None else None,
            scales.stride(0) if scales is not None else 0, xdec, ydec, p("sad"), p("satd"), p("eob"),
            p("dist"), p("qcoeffs"), p("rec"), _stream_ptr()), "r1_rdo_pixel_cand_batch")
        return o

    def tx_type_mask(self, tx_size, is_inter, use_reduced_set=False, rav1e_types_only=True):
        """av1_tx_used[get_tx_set(..)] (src/context/transform_unit.rs:37-44, 123-148) as a TxType bit
        mask, by default cut down to RAV1E_TX_TYPES (src/transform/mod.rs:28-44)"""
        return int(self.lib.r1_tx_type_mask(int(tx_size), int(bool(is_inter)), int(bool(use_reduced_set)),
                                            int(bool(rav1e_types_only))))

    def rdo_txsearch_batch(self, org, ref, w, h, cands, tx_type_mask, qindex, dist_kind, scales=None, xdec=0,
                           ydec=0, is_intra=0, dc_delta_q=0, ac_delta_q=0, n=None, want_sad=False,
                           want_satd=False, want_est_rate=False, want_qcoeffs=False, want_rec=False,
                           outs=None, pred=None):
        """r1_rdo_txsearch_batch: rdo_tx_type_decision's per-type loop (src/rdo.rs:1701-1817) on one
        prediction per candidate -- put_8tap(ref), or the dense `pred` (n, h, w) tensor when ref is None.
        eob / dist (/ est_rate): (n, nt); qcoeffs: (n, nt, coded area); rec: (n, nt, h, w); slot j = the
        j-th set bit of tx_type_mask."""
        from .types import TxSize
        tx_size = int(TxSize.by_dims(w, h))
        dc = _dev_cands(cands, RDO_CAND)
        n = dc.numel() // RDO_CAND.itemsize if n is None else n
        nt = bin(int(tx_type_mask)).count("1")
        ct = torch.int16 if org.bpp == 1 else torch.int32
        o = outs if outs is not None else {}
        if "eob" not in o:
            o["eob"] = torch.empty((n, nt), dtype=torch.int16, device="cuda")
        if "dist" not in o:
            o["dist"] = torch.empty((n, nt), dtype=torch.int64, device="cuda")
        if want_sad:
            if "sad" not in o:
                o["sad"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_satd:
            if "satd" not in o:
                o["satd"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_est_rate:
            if "est_rate" not in o:
                o["est_rate"] = torch.empty((n, nt), dtype=torch.int64, device="cuda")
        if want_qcoeffs:
            if "qcoeffs" not in o:
                o["qcoeffs"] = torch.empty((n, nt, min(w, 32) * min(h, 32)), dtype=ct, device="cuda")
        if want_rec:
            if "rec" not in o:
                o["rec"] = torch.empty((n, nt, h, w), dtype=torch.uint8 if org.bpp == 1 else torch.int16,
                                            device="cuda")
        po = org.cstruct()
        pr = ref.cstruct() if ref is not None else None
        qp = self._qparams(qindex, org.bit_depth, is_intra, dc_delta_q, ac_delta_q)

        def p(k):
            return o[k].data_ptr() if k in o else None
        self._check(self.lib.r1_rdo_txsearch_batch(
            self.h, C.byref(po), C.byref(pr) if pr is not None else None,
            pred.data_ptr() if pred is not None else None, w, h, tx_size, dc.data_ptr(), n, int(tx_type_mask),
            C.byref(qp), dist_kind, scales.data_ptr() if scales is not None else None,
            scales.stride(0) if scales is not None else 0, xdec, ydec, p("sad"), p("satd"), p("eob"), p("dist"),
            p("est_rate"), p("qcoeffs"), p("rec"), _stream_ptr()), "r1_rdo_txsearch_batch")
        return o

    # ---- lrf:: ----
    def lrf_sgrproj_plane(self, cdeffed, deblocked, out, ydec, crop_w, crop_h, frame_height, unit_size,
                          units, stripe_height):
        """lrf_filter_frame's Sgrproj arm for one plane (src/lrf.rs:1482-1585); units: uint8 device
        tensor (unit_rows, unit_cols, 4) of R1LrfUnit; `out` must already hold the CDEF output."""
        pc, pd, po = cdeffed.cstruct(), deblocked.cstruct(), out.cstruct()
        assert units.dtype == torch.uint8 and units.shape[2] == 4 and units.is_contiguous()
        self._check(self.lib.r1_lrf_sgrproj_plane(self.h, C.byref(pc), C.byref(pd), C.byref(po), ydec,
                                                  crop_w, crop_h, frame_height, unit_size,
                                                  units.shape[1], units.shape[0], stripe_height,
                                                  units.data_ptr(), _stream_ptr()),
                    "r1_lrf_sgrproj_plane")

    def sgrproj_solve_batch(self, cdeffed, inp, units, max_w=256, max_h=256):
        """sgrproj_solve (src/lrf.rs:847-1096) for (unit, set) pairs; units: SGR_SOLVE_UNIT array (`edges`: SGR_EDGE_* --
        the unit's place in its rdo_loop_decision area, rdo_glue.restoration_unit_edges) -> (n, 2) int8 xqd"""
        dc = _dev_cands(units, SGR_SOLVE_UNIT)
        n = dc.numel() // SGR_SOLVE_UNIT.itemsize
        scratch = torch.empty(n * 5, dtype=torch.int64, device="cuda")
        out = torch.empty((n, 2), dtype=torch.int8, device="cuda")
        pc, pi = cdeffed.cstruct(), inp.cstruct()
        self._check(self.lib.r1_sgrproj_solve_batch(self.h, C.byref(pc), C.byref(pi), dc.data_ptr(), n,
                                                    max_w, max_h, scratch.data_ptr(), out.data_ptr(),
                                                    _stream_ptr()), "r1_sgrproj_solve_batch")
        return out

    def lrf_search_batch(self, lrf_in, src, units, is_chroma=False, xdec=0, ydec=0, scales=None, dist_scale=1 << 14,
                         max_w=256, max_h=256):
        """the restoration leg of rdo_loop_decision for one plane (src/rdo.rs:2575-2763) but the rate:
        units: SGR_SOLVE_UNIT array (set 255 = the no-filter option; `edges` from rdo_glue.restoration_unit_edges, 0 for
        one unit per plane and area; list the sets of a unit next to each other); scales: 2-D int32/uint32 device
        tensor (one DistortionScale per 8x8 luma block) or None -> ((n, 2) int8 xqd, (n,) int64 err)"""
        dc = _dev_cands(units, SGR_SOLVE_UNIT)
        n = dc.numel() // SGR_SOLVE_UNIT.itemsize
        scratch = torch.empty(n * 6, dtype=torch.int64, device="cuda")
        xqd = torch.empty((n, 2), dtype=torch.int8, device="cuda")
        err = torch.empty(n, dtype=torch.int64, device="cuda")
        pc, ps = lrf_in.cstruct(), src.cstruct()
        self._check(self.lib.r1_lrf_search_batch(self.h, C.byref(pc), C.byref(ps), dc.data_ptr(), n, max_w, max_h,
                                                 int(bool(is_chroma)), xdec, ydec,
                                                 scales.data_ptr() if scales is not None else None,
                                                 scales.shape[1] if scales is not None else 0, int(dist_scale),
                                                 scratch.data_ptr(), xqd.data_ptr(), err.data_ptr(), _stream_ptr()),
                    "r1_lrf_search_batch")
        return xqd, err

    # ---- fused candidate ----
    def rdo_cand_batch(self, org, ref, w, h, cands, n=None, want_sad=True, want_satd=True,
                       want_coeffs=True, want_pred=False, outs=None):
        """mc -> sad/satd -> diff -> forward_transform for every candidate."""
        from .types import TxSize
        tx_size = int(TxSize.by_dims(w, h))
        dc = _dev_cands(cands, RDO_CAND)
        n = dc.numel() // RDO_CAND.itemsize if n is None else n
        o = outs or {}
        if want_sad and "sad" not in o:
            o["sad"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_satd and "satd" not in o:
            o["satd"] = torch.empty(n, dtype=torch.int32, device="cuda")
        if want_coeffs and "coeffs" not in o:
            o["coeffs"] = torch.empty((n, w * h), dtype=torch.int16 if org.bpp == 1 else torch.int32,
                                      device="cuda")
        if want_pred and "pred" not in o:
            o["pred"] = torch.empty((n, h, w), dtype=torch.uint8 if org.bpp == 1 else torch.int16,
                                    device="cuda")
        po, pr = org.cstruct(), ref.cstruct()

        def p(k, want):
            return o[k].data_ptr() if want else None
        self._check(self.lib.r1_rdo_cand_batch(self.h, C.byref(po), C.byref(pr), w, h, tx_size,
                                               dc.data_ptr(), n, p("sad", want_sad),
                                               p("satd", want_satd), p("coeffs", want_coeffs),
                                               p("pred", want_pred), _stream_ptr()),
                    "r1_rdo_cand_batch")
        return o
