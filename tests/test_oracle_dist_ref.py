"""Pin oracle/dist.c (and the DistortionScale / ssim-boost arithmetic) to
tests/golden/dist_ref.npz: vectors produced by EXECUTING THE REFERENCE'S OWN
SOURCE TEXT (src/dist.rs, src/activity.rs, src/rdo.rs transpiled by
tools/rustlite; generator: tests/golden/gen_dist_ref.py)."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dist_ref.npz"))


def _pair(prefix, k):
    return np.ascontiguousarray(G[prefix + "org_" + k]), np.ascontiguousarray(G[prefix + "ref_" + k])


def test_sad_satd(oracle):
    for i, k in enumerate(G["d_keys"]):
        bd, w, h, _ = map(int, k.split("_"))
        a, b = _pair("d_", k)
        assert oracle.r1o_get_sad(O.ptr(a), w, O.ptr(b), w, w, h, int(bd > 8)) == G["d_sad"][i], k
        assert oracle.r1o_get_satd(O.ptr(a), w, O.ptr(b), w, w, h, int(bd > 8)) == G["d_satd"][i], k


def test_weighted_sse(oracle):
    for i, k in enumerate(G["w_keys"]):
        bd, w, h, _ = map(int, k.split("_"))
        a, b = _pair("w_", k)
        sc = np.ascontiguousarray(G["w_scale_" + k])
        got = oracle.r1o_get_weighted_sse(O.ptr(a), w, O.ptr(b), w, O.ptr(sc), sc.shape[1], w, h,
                                          int(bd > 8))
        assert got == G["w_out"][i], k


def test_cdef_dist_kernel(oracle):
    for i, k in enumerate(G["k_keys"]):
        bd, w, h, _ = map(int, k.split("_"))
        a, b = _pair("k_", k)
        got = oracle.r1o_cdef_dist_kernel(O.ptr(a), w, O.ptr(b), w, w, h, bd, int(bd > 8))
        assert got == G["k_out"][i], k


def test_apply_ssim_boost(oracle):
    for row, want in zip(G["b_in"], G["b_out"]):
        inp, svar, dvar, bd = (int(v) for v in row)
        assert oracle.r1o_apply_ssim_boost(inp, svar, dvar, bd) == want, row


def test_distortion_scale_arithmetic():
    """DistortionScale::new / mul_u64 as the oracle states them (oracle/dist.c uses
    (s*d + 8192) >> 14; `new` is only needed for den = 64 there)."""
    for (num, den), want in zip(G["s_new_in"], G["s_new_out"]):
        num, den = int(num), int(den)
        raw = min(((num << 14) + den // 2), (1 << 64) - 1) // den
        assert min(raw, (1 << 28) - 1) == want
    for (s, d), want in zip(G["s_mul_in"], G["s_mul_out"]):
        assert (int(s) * int(d) + 8192) >> 14 == want


def frame_cases():
    for k in G["f_keys"]:
        bd, kind, w, h, xdec, use_grid = map(int, k.split("_"))
        yield k, bd, kind, w, h, xdec, use_grid


def frame_planes(bd, xdec):
    org, ref = G["f_org_%d" % bd], G["f_ref_%d" % bd]
    H, W = org.shape
    ph, pw = H >> xdec, W >> xdec
    a = O.HostPlane(pw, ph, bd, 16, 16)
    b = O.HostPlane(pw, ph, bd, 24, 24)
    a.view()[:] = org[:ph, :pw]
    b.view()[:] = ref[:ph, :pw]
    return a, b


def test_wxh_glue_on_planes(oracle):
    """cdef_dist_wxh / sse_wxh with the per-8x8 DistortionScale grid looked up as
    distortion_scale() does (src/rdo.rs:142-224, 443-459), luma and 4:2:0 chroma."""
    for k, bd, kind, w, h, xdec, use_grid in frame_cases():
        a, b = frame_planes(bd, xdec)
        scales = np.ascontiguousarray(G["f_scales_%d" % bd])
        cands = G["f_cands_" + k]
        c = np.zeros(len(cands), O.DIST_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"] = cands.T
        want = np.zeros(len(c), np.uint64)
        pa, pb = a.cstruct(), b.cstruct()
        assert oracle.r1o_dist_scaled_batch(kind, C.byref(pa), C.byref(pb), w, h, O.ptr(c), len(c),
                                            O.ptr(scales) if use_grid else None, scales.shape[1],
                                            xdec, xdec, O.ptr(want)) == 0
        assert np.array_equal(want, G["f_out_" + k]), k


def test_activity_scales_reference_vectors(oracle):
    """ActivityMask::from_plane + fill_scales as the reference's text computes them
    (tests/golden/activity_ref.npz, gen_activity_ref.py executes src/activity.rs:21-186)."""
    A = np.load(os.path.join(os.path.dirname(__file__), "golden", "activity_ref.npz"))
    for k in A["keys"]:
        bd, w, h = map(int, k.split("_"))
        hp = O.plane_from_image(A["img_" + k], bd, 16, 16)       # edge-replicated padding (Frame::pad)
        hb, wb = (h + 7) // 8, (w + 7) // 8
        var, sc = np.zeros((hb, wb), np.uint32), np.zeros((hb, wb), np.uint32)
        pc = hp.cstruct()
        oracle.r1o_activity_scales(C.byref(pc), O.ptr(var), O.ptr(sc))
        assert np.array_equal(var, A["var_" + k]), k
        assert np.array_equal(sc, A["scale_" + k]), k
