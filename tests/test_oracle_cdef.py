"""Pin the CDEF oracle: the reference's first_max_element known answers
(src/cdef.rs:304-309, via the direction search's tie-break) and whole-frame
vectors of an independent AV1-spec-formulation model
(tests/golden/gen_cdef_golden.py)."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O

HERE = os.path.dirname(__file__)


def host_plane(arr, bd, pad=16):
    h, w = arr.shape
    hp = O.HostPlane(w, h, bd, pad, pad)
    hp.view()[:] = arr
    return hp


def cdef_frame_oracle(oracle, G, k):
    W, H, xdec, ydec, bd, damping = (int(v) for v in G[k + "_meta"])
    ins = [host_plane(G[k + "_in%d" % p].astype(np.uint16 if bd > 8 else np.uint8), bd) for p in range(3)]
    outs = [host_plane(np.zeros_like(G[k + "_in%d" % p]).astype(np.uint16 if bd > 8 else np.uint8), bd)
            for p in range(3)]
    skip = np.ascontiguousarray(G[k + "_skip"])
    ci = np.ascontiguousarray(G[k + "_ci"])
    ystr, uvstr = np.ascontiguousarray(G[k + "_ystr"]), np.ascontiguousarray(G[k + "_uvstr"])
    luma = ins[0].cstruct()
    res = []
    for p in range(3):
        xd, yd = (0, 0) if p == 0 else (xdec, ydec)
        a, b = ins[p].cstruct(), outs[p].cstruct()
        oracle.r1o_cdef_filter_tile_plane(C.byref(luma), C.byref(a), C.byref(b), p, xd, yd, W, H,
                                          O.ptr(skip), skip.shape[1], skip.shape[1], skip.shape[0],
                                          O.ptr(ci), ci.shape[1], O.ptr(ystr), O.ptr(uvstr), damping, bd)
        res.append(outs[p].view().copy())
    return res


def test_first_max_tie_break(oracle):
    """cdef.rs:304-309: ties go to the first element.  A flat block has all
    costs equal -> direction 0, variance 0."""
    flat = np.full((8, 8), 77, np.uint8)
    var = C.c_uint32(5)
    assert oracle.r1o_cdef_find_dir(O.ptr(flat), 8, C.byref(var), 0, 0) == 0 and var.value == 0
    # horizontal stripes -> direction 2 (horizontal), vertical stripes -> 6
    hs = np.repeat(np.array([[10], [200]] * 4, np.uint8), 8, axis=1)
    assert oracle.r1o_cdef_find_dir(O.ptr(np.ascontiguousarray(hs)), 8, C.byref(var), 0, 0) == 2
    assert oracle.r1o_cdef_find_dir(O.ptr(np.ascontiguousarray(hs.T)), 8, C.byref(var), 0, 0) == 6


import pytest


@pytest.mark.parametrize("fixture,ncases", [("cdef_ref", 12), ("cdef_golden", 10)])
def test_spec_model_frames(oracle, fixture, ncases):
    """cdef_ref.npz: whole frames filtered by the reference's own source text
    (gen_cdef_ref.py executes src/cdef.rs incl. cdef_filter_tile); cdef_golden.npz: an
    independent AV1-spec-formulation model."""
    Z = np.load(os.path.join(HERE, "golden", fixture + ".npz"))
    G = {k: Z[k] for k in Z.files}
    ncase = len([k for k in G if k.endswith("_meta")])
    assert ncase == ncases
    for c in range(ncase):
        k = "c%d" % c
        W, H, xdec, ydec, bd, damping = (int(v) for v in G[k + "_meta"])
        # directions / variances of the non-skipped blocks
        Y = G[k + "_in0"].astype(np.uint16 if bd > 8 else np.uint8)
        for by in range(H // 8):
            for bx in range(W // 8):
                if G[k + "_skip"][2 * by:2 * by + 2, 2 * bx:2 * bx + 2].all():
                    continue
                var = C.c_uint32()
                blk = np.ascontiguousarray(Y[8 * by:8 * by + 8, 8 * bx:8 * bx + 8])
                d = oracle.r1o_cdef_find_dir(O.ptr(blk), 8, C.byref(var), bd - 8, int(bd > 8))
                assert (d, var.value) == (G[k + "_dir"][by, bx], G[k + "_var"][by, bx]), (c, bx, by)
        res = cdef_frame_oracle(oracle, G, k)
        for p in range(3):
            assert np.array_equal(res[p].astype(np.uint16), G[k + "_out%d" % p]), (c, p)


def test_zero_strength_is_identity(oracle):
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (16, 16)).astype(np.uint8)
    dst = np.zeros((8, 8), np.uint8)
    oracle.r1o_cdef_filter_block(O.ptr(dst), 8, O.ptr(src[4:, 4:].copy()), 12, 0, 0, 3, 5, 8, 0, 0, 0, 0)
    # strength 0 both -> output == input block
    blk = np.ascontiguousarray(src[4:12, 4:12])
    oracle.r1o_cdef_filter_block(O.ptr(dst), 8, O.ptr(blk), 8, 0, 0, 3, 5, 8, 0, 0, 0, 0)
    assert np.array_equal(dst, blk)
