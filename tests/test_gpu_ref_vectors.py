"""GPU parity against REFERENCE-DERIVED vectors: tests/golden/*_ref.npz hold outputs
computed by executing the reference's own Rust source text (tools/rustlite,
tests/golden/gen_*_ref.py).  The HIP kernels are called through the C ABI on the
same inputs and must reproduce them bit for bit -- no oracle in between."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dev_plane(hp):
    from rav1e_amd.api import Plane
    return Plane.from_numpy(hp.data, hp.width, hp.height, hp.bit_depth, hp.xpad, hp.ypad)


def block_plane(arr, bd, pad):
    h, w = arr.shape
    hp = O.HostPlane(w, h, bd, pad, pad)
    hp.view()[:] = arr
    return hp


def test_dist_ref_sad_satd(ctx):
    G = np.load(os.path.join(GOLD, "dist_ref.npz"))
    c = np.zeros(1, O.DIST_CAND)
    for i, k in enumerate(G["d_keys"]):
        bd, w, h, _ = map(int, k.split("_"))
        a, b = block_plane(G["d_org_" + k], bd, 8), block_plane(G["d_ref_" + k], bd, 16)
        da, db = dev_plane(a), dev_plane(b)
        sad = ctx.dist_batch(0, da, db, w, h, c).cpu().numpy().view(np.uint32)[0]
        satd = ctx.dist_batch(1, da, db, w, h, c).cpu().numpy().view(np.uint32)[0]
        assert (sad, satd) == (G["d_sad"][i], G["d_satd"][i]), k


def test_dist_ref_cdef_dist_kernel(ctx):
    """cdef_dist_kernel for every w, h in 1..8: cdef_dist_wxh of one kernel with the default
    DistortionScale (1 << 14) is the kernel value itself."""
    G = np.load(os.path.join(GOLD, "dist_ref.npz"))
    c = np.zeros(1, O.DIST_CAND)
    for i, k in enumerate(G["k_keys"]):
        bd, w, h, _ = map(int, k.split("_"))
        a, b = block_plane(G["k_org_" + k], bd, 8), block_plane(G["k_ref_" + k], bd, 8)
        got = ctx.dist_scaled_batch(3, dev_plane(a), dev_plane(b), w, h, c).cpu().numpy().view(np.uint64)[0]
        assert got == G["k_out"][i], k


def test_dist_ref_wxh_glue_on_planes(ctx):
    """cdef_dist_wxh / sse_wxh + distortion_scale() lookups, luma and 4:2:0 chroma."""
    import torch
    G = np.load(os.path.join(GOLD, "dist_ref.npz"))
    cache = {}
    for k in G["f_keys"]:
        bd, kind, w, h, xdec, use_grid = map(int, k.split("_"))
        if (bd, xdec) not in cache:
            org, ref = G["f_org_%d" % bd], G["f_ref_%d" % bd]
            H, W = org.shape
            a = block_plane(org[:H >> xdec, :W >> xdec], bd, 16)
            b = block_plane(ref[:H >> xdec, :W >> xdec], bd, 24)
            sc = torch.from_numpy(np.ascontiguousarray(G["f_scales_%d" % bd]).view(np.int32)).cuda()
            cache[(bd, xdec)] = (dev_plane(a), dev_plane(b), sc)
        da, db, sc = cache[(bd, xdec)]
        cands = G["f_cands_" + k]
        c = np.zeros(len(cands), O.DIST_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"] = cands.T
        got = ctx.dist_scaled_batch(kind, da, db, w, h, c, sc if use_grid else None, xdec, xdec)
        assert np.array_equal(got.cpu().numpy().view(np.uint64), G["f_out_" + k]), k
