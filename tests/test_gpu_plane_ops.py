"""r1_plane_pad / r1_plane_downsample (csrc/plane_ops.hip) against oracle/plane.c, bit for bit,
over the whole allocation (borders included)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _dev_plane(hp):
    from rav1e_amd.api import Plane
    return Plane.from_numpy(hp.data, hp.width, hp.height, hp.bit_depth, hp.xpad, hp.ypad)


def _host(t, bd):
    a = t.cpu().numpy()
    return a if bd == 8 else a.view(np.uint16)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("w,h,fw,fh", [(64, 48, 64, 48), (72, 40, 71, 37), (3840, 2160, 3840, 2160),
                                       (1920, 1080, 1920, 1080), (16, 8, 13, 6)])
def test_pad_matches_oracle(ctx, w, h, fw, fh, bd):
    L = O.lib()
    hp = O.HostPlane(w, h, bd, 88, 88, rng=np.random.default_rng(w + bd))
    dp = _dev_plane(hp)
    pc = hp.cstruct()
    L.r1o_plane_pad(C.byref(pc), fw, fh, 0, 0)
    ctx.plane_pad(dp, fw, fh)
    assert np.array_equal(_host(dp.data, bd), hp.data)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("w,h,fw,fh", [(64, 48, 64, 48), (72, 40, 71, 37), (3840, 2160, 3840, 2160),
                                       (1920, 1080, 1920, 1080), (200, 16, 200, 9)])
def test_downsample_pyramid_matches_oracle(ctx, w, h, fw, fh, bd):
    L = O.lib()
    full = O.HostPlane(w, h, bd, 88, 88, rng=np.random.default_rng(h + bd))
    fc = full.cstruct()
    L.r1o_plane_pad(C.byref(fc), fw, fh, 0, 0)
    dfull = _dev_plane(full)
    half = O.HostPlane((w + 1) // 2, (h + 1) // 2, bd, 44, 44, rng=np.random.default_rng(1))
    hc = half.cstruct()
    assert L.r1o_plane_downsample(C.byref(fc), C.byref(hc), fw, fh, 1, 1) == 0
    dhalf = ctx.plane_downsample(dfull, fw, fh, 1)
    assert (dhalf.stride, dhalf.alloc_height, dhalf.xorigin, dhalf.yorigin) == (
        half.stride, half.alloc_height, half.xorigin, half.yorigin)
    assert np.array_equal(_host(dhalf.data, bd), half.data)
    quarter = O.HostPlane((half.width + 1) // 2, (half.height + 1) // 2, bd, 22, 22,
                          rng=np.random.default_rng(2))
    qc = quarter.cstruct()
    assert L.r1o_plane_downsample(C.byref(hc), C.byref(qc), fw, fh, 2, 2) == 0
    dq = ctx.plane_downsample(dhalf, fw, fh, 2)
    assert np.array_equal(_host(dq.data, bd), quarter.data)


def test_pad_on_an_unaligned_allocation(ctx):
    """the scalar path: a plane whose rows are not 16-byte multiples (v_frame's own test layout)"""
    import torch
    from rav1e_amd import _lib
    L = O.lib()
    hp = O.HostPlane(4, 4, 8, 0, 0)
    hp.stride, hp.alloc_height, hp.xorigin, hp.yorigin = 8, 9, 2, 3
    hp.data = np.zeros((9, 8), np.uint8)
    hp.data[3:7, 2:6] = [[1, 2, 3, 4], [8, 7, 6, 5], [9, 8, 7, 6], [2, 3, 4, 5]]
    buf = torch.zeros(9 * 8 + 1, dtype=torch.uint8, device="cuda")
    buf[1:].copy_(torch.from_numpy(hp.data.reshape(-1)))
    pl = _lib.R1Plane(buf.data_ptr() + 1, 8, 9, 4, 4, 2, 3, 1, 8)
    assert ctx.lib.r1_plane_pad(ctx.h, C.byref(pl), 4, 4, 0, 0, None) == 0
    torch.cuda.synchronize()
    pc = hp.cstruct()
    L.r1o_plane_pad(C.byref(pc), 4, 4, 0, 0)
    assert np.array_equal(buf[1:].cpu().numpy().reshape(9, 8), hp.data)


def test_rejects_bad_geometry(ctx):
    from rav1e_amd.api import Plane
    a, b = Plane(64, 64, 8), Plane(30, 32, 8, 44, 44)
    pa, pb = a.cstruct(), b.cstruct()
    assert ctx.lib.r1_plane_downsample(ctx.h, C.byref(pa), C.byref(pb), 64, 64, 1, 1, None) != 0
    assert ctx.lib.r1_plane_pad(ctx.h, C.byref(pa), 4000, 64, 0, 0, None) != 0
