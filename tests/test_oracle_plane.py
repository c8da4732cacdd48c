"""oracle/plane.c (v_frame 0.3.9 Plane::pad / Plane::downsampled, restated -- the crate is not
under /root/reference) against (1) the 4x4-in-8x9 example of the crate's own unit tests, as
recalled from its published source, and (2) the independent NumPy statement the ME tests have
used since round 1 (oracle_lib.plane_from_image / box_down2)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


def _tiny_plane():
    p = O.HostPlane(4, 4, 8, 0, 0)
    p.stride, p.alloc_height, p.xorigin, p.yorigin = 8, 9, 2, 3
    p.data = np.zeros((9, 8), np.uint8)
    p.data[3:7, 2:6] = [[1, 2, 3, 4], [8, 7, 6, 5], [9, 8, 7, 6], [2, 3, 4, 5]]
    return p


def test_pad_crate_example():
    L = O.lib()
    p = _tiny_plane()
    pc = p.cstruct()
    L.r1o_plane_pad(C.byref(pc), 4, 4, 0, 0)
    want = np.array([[1, 1, 1, 2, 3, 4, 4, 4]] * 4 + [[8, 8, 8, 7, 6, 5, 5, 5], [9, 9, 9, 8, 7, 6, 6, 6]]
                    + [[2, 2, 2, 3, 4, 5, 5, 5]] * 3, np.uint8)
    assert np.array_equal(p.data, want)


def test_downsample_crate_example():
    L = O.lib()
    p = _tiny_plane()
    d = O.HostPlane(2, 2, 8, 0, 0)
    pc, dc = p.cstruct(), d.cstruct()
    assert L.r1o_plane_downsample(C.byref(pc), C.byref(dc), 4, 4, 1, 1) == 0
    assert d.view().tolist() == [[5, 5], [6, 6]]
    # everything right of / below the visible area replicates it
    assert (d.data[0] == [5, 5] + [5] * (d.stride - 2)).all() and (d.data[1, 1:] == 6).all()


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("w,h", [(64, 48), (71, 37), (200, 9)])
def test_pad_and_downsample_match_numpy_statement(w, h, bd):
    L = O.lib()
    rng = np.random.default_rng(w + h + bd)
    img = rng.integers(0, 1 << bd, (h, w))
    full = O.HostPlane(w, h, bd, 88, 88, rng=np.random.default_rng(1))   # junk in the padding
    full.view()[...] = img
    fc = full.cstruct()
    L.r1o_plane_pad(C.byref(fc), w, h, 0, 0)
    assert np.array_equal(full.data, O.plane_from_image(img, bd, 88, 88).data)
    pyr = O.me_pyramid(img, bd)
    half = O.HostPlane((w + 1) // 2, (h + 1) // 2, bd, 44, 44, rng=np.random.default_rng(2))
    hc = half.cstruct()
    assert L.r1o_plane_downsample(C.byref(fc), C.byref(hc), w, h, 1, 1) == 0
    assert np.array_equal(half.data, pyr[1].data)
    quarter = O.HostPlane((half.width + 1) // 2, (half.height + 1) // 2, bd, 22, 22, rng=np.random.default_rng(3))
    qc = quarter.cstruct()
    assert L.r1o_plane_downsample(C.byref(hc), C.byref(qc), w, h, 2, 2) == 0
    # pad() derives the padded-from size as (frame + dec) >> dec, which for dec = 2 is not the
    # ceiling: the quarter plane can lose its last row / column to the replication
    pw, ph = (w + 2) >> 2, (h + 2) >> 2
    want = O.plane_from_image(pyr[2].view()[:ph, :pw], bd, 22, 22)
    assert want.data.shape[1] == quarter.stride
    assert np.array_equal(quarter.data[:22 + ph], want.data[:22 + ph])
    assert (quarter.data[22 + ph:] == quarter.data[22 + ph - 1]).all()


def test_pad_from_a_frame_size_smaller_than_the_plane():
    """rav1e's planes are 8-aligned; pad() is given the frame size (src/api/internal.rs:1436), so
    the columns / rows between the frame edge and the plane edge are overwritten too"""
    L = O.lib()
    p = O.HostPlane(16, 8, 8, 8, 8, rng=np.random.default_rng(5))
    before = p.view().copy()
    pc = p.cstruct()
    L.r1o_plane_pad(C.byref(pc), 13, 6, 0, 0)
    v = p.view()
    assert np.array_equal(v[:6, :13], before[:6, :13])
    assert (v[:6, 13:] == before[:6, 12:13]).all() and (v[6:, :13] == before[5, :13]).all()
    assert (p.data[:, -1] == p.data[:, p.xorigin + 12]).all()
