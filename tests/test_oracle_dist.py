"""Pin the oracle's dist kernels against the reference's own known-answer
tables (src/dist.rs:383-533) and check the unpinned ones by construction."""
import numpy as np
import pytest

import oracle_lib as O

# (w, h, value) tables copied as DATA from the reference's tests
# get_sad_same_inner (src/dist.rs:418-441) / get_satd_same_inner (477-500)
SAD = [(4, 4, 1912), (4, 8, 4296), (8, 4, 3496), (8, 8, 7824), (8, 16, 16592), (16, 8, 14416),
       (16, 16, 31136), (16, 32, 60064), (32, 16, 59552), (32, 32, 120128), (32, 64, 186688),
       (64, 32, 250176), (64, 64, 438912), (64, 128, 654272), (128, 64, 1016768),
       (128, 128, 1689792), (4, 16, 8680), (16, 4, 6664), (8, 32, 31056), (32, 8, 27600),
       (16, 64, 93344), (64, 16, 116384)]
SATD = [(4, 4, 1408), (4, 8, 2016), (8, 4, 1816), (8, 8, 3984), (8, 16, 5136), (16, 8, 4864),
        (16, 16, 9984), (16, 32, 13824), (32, 16, 13760), (32, 32, 27952), (32, 64, 37168),
        (64, 32, 45104), (64, 64, 84176), (64, 128, 127920), (128, 64, 173680),
        (128, 128, 321456), (4, 16, 3136), (16, 4, 2632), (8, 32, 7056), (32, 8, 6624),
        (16, 64, 18432), (64, 16, 21312)]


def reference_test_planes(bit_depth):
    """setup_planes() of src/dist.rs:384-413: 640x480, pads 136 / 264 (different
    strides), closed-form pattern robust to alignment."""
    a = O.HostPlane(640, 480, bit_depth, 128 + 8, 128 + 8)
    b = O.HostPlane(640, 480, bit_depth, 2 * 128 + 8, 2 * 128 + 8)
    for p, sign in ((a, 1), (b, -1)):
        xoff = p.xorigin - p.xpad - 8
        i = np.arange(p.alloc_height)[:, None]
        j = np.arange(p.stride)[None, :]
        p.data[:] = (j + sign * i - xoff) & 255
    return a, b


@pytest.mark.parametrize("bit_depth", [8, 10])
def test_sad_satd_known_answers(oracle, bit_depth):
    a, b = reference_test_planes(bit_depth)
    hbd = int(bit_depth > 8)
    for w, h, v in SAD:
        assert oracle.r1o_get_sad(a.block_ptr(32, 40), a.stride, b.block_ptr(32, 40), b.stride,
                                  w, h, hbd) == v, (w, h)
    for w, h, v in SATD:
        assert oracle.r1o_get_satd(a.block_ptr(32, 40), a.stride, b.block_ptr(32, 40), b.stride,
                                   w, h, hbd) == v, (w, h)


def test_satd_edge_chunks_fall_back_to_sad(oracle):
    """dist.rs:186-191: chunks that do not fit the Hadamard size use SAD."""
    rng = np.random.default_rng(1)
    a = O.HostPlane(64, 64, 8, 8, 8, rng=rng)
    b = O.HostPlane(64, 64, 8, 8, 8, rng=rng)
    # 12x8: one 8x8 Hadamard + a 4x8 SAD strip
    full = oracle.r1o_get_satd(a.block_ptr(0, 0), a.stride, b.block_ptr(0, 0), b.stride, 12, 8, 0)
    h8 = oracle.r1o_get_satd(a.block_ptr(0, 0), a.stride, b.block_ptr(0, 0), b.stride, 8, 8, 0)
    s = oracle.r1o_get_sad(a.block_ptr(8, 0), a.stride, b.block_ptr(8, 0), b.stride, 4, 8, 0)
    # un-normalised sums add, then one rounding
    va = a.view().astype(np.int64)[:8, :8] - b.view().astype(np.int64)[:8, :8]
    H = np.array([[1]])
    for _ in range(3):
        H = np.block([[H, H], [H, -H]])
    raw = np.abs(H @ va @ H.T).sum()
    assert h8 == (raw + 4) >> 3
    assert full == (raw + s + 4) >> 3


def test_weighted_sse_matches_definition(oracle):
    rng = np.random.default_rng(2)
    for bd in (8, 10, 12):
        a = O.HostPlane(64, 64, bd, 8, 8, rng=rng)
        b = O.HostPlane(64, 64, bd, 8, 8, rng=rng)
        for w, h in ((4, 4), (8, 8), (16, 32), (64, 64)):
            stride = 1 << max(0, (w // 4 - 1).bit_length())
            scale = rng.integers(1 << 13, 3 << 13, size=(h // 4, stride)).astype(np.uint32)
            got = oracle.r1o_get_weighted_sse(a.block_ptr(0, 0), a.stride, b.block_ptr(0, 0),
                                              b.stride, O.ptr(scale), stride, w, h, int(bd > 8))
            d = a.view()[:h, :w].astype(np.int64) - b.view()[:h, :w].astype(np.int64)
            cells = (d * d).reshape(h // 4, 4, w // 4, 4).sum(axis=(1, 3))
            tot = int((((cells * scale[:, :w // 4].astype(np.int64)) + 128) >> 8).sum())
            assert got == (tot + 32) // 64
        # unit scale (1<<14) == plain SSE (src/asm/shared/dist/sse.rs scale=1 case)
        scale = np.full((16, 16), 1 << 14, dtype=np.uint32)
        got = oracle.r1o_get_weighted_sse(a.block_ptr(0, 0), a.stride, b.block_ptr(0, 0), b.stride,
                                          O.ptr(scale), 16, 64, 64, int(bd > 8))
        d = a.view().astype(np.int64) - b.view().astype(np.int64)
        cells = (d * d).reshape(16, 4, 16, 4).sum(axis=(1, 3))
        assert got == (int((((cells << 14) + 128) >> 8).sum()) + 32) // 64


def test_ssim_boost_against_float(oracle):
    """activity.rs:194-274 accuracy test: within 5 % of the float formula."""
    rng = np.random.default_rng(3)
    for bd in (8, 10, 12):
        sh = 2 * (bd - 8)
        for _ in range(200):
            svar = int(rng.integers(0, 1 << 14)) << sh
            dvar = int(rng.integers(0, 1 << 14)) << sh
            got = oracle.r1o_apply_ssim_boost(1 << 14, svar, dvar, bd) / float(1 << 14)
            s, d = svar >> sh, dvar >> sh
            ref = (3355 / 12338) * (s + d + 16128) / np.sqrt(3355.0 ** 2 + s * d)
            assert abs(got - ref) / ref < 0.05


def test_cdef_dist_identical_blocks_is_zero(oracle):
    rng = np.random.default_rng(4)
    a = O.HostPlane(16, 16, 8, 8, 8, rng=rng)
    for w in range(1, 9):
        for h in range(1, 9):
            assert oracle.r1o_cdef_dist_kernel(a.block_ptr(0, 0), a.stride, a.block_ptr(0, 0),
                                               a.stride, w, h, 8, 0) == 0
