"""Pin the forward-transform oracles (C and NumPy restatements) against the
vectors derived from the reference's own source text (tests/golden/)."""
import os

import numpy as np
import pytest

import fwd_tx_np as F
import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fwd_tx_golden.npz"))


def test_1d_networks_match_reference_source_vectors(oracle):
    for t in range(13):
        x = G["in1d_%d" % t]
        n = x.shape[1]
        y = np.stack(F.TXFM_FUNCS[t]([np.ascontiguousarray(x[:, i]) for i in range(n)]), axis=1)
        assert np.array_equal(y, G["out1d_%d" % t]), "numpy 1-D type %d" % t
        yc = x.copy()
        for row in yc:
            oracle.r1o_fwd_txfm_1d(O.ptr(row), t)
        assert np.array_equal(yc, G["out1d_%d" % t]), "C 1-D type %d" % t


def test_2d_all_sizes_types_bitdepths(oracle):
    keys = list(G["keys2d"])
    assert len(keys) == 480
    for k in keys:
        bd, ts, tt = map(int, k.split("_"))
        res = np.ascontiguousarray(G["res2d_" + k])
        h, w = res.shape
        want = G["coef2d_" + k]
        out32 = np.zeros(w * h, np.int32)
        assert oracle.r1o_forward_transform(O.ptr(res), O.ptr(out32), w, ts, tt, bd, 1) == 0
        assert np.array_equal(out32, want), k
        assert np.array_equal(F.forward_transform(res, ts, tt, bd), want), k
        if bd == 8:  # T::Coeff = i16 for u8 pixels: truncating cast (forward.rs:157)
            out16 = np.zeros(w * h, np.int16)
            oracle.r1o_forward_transform(O.ptr(res), O.ptr(out16), w, ts, tt, bd, 0)
            assert np.array_equal(out16, want.astype(np.int16)), k


def test_invalid_pairs_rejected(oracle):
    # valid_av1_transform (transform/mod.rs:405-417)
    assert oracle.r1o_valid_av1_transform(4, 0) == 1      # 64x64 DCT
    assert oracle.r1o_valid_av1_transform(4, 1) == 0      # 64x64 ADST
    assert oracle.r1o_valid_av1_transform(3, 9) == 1      # 32x32 IDTX
    assert oracle.r1o_valid_av1_transform(3, 3) == 0
    assert oracle.r1o_valid_av1_transform(17, 9) == 0     # 16x64 IDTX
    assert oracle.r1o_valid_av1_transform(0, 16) == 1     # WHT 4x4
    assert oracle.r1o_valid_av1_transform(1, 16) == 0
    buf = np.zeros(64 * 64, np.int16)
    out = np.zeros(64 * 64, np.int32)
    assert oracle.r1o_forward_transform(O.ptr(buf), O.ptr(out), 64, 4, 1, 8, 1) == -1


def test_survey_check_vectors_4x4_dct(oracle):
    """TX_4X4 DCT_DCT bd 8."""
    ones = np.ones((4, 4), np.int16)
    out = np.zeros(16, np.int32)
    oracle.r1o_forward_transform(O.ptr(ones), O.ptr(out), 4, 0, 0, 8, 1)
    assert out[0] == 32 and not out[1:].any()
    imp = np.zeros((4, 4), np.int16)
    imp[0, 0] = 100
    oracle.r1o_forward_transform(O.ptr(imp), O.ptr(out), 4, 0, 0, 8, 1)
    assert list(out[:4]) == [200, 262, 200, 108]


@pytest.mark.parametrize("t,n", [(0, 4), (1, 8), (2, 16), (3, 32), (4, 64), (6, 8), (7, 16)])
def test_1d_close_to_real_transform(t, n):
    """Independent sanity: the lifting networks approximate the orthonormal
    DCT-II / DST-IV to within a few LSB."""
    from scipy.fft import dct, dst
    rng = np.random.default_rng(t)
    x = rng.integers(-4000, 4000, size=(500, n)).astype(np.int32)
    y = np.stack(F.TXFM_FUNCS[t]([np.ascontiguousarray(x[:, i]) for i in range(n)]), axis=1)
    ref = dct(x.astype(float), type=2, norm="ortho", axis=1) if t < 5 else \
        dst(x.astype(float), type=4, norm="ortho", axis=1)
    assert np.abs(y - ref).max() < 8


def test_generated_inc_is_current():
    """rav1e_amd/csrc/fwd_tx_1d.inc and oracle/fwd_tx_1d.inc are what
    tools/gen_tx1d.py emits from the (golden-pinned) NumPy restatement."""
    import subprocess
    import sys
    root = O.ROOT
    a = open(os.path.join(root, "oracle", "fwd_tx_1d.inc")).read()
    b = open(os.path.join(root, "rav1e_amd", "csrc", "fwd_tx_1d.inc")).read()
    assert a == b
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_tx1d
    txt, _ = gen_tx1d.trace(1)
    assert txt in a
