"""Pin the inverse-transform oracles (C and rule-based NumPy restatements)
against vectors produced by executing the reference's own source text:
  inv_tx_ref.npz     the WHOLE of inverse_transform_add (inverse.rs:1633-1705, 2-D driver
                     included) run through tools/rustlite (tests/golden/gen_inv_tx_ref.py)
  inv_tx_golden.npz  round 1: the 1-D networks executed from the text; its 2-D cases use a
                     hand-stated driver and are kept only as extra (wrapping-i32) inputs
and against the reference's own round-trip test (src/transform/mod.rs:555-603)."""
import os

import numpy as np
import pytest

import inv_tx_np as I
import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "inv_tx_golden.npz"))
GREF = np.load(os.path.join(os.path.dirname(__file__), "golden", "inv_tx_ref.npz"))
CLS = {"dct": 0, "adst": 1, "flipadst": 2, "identity": 3, "wht": 4}


def test_1d_networks_match_reference_source_vectors(oracle):
    keys = [k for k in G.files if k.startswith("d1_") and k.endswith("_in")]
    assert len(keys) == 48
    for k in keys:
        _, cls, n, rb, _ = k.split("_")
        n, rb = int(n), int(rb[1:])
        x, want = G[k], G[k[:-3] + "_out"]
        got = np.stack(I.inv_1d(CLS[cls], [x[:, i] for i in range(n)], rb), axis=1)
        assert np.array_equal(got, want), ("numpy", k)
        yc = np.ascontiguousarray(x.copy())
        for row in yc:
            assert oracle.r1o_inv_txfm_1d(O.ptr(row), CLS[cls], n, rb) == 0
        assert np.array_equal(yc, want), ("C", k)


def test_2d_reference_executed_driver(oracle):
    """REF-SRC pin of the 2-D driver: every (size, type, bit depth) the reference's
    INV_TXFM_FNS table implements, T::Coeff-typed coefficients, 3-5 blocks each."""
    keys = [k for k in GREF.files if k.endswith("_co")]
    assert len(keys) == 480
    for j, k in enumerate(keys):
        _, ts, tt, bd, _ = k.split("_")
        ts, tt, bd = int(ts), int(tt), int(bd)
        co, pred, rec = GREF[k], GREF[k[:-3] + "_pred"], GREF[k[:-3] + "_rec"]
        h, w = pred.shape[1:]
        hbd = int(bd > 8)
        assert co.dtype == (np.int32 if hbd else np.int16) and pred.dtype == (np.uint16 if hbd else np.uint8)
        if j % 5 == 0:
            got = I.inverse_transform_add(co.astype(np.int32), pred.astype(np.int32), ts, tt, bd)
            assert np.array_equal(got, rec.astype(np.int32)), ("numpy", k)
        for i in range(co.shape[0]):
            d = pred[i].copy()
            ci = np.ascontiguousarray(co[i])
            assert oracle.r1o_inverse_transform_add(O.ptr(ci), O.ptr(d), w, ts, tt, bd, hbd, hbd) == 0
            assert np.array_equal(d, rec[i]), ("C", k, i)
        # the batch entry the GPU parity tests compare against
        n = co.shape[0]
        stride = w * h
        cb = np.zeros((n, stride), co.dtype)
        cb[:, :co.shape[1]] = co
        want = np.zeros_like(pred)
        assert oracle.r1o_inv_txfm_add_batch(O.ptr(cb), stride, O.ptr(np.ascontiguousarray(pred)), O.ptr(want),
                                             n, ts, tt, bd, cb.itemsize, 1 + hbd) == 0
        assert np.array_equal(want, rec), ("C batch", k)


def test_2d_all_sizes_types_bitdepths(oracle):
    keys = [k for k in G.files if k.startswith("d2_") and k.endswith("_co")]
    assert len(keys) == 480
    for j, k in enumerate(keys):
        _, ts, tt, bd, _ = k.split("_")
        ts, tt, bd = int(ts), int(tt), int(bd)
        co = G[k]
        pred, rec = G[k[:-3] + "_pred"], G[k[:-3] + "_rec"]
        h, w = pred.shape[1:]
        if j % 7 == 0:   # the NumPy restatement is slow: sample it (all pass offline)
            got = I.inverse_transform_add(co, pred.astype(np.int32), ts, tt, bd)
            assert np.array_equal(got, rec.astype(np.int32)), ("numpy", k)
        for i in range(co.shape[0]):
            if bd == 8:
                d = np.ascontiguousarray(pred[i].astype(np.uint8))
                # T::Coeff = i16 for u8 pixels: only vectors that fit are comparable
                if np.abs(co[i]).max() < 32768:
                    c16 = np.ascontiguousarray(co[i].astype(np.int16))
                    assert oracle.r1o_inverse_transform_add(O.ptr(c16), O.ptr(d), w, ts, tt, bd, 0, 0) == 0
                    assert np.array_equal(d, rec[i].astype(np.uint8)), ("C u8", k, i)
            d = np.ascontiguousarray(pred[i].astype(np.uint16))
            c32 = np.ascontiguousarray(co[i])
            assert oracle.r1o_inverse_transform_add(O.ptr(c32), O.ptr(d), w, ts, tt, bd, 1, 1) == 0
            assert np.array_equal(d, rec[i]), ("C u16", k, i)


# the reference's own round-trip table (transform/mod.rs:559-603): (tx_size, tx_type, tolerance)
ROUNDTRIPS = [
    (0, 16, 0), (0, 0, 0), (0, 1, 0), (0, 2, 0), (0, 3, 0), (0, 4, 0), (0, 5, 0), (0, 9, 0),
    (0, 10, 0), (0, 11, 0), (0, 12, 0), (0, 13, 0),
    (1, 0, 1), (1, 1, 1), (1, 2, 1), (1, 3, 1), (1, 4, 1), (1, 5, 1), (1, 9, 0), (1, 10, 0),
    (1, 11, 0), (1, 12, 0), (1, 13, 1),
    (2, 0, 1), (2, 1, 1), (2, 2, 1), (2, 3, 1), (2, 4, 1), (2, 5, 1), (2, 9, 0), (2, 10, 1),
    (2, 11, 1),
    (3, 0, 2), (3, 9, 0),
    (5, 0, 1), (6, 0, 1), (13, 0, 1), (14, 0, 1), (7, 0, 1), (8, 0, 1), (15, 0, 2), (16, 0, 2),
    (9, 0, 2), (10, 0, 2),
]


@pytest.mark.parametrize("hbd", [0, 1])
def test_reference_roundtrips(oracle, hbd):
    """forward_transform -> inverse_transform_add reproduces the source within
    the tolerances the reference asserts for its own kernels."""
    rng = np.random.default_rng(11)
    for ts, tt, tol in ROUNDTRIPS:
        w, h = I.TX_W[ts], I.TX_H[ts]
        for _ in range(8):
            src = rng.integers(0, 256, (h, w))
            dst = rng.integers(0, 256, (h, w))
            res = np.ascontiguousarray((src - dst).astype(np.int16))
            freq = np.zeros(w * h, np.int32 if hbd else np.int16)
            assert oracle.r1o_forward_transform(O.ptr(res), O.ptr(freq), w, ts, tt, 8, hbd) == 0
            d = np.ascontiguousarray(dst.astype(np.uint16 if hbd else np.uint8))
            assert oracle.r1o_inverse_transform_add(O.ptr(freq), O.ptr(d), w, ts, tt, 8, hbd, hbd) == 0
            assert np.abs(d.astype(np.int32) - src).max() <= tol, (ts, tt)


def test_dc_only_is_flat(oracle):
    """A DC-only block reconstructs to a constant offset (any size)."""
    for ts in range(19):
        w, h = I.TX_W[ts], I.TX_H[ts]
        co = np.zeros(min(w, 32) * min(h, 32), np.int32)
        co[0] = 1000
        d = np.full((h, w), 300, np.uint16)
        oracle.r1o_inverse_transform_add(O.ptr(co), O.ptr(d), w, ts, 0, 10, 1, 1)
        assert (d == d[0, 0]).all() and d[0, 0] != 300, ts


def test_generated_inc_is_current():
    import sys
    root = O.ROOT
    a = open(os.path.join(root, "oracle", "inv_tx_1d.inc")).read()
    b = open(os.path.join(root, "rav1e_amd", "csrc", "inv_tx_1d.inc")).read()
    assert a == b
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_inv_tx1d
    gen_inv_tx1d.install_zero_folding()
    txt, _ = gen_inv_tx1d.trace("idct16", 0, 16, 16)
    assert txt in a
