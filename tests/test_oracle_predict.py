"""Pin the intra-prediction oracle: the reference's own known answers
(src/predict.rs:1523-1693: 4x4, ten modes, 27 directional angles, pred_max) and
the vectors of an independent AV1-spec-formulation model for every size, angle
delta, edge filter and upsample path (tests/golden/gen_predict_golden.py)."""
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(__file__)
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]

# src/predict.rs:1525-1527: edge_buf[i] = (i + 32).saturating_sub(MAX_TX_SIZE * 2)
KAT_EDGE = np.array([max(i + 32 - 128, 0) for i in range(257)], np.uint8)
ANGLES = [3, 6, 9, 14, 17, 20, 23, 26, 29, 32, 36, 39, 42, 45, 48, 51, 54, 58, 61,
          64, 67, 70, 73, 76, 81, 84, 87]
# expected outputs copied as DATA from the reference test (predict.rs:1573-1601)
ANGLE_EXPECTED = [
    [40] * 16, [40] * 16, [39] + [40] * 15, [37, 38, 39] + [40] * 13, [36, 37, 38, 39] + [40] * 12,
    [36, 37, 38, 39, 39] + [40] * 11, [35, 36, 37, 38, 38, 39] + [40] * 10,
    [35, 36, 37, 38, 37, 38, 39, 40, 39] + [40] * 7,
    [35, 36, 37, 38, 37, 38, 39, 40, 38, 39] + [40] * 6,
    [35, 36, 37, 38, 36, 37, 38, 39, 38, 39, 40, 40, 39, 40, 40, 40],
    [34, 35, 36, 37, 36, 37, 38, 39, 37, 38, 39, 40, 39, 40, 40, 40],
    [34, 35, 36, 37, 36, 37, 38, 39, 37, 38, 39, 40, 38, 39, 40, 40],
    [34, 35, 36, 37, 35, 36, 37, 38, 36, 37, 38, 39, 37, 38, 39, 40],
    [34, 35, 36, 37, 35, 36, 37, 38, 36, 37, 38, 39, 37, 38, 39, 40],
    [34, 35, 36, 37, 35, 36, 37, 38, 36, 37, 38, 39, 37, 38, 39, 40],
    [34, 35, 36, 37, 35, 36, 37, 38, 35, 36, 37, 38, 36, 37, 38, 39],
    [34, 35, 36, 37, 34, 35, 36, 37, 35, 36, 37, 38, 36, 37, 38, 39],
    [34, 35, 36, 37, 34, 35, 36, 37, 35, 36, 37, 38, 36, 37, 38, 39],
    [34, 35, 36, 37, 34, 35, 36, 37, 35, 36, 37, 38, 35, 36, 37, 38],
    [33, 34, 35, 36, 34, 35, 36, 37, 34, 35, 36, 37, 35, 36, 37, 38],
    [33, 34, 35, 36, 34, 35, 36, 37, 34, 35, 36, 37, 35, 36, 37, 38],
    [33, 34, 35, 36, 34, 35, 36, 37, 34, 35, 36, 37, 34, 35, 36, 37],
    [33, 34, 35, 36, 34, 35, 36, 37, 34, 35, 36, 37, 34, 35, 36, 37],
    [33, 34, 35, 36, 33, 34, 35, 36, 34, 35, 36, 37, 34, 35, 36, 37],
    [33, 34, 35, 36, 33, 34, 35, 36, 34, 35, 36, 37, 34, 35, 36, 37],
    [33, 34, 35, 36] * 4, [33, 34, 35, 36] * 4]
KAT = [  # (mode, variant, angle, expected) -- predict.rs:1530-1565
    (0, 3, 0, [32] * 16), (0, 2, 0, [35] * 16), (0, 1, 0, [30] * 16), (0, 0, 0, [128] * 16),
    (1, 3, 90, [33, 34, 35, 36] * 4),
    (2, 3, 180, [31] * 4 + [30] * 4 + [29] * 4 + [28] * 4),
    (12, 3, 0, [32, 34, 35, 36, 30, 32, 32, 36, 29, 32, 32, 32, 28, 28, 32, 32]),
    (9, 3, 0, [32, 34, 35, 35, 30, 32, 33, 34, 29, 31, 32, 32, 29, 30, 32, 32]),
    (11, 3, 0, [31, 33, 34, 35, 30, 33, 34, 35, 29, 32, 34, 34, 28, 31, 33, 34]),
    (10, 3, 0, [33, 34, 35, 36, 31, 31, 32, 33, 30, 30, 30, 31, 29, 30, 30, 30]),
]


def run(oracle, mode, variant, angle, edge, ts=0, bd=8, ief=0, left_len=4, above_len=4, ac=None,
        aw=None, ah=None):
    w, h = TX_W[ts], TX_H[ts]
    hbd = edge.dtype == np.uint16
    out = np.zeros((h, w), edge.dtype)
    rc = oracle.r1o_dispatch_predict_intra(mode, variant, O.ptr(out), w, ts, bd,
                                           O.ptr(ac) if ac is not None else None, angle, ief,
                                           O.ptr(edge), left_len, above_len, aw or w, ah or h,
                                           int(hbd))
    assert rc == 0
    return out


def test_reference_known_answers_4x4(oracle):
    for mode, variant, angle, want in KAT:
        assert run(oracle, mode, variant, angle, KAT_EDGE).ravel().tolist() == want, (mode, variant)
    for angle, want in zip(ANGLES, ANGLE_EXPECTED):
        got = run(oracle, 3, 3, angle, KAT_EDGE, left_len=8, above_len=8)
        assert got.ravel().tolist() == want, angle


def test_reference_pred_max(oracle):
    """predict.rs:1621-1693: all-max 12-bit edges predict all-max blocks."""
    edge = np.full(257, 4095, np.uint16)
    for mode, angle in ((0, 0), (2, 180), (1, 90), (12, 0), (9, 0), (11, 0), (10, 0)):
        assert (run(oracle, mode, 3, angle, edge, bd=12) == 4095).all()


@pytest.mark.parametrize("fixture", ["predict_ref", "predict_golden"])
def test_spec_model_vectors(oracle, fixture):
    """predict_ref.npz: outputs of the reference's own source text (gen_predict_ref.py executes
    src/predict.rs dispatch_predict_intra and every kernel below it); predict_golden.npz: the
    independent AV1-spec-formulation model."""
    Z = np.load(os.path.join(HERE, "golden", fixture + ".npz"))
    G = {k: Z[k] for k in Z.files}      # decompress once
    n = len(G["ts"])
    assert n > 2000
    for i in range(n):
        ts, bd = int(G["ts"][i]), int(G["bd"][i])
        w, h = TX_W[ts], TX_H[ts]
        edge = G["edges"][i] if bd > 8 else G["edges"][i].astype(np.uint8)
        ac = None
        if G["ac_off"][i + 1] > G["ac_off"][i]:
            ac = np.ascontiguousarray(G["ac"][G["ac_off"][i]:G["ac_off"][i + 1]])
        got = run(oracle, int(G["mode"][i]), int(G["variant"][i]), int(G["angle"][i]),
                  np.ascontiguousarray(edge), ts, bd, int(G["ief"][i]), int(G["left_len"][i]),
                  int(G["above_len"][i]), ac, int(G["avail_w"][i]), int(G["avail_h"][i]))
        want = G["out"][G["off"][i]:G["off"][i] + w * h].reshape(h, w)
        assert np.array_equal(got.astype(np.uint16), want), \
            (i, ts, int(G["mode"][i]), int(G["angle"][i]), int(G["ief"][i]), bd)


def test_predict_intra_remaps(oracle):
    """PAETH degrades by position, CFL with alpha 0 is DC (predict.rs:225-235)."""
    rng = np.random.default_rng(1)
    edge = rng.integers(0, 256, 257).astype(np.uint8)
    for (x, y, equiv_mode, variant, angle) in ((0, 0, 0, 0, 0), (0, 8, 1, 2, 90), (8, 0, 2, 1, 180),
                                               (8, 8, 12, 3, 0)):
        out = np.zeros((8, 8), np.uint8)
        oracle.r1o_predict_intra(12, x, y, O.ptr(out), 8, 1, 8, None, 0, 0, 0, O.ptr(edge), 8, 8,
                                 8, 8, 0)
        assert np.array_equal(out, run(oracle, equiv_mode, variant, angle, edge, ts=1, left_len=8,
                                       above_len=8))


def test_get_intra_edges_geometry(oracle):
    """Frame-edge base values 128<<(bd-8) +-1, replication beyond the visible
    rectangle, top-right / bottom-left availability (partition.rs:639-898)."""
    rng = np.random.default_rng(2)
    for hbd, bd in ((0, 8), (1, 10)):
        dt = np.uint16 if hbd else np.uint8
        tile = rng.integers(0, 1 << bd, (64, 96)).astype(dt)
        edge = np.zeros(257, dt)
        lens = (O.C.c_int * 2)()
        base = 128 << (bd - 8)
        # top-left block of the tile: nothing available
        oracle.r1o_get_intra_edges(O.ptr(edge), lens, O.ptr(tile), 96, 0, 0, 96, 64, 1, bd, -1, 0, 0,
                                   0, 0, hbd)
        assert list(lens) == [16, 16]
        assert (edge[129:129 + 8] == base - 1).all() and (edge[120:128] == base + 1).all()
        assert edge[128] == base
        # interior block, everything available
        oracle.r1o_get_intra_edges(O.ptr(edge), lens, O.ptr(tile), 96, 16, 24, 96, 64, 1, bd, -1, 0,
                                   0, 1, 1, hbd)
        assert np.array_equal(edge[129:129 + 16], tile[23, 16:32])
        assert np.array_equal(edge[112:128][::-1], tile[24:40, 15])
        assert edge[128] == tile[23, 15]
        # no top-right / bottom-left: replicate the last available pixel
        oracle.r1o_get_intra_edges(O.ptr(edge), lens, O.ptr(tile), 96, 16, 24, 96, 64, 1, bd, -1, 0,
                                   0, 0, 0, hbd)
        assert (edge[129 + 8:129 + 16] == tile[23, 23]).all()
        assert (edge[112:120] == tile[31, 15]).all()
        # DC at the left tile edge needs no left; V_PRED needs no left / top-left
        oracle.r1o_get_intra_edges(O.ptr(edge), lens, O.ptr(tile), 96, 0, 24, 96, 64, 1, bd, 0, 0, 0,
                                   0, 0, hbd)
        assert list(lens) == [0, 8]


def test_get_intra_edges_reference_vectors(oracle):
    """get_intra_edges as the reference's own text computes it (src/partition.rs:639-898
    executed by gen_predict_ref.py), with the reference's has_top_right / has_bottom_left
    answers passed in as flags."""
    Z = np.load(os.path.join(HERE, "golden", "predict_ref.npz"))
    G = {k: Z[k] for k in Z.files if k.startswith("e_")}
    n = len(G["e_x"])
    assert n > 400
    for i in range(n):
        bd = int(G["e_bd"][i])
        hbd = int(bd > 8)
        tile = np.ascontiguousarray(G["e_tile_%d" % G["e_case"][i]].astype(np.uint16 if hbd else np.uint8))
        edge = np.full(257, 0xFFFF if hbd else 0xFF, tile.dtype)
        lens = (O.C.c_int * 2)()
        oracle.r1o_get_intra_edges(O.ptr(edge), lens, O.ptr(tile), tile.shape[1], int(G["e_x"][i]),
                                   int(G["e_y"][i]), int(G["e_rect_w"][i]), int(G["e_rect_h"][i]),
                                   int(G["e_ts"][i]), bd, int(G["e_mode"][i]), int(G["e_enable_ief"][i]),
                                   int(G["e_angle_delta"][i]), int(G["e_has_tr"][i]), int(G["e_has_bl"][i]),
                                   hbd)
        ll, al = int(G["e_left_len"][i]), int(G["e_above_len"][i])
        assert list(lens) == [ll, al], i
        want = G["e_edges"][i]
        assert np.array_equal(edge[128 - ll:129 + al].astype(np.uint16), want[128 - ll:129 + al]), \
            (i, int(G["e_mode"][i]), int(G["e_x"][i]), int(G["e_y"][i]))


def test_pred_cfl_ac_reference_vectors(oracle):
    """pred_cfl_ac::<T, XDEC, YDEC> of the reference's text (src/predict.rs:1020-1063)."""
    Z = np.load(os.path.join(HERE, "golden", "predict_ref.npz"))
    G = {k: Z[k] for k in Z.files if k.startswith("a_")}
    for i in range(len(G["a_bd"])):
        bd, bw, bh = int(G["a_bd"][i]), int(G["a_bw"][i]), int(G["a_bh"][i])
        xdec, ydec = int(G["a_xdec"][i]), int(G["a_ydec"][i])
        lw, lh = max(bw << xdec, 8), max(bh << ydec, 8)
        luma = G["a_luma"][G["a_luma_off"][i]:G["a_luma_off"][i + 1]].reshape(lh, lw)
        luma = np.ascontiguousarray(luma.astype(np.uint16 if bd > 8 else np.uint8))
        ac = np.zeros(bw * bh, np.int16)
        oracle.r1o_pred_cfl_ac(O.ptr(ac), O.ptr(luma), lw, bw, bh, int(G["a_w_pad"][i]),
                               int(G["a_h_pad"][i]), xdec, ydec, int(bd > 8))
        off = int(G["a_off"][i])
        assert np.array_equal(ac, G["a_out"][off:off + bw * bh]), i
