"""GPU parity: every batch entry point of librav1e_hip.so, called through the
C ABI, against the CPU oracle on the same seeded inputs (bit-exact), plus the
committed golden fixtures and the reference's known-answer tables."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
BLOCK_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16),
               (32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128),
               (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
TX_SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8),
            (16, 32), (32, 16), (32, 64), (64, 32), (4, 16), (16, 4), (8, 32), (32, 8),
            (16, 64), (64, 16)]


def dev_plane(hp):
    from rav1e_amd.api import Plane
    return Plane.from_numpy(hp.data, hp.width, hp.height, hp.bit_depth, hp.xpad, hp.ypad)


def planes(bd, w=320, h=192, seed=0, pads=(88, 88)):
    rng = np.random.default_rng(seed)
    a = O.HostPlane(w, h, bd, pads[0], pads[0], rng=rng)
    b = O.HostPlane(w, h, bd, pads[1], pads[1], rng=rng)
    return a, b


def rand_dist_cands(rng, n, pw, ph, w, h, slack):
    c = np.zeros(n, O.DIST_CAND)
    c["ox"] = rng.integers(0, pw - w + 1, n)
    c["oy"] = rng.integers(0, ph - h + 1, n)
    c["rx"] = rng.integers(-slack, pw - w + slack + 1, n)   # may reach into the padding
    c["ry"] = rng.integers(-slack, ph - h + slack + 1, n)
    return c


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_dist_batch_all_block_sizes(ctx, oracle, bd):
    a, b = planes(bd, seed=bd, pads=(88, 152))   # different strides / alignments
    da, db = dev_plane(a), dev_plane(b)
    rng = np.random.default_rng(100 + bd)
    for (w, h) in BLOCK_SIZES:
        n = 193 if w * h <= 1024 else 37    # ragged: not a multiple of any group size
        c = rand_dist_cands(rng, n, a.width, a.height, w, h, 40)
        for kind in (0, 1):
            want = np.zeros(n, np.uint32)
            pa, pb = a.cstruct(), b.cstruct()
            import ctypes as C
            assert oracle.r1o_dist_batch(kind, C.byref(pa), C.byref(pb), w, h, O.ptr(c), n,
                                         O.ptr(want)) == 0
            got = ctx.dist_batch(kind, da, db, w, h, c).cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want), (bd, w, h, kind)


@pytest.mark.parametrize("bd", [8, 10])
def test_dist_reference_known_answers(ctx, bd):
    """The reference's own SAD/SATD tables (src/dist.rs:418-441,477-500)."""
    from test_oracle_dist import SAD, SATD, reference_test_planes
    a, b = reference_test_planes(bd)
    da, db = dev_plane(a), dev_plane(b)
    c = np.zeros(1, O.DIST_CAND)
    c["ox"], c["oy"], c["rx"], c["ry"] = 32, 40, 32, 40
    for kind, table in ((0, SAD), (1, SATD)):
        for w, h, v in table:
            got = int(ctx.dist_batch(kind, da, db, w, h, c).cpu().numpy().view(np.uint32)[0])
            assert got == v, (kind, w, h)


def test_dist_empty_and_bad_args(ctx):
    from rav1e_amd.api import R1Error
    a, b = planes(8)
    da, db = dev_plane(a), dev_plane(b)
    assert ctx.dist_batch(0, da, db, 8, 8, np.zeros(0, O.DIST_CAND)).numel() == 0
    with pytest.raises(R1Error):
        ctx.dist_batch(0, da, db, 12, 8, np.zeros(1, O.DIST_CAND))   # not a block size
    with pytest.raises(R1Error):
        ctx.dist_batch(7, da, db, 8, 8, np.zeros(1, O.DIST_CAND))


def test_fwd_txfm_golden(ctx):
    """All 480 (bd, size, type) cases derived from the reference source text."""
    import torch
    G = np.load(os.path.join(GOLD, "fwd_tx_golden.npz"))
    for k in G["keys2d"]:
        bd, ts, tt = map(int, k.split("_"))
        res = torch.from_numpy(np.ascontiguousarray(G["res2d_" + k])).cuda()
        want = G["coef2d_" + k]
        got = ctx.forward_transform_batch(res, ts, tt, bd, coeff_bytes=4).cpu().numpy()[0]
        assert np.array_equal(got, want), k
        if bd == 8:
            got16 = ctx.forward_transform_batch(res, ts, tt, bd, coeff_bytes=2).cpu().numpy()[0]
            assert np.array_equal(got16, want.astype(np.int16)), k


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_fwd_txfm_batch_vs_oracle(ctx, oracle, bd):
    import torch
    from rav1e_amd.types import valid_av1_transform
    rng = np.random.default_rng(bd)
    lim = (1 << bd) - 1
    cb = 2 if bd == 8 else 4
    for ts, (w, h) in enumerate(TX_SIZES):
        for tt in range(17):
            if not valid_av1_transform(ts, tt):
                continue
            n = 67 if w * h <= 256 else 9    # ragged vs blocks-per-wave
            res = rng.integers(-lim, lim + 1, size=(n, h, w)).astype(np.int16)
            want = np.zeros((n, w * h), np.int16 if cb == 2 else np.int32)
            assert oracle.r1o_fwd_txfm_batch(O.ptr(res), O.ptr(want), n, ts, tt, bd, cb) == 0
            got = ctx.forward_transform_batch(torch.from_numpy(res).cuda(), ts, tt, bd, cb)
            assert np.array_equal(got.cpu().numpy(), want), (bd, ts, tt)


def test_fwd_txfm_full_i16_range_wraps_like_rust_release(ctx, oracle):
    """i32 wrapping semantics for arbitrary int16 input (forward.rs:42-44)."""
    import torch
    rng = np.random.default_rng(5)
    for ts in (0, 1, 2, 3, 4, 11, 18):
        w, h = TX_SIZES[ts]
        res = rng.integers(-32768, 32768, size=(5, h, w)).astype(np.int16)
        want = np.zeros((5, w * h), np.int32)
        oracle.r1o_fwd_txfm_batch(O.ptr(res), O.ptr(want), 5, ts, 0, 12, 4)
        got = ctx.forward_transform_batch(torch.from_numpy(res).cuda(), ts, 0, 12, 4)
        assert np.array_equal(got.cpu().numpy(), want), ts


def test_fwd_txfm_invalid_pair_is_einval(ctx):
    import torch
    from rav1e_amd.api import R1Error
    with pytest.raises(R1Error):
        ctx.forward_transform_batch(torch.zeros((1, 64, 64), dtype=torch.int16).cuda(), 4, 1, 8)


def rand_mc_cands(rng, n, pw, ph, w, h, slack):
    c = np.zeros(n, O.MC_CAND)
    c["rx"] = rng.integers(-slack, pw - w + slack + 1, n)
    c["ry"] = rng.integers(-slack, ph - h + slack + 1, n)
    c["col_frac"] = rng.integers(0, 16, n)
    c["row_frac"] = rng.integers(0, 16, n)
    c["mode_x"] = rng.integers(0, 4, n)
    c["mode_y"] = rng.integers(0, 4, n)
    # make sure the four structural cases all occur
    c["col_frac"][: n // 4] = 0
    c["row_frac"][n // 8: n // 4 + n // 8] = 0
    return c


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_mc_put_prep_avg_vs_oracle(ctx, oracle, bd):
    import ctypes as C
    a, _ = planes(bd, seed=20 + bd)
    da = dev_plane(a)
    rng = np.random.default_rng(200 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    sizes = [(2, 2), (2, 4), (4, 2), (4, 4), (4, 8), (8, 4), (8, 8), (16, 8), (8, 16), (16, 16),
             (32, 32), (64, 64), (128, 128), (128, 64), (64, 128), (16, 64), (4, 16), (32, 8)]
    for (w, h) in sizes:
        n = 41 if w * h <= 1024 else 7
        c = rand_mc_cands(rng, n, a.width, a.height, w, h, 60)
        pa = a.cstruct()
        want_put = np.zeros((n, h, w), dt)
        want_prep = np.zeros((n, h, w), np.int16)
        assert oracle.r1o_mc_put_batch(C.byref(pa), w, h, O.ptr(c), n, O.ptr(want_put)) == 0
        assert oracle.r1o_mc_prep_batch(C.byref(pa), w, h, O.ptr(c), n, O.ptr(want_prep)) == 0
        got_put = ctx.put_8tap_batch(da, w, h, c).cpu().numpy().view(dt)
        got_prep = ctx.prep_8tap_batch(da, w, h, c)
        assert np.array_equal(got_put, want_put), (bd, w, h, "put")
        assert np.array_equal(got_prep.cpu().numpy(), want_prep), (bd, w, h, "prep")
        # compound average of the prep output with a shuffled copy of itself
        import torch
        t2 = got_prep.flip(0).contiguous()
        want_avg = np.zeros((n, h, w), dt)
        oracle.r1o_mc_avg_batch(O.ptr(want_prep), O.ptr(np.ascontiguousarray(want_prep[::-1])),
                                w, h, n, bd, a.bpp, O.ptr(want_avg))
        got_avg = ctx.mc_avg_batch(got_prep, t2, w, h, bd).cpu().numpy().view(dt)
        assert np.array_equal(got_avg, want_avg), (bd, w, h, "avg")


@pytest.mark.parametrize("fixture", ["mc_ref", "mc_golden"])
def test_mc_golden(ctx, fixture):
    """mc_ref.npz: outputs of the reference's own source text (gen_mc_ref.py executes
    src/mc.rs); mc_golden.npz: reference tap data + an independent model."""
    G = np.load(os.path.join(GOLD, fixture + ".npz"))
    from rav1e_amd.api import Plane
    for k in G["cases"]:
        bd, w, h, cf, rf, mx, my, _ = map(int, k.split("_"))
        win = G["win_" + k]
        hp = O.HostPlane(w, h, bd, 4, 4)           # window = block + 3/4 px of padding
        # place the window so that block origin (0,0) sits at win[3,3]
        hp.data[hp.yorigin - 3: hp.yorigin + h + 4, hp.xorigin - 3: hp.xorigin + w + 4] = win
        dp = Plane.from_numpy(hp.data, w, h, bd, 4, 4)
        c = np.zeros(1, O.MC_CAND)
        c["col_frac"], c["row_frac"], c["mode_x"], c["mode_y"] = cf, rf, mx, my
        dt = np.uint8 if bd == 8 else np.uint16
        put = ctx.put_8tap_batch(dp, w, h, c).cpu().numpy().view(dt)[0]
        prep = ctx.prep_8tap_batch(dp, w, h, c).cpu().numpy()[0]
        assert np.array_equal(put, G["put_" + k]), k
        assert np.array_equal(prep, G["prep_" + k]), k


def rand_rdo_cands(rng, n, pw, ph, w, h, slack, ts):
    from rav1e_amd.types import valid_av1_transform
    c = np.zeros(n, O.RDO_CAND)
    c["ox"] = rng.integers(0, pw - w + 1, n)
    c["oy"] = rng.integers(0, ph - h + 1, n)
    c["rx"] = rng.integers(-slack, pw - w + slack + 1, n)
    c["ry"] = rng.integers(-slack, ph - h + slack + 1, n)
    c["col_frac"] = rng.integers(0, 16, n)
    c["row_frac"] = rng.integers(0, 16, n)
    c["col_frac"][: n // 4] = 0
    c["row_frac"][n // 8: n // 4 + n // 8] = 0
    c["mode_x"] = rng.integers(0, 4, n)
    c["mode_y"] = rng.integers(0, 4, n)
    valid = [t for t in range(16) if valid_av1_transform(ts, t)]
    c["tx_type"] = rng.choice(valid, n)
    return c


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_rdo_cand_fused_vs_oracle(ctx, oracle, bd):
    import ctypes as C
    a, b = planes(bd, seed=40 + bd, pads=(88, 120))
    da, db = dev_plane(a), dev_plane(b)
    rng = np.random.default_rng(300 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    ct = np.int16 if bd == 8 else np.int32
    for ts, (w, h) in enumerate(TX_SIZES):
        n = 45 if w * h <= 1024 else 11
        c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 50, ts)
        pa, pb = a.cstruct(), b.cstruct()
        wsad, wsatd = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        wco = np.zeros((n, w * h), ct)
        wpred = np.zeros((n, h, w), dt)
        assert oracle.r1o_rdo_cand_batch(C.byref(pa), C.byref(pb), w, h, ts, O.ptr(c), n,
                                         O.ptr(wsad), O.ptr(wsatd), O.ptr(wco), O.ptr(wpred)) == 0
        o = ctx.rdo_cand_batch(da, db, w, h, c, want_pred=True)
        assert np.array_equal(o["pred"].cpu().numpy().view(dt), wpred), (bd, w, h, "pred")
        assert np.array_equal(o["sad"].cpu().numpy().view(np.uint32), wsad), (bd, w, h, "sad")
        assert np.array_equal(o["satd"].cpu().numpy().view(np.uint32), wsatd), (bd, w, h, "satd")
        assert np.array_equal(o["coeffs"].cpu().numpy(), wco), (bd, w, h, "coeffs")
        # partial outputs: distortion only (the ME / pre-screen use)
        o2 = ctx.rdo_cand_batch(da, db, w, h, c, want_coeffs=False, want_sad=False)
        assert np.array_equal(o2["satd"].cpu().numpy().view(np.uint32), wsatd)


def test_rdo_cand_zero_mv_residual_properties(ctx):
    """Size-independent property at full frame size: a candidate predicting a
    plane from itself at zero MV has zero SAD/SATD and all-zero coefficients;
    linearity: SAD equals the sum over its four quadrant candidates."""
    a, _ = planes(8, 1920, 1080, seed=9)
    da = dev_plane(a)
    rng = np.random.default_rng(9)
    n = 4096
    c = np.zeros(n, O.RDO_CAND)
    c["ox"] = rng.integers(0, 1920 - 64, n)
    c["oy"] = rng.integers(0, 1080 - 64, n)
    c["rx"], c["ry"] = c["ox"], c["oy"]
    o = ctx.rdo_cand_batch(da, da, 64, 64, c)
    assert not o["sad"].any() and not o["satd"].any() and not o["coeffs"].any()
    # linearity of SAD over a 2x2 split, different planes
    b = O.HostPlane(1920, 1080, 8, rng=np.random.default_rng(10))
    db = dev_plane(b)
    c["rx"] = rng.integers(-8, 1920 - 64 + 8, n)
    c["ry"] = rng.integers(-8, 1080 - 64 + 8, n)
    c["col_frac"] = rng.integers(0, 16, n)
    c["row_frac"] = rng.integers(0, 16, n)
    # quadrant candidates must use the same filter family: 64 and 32 are both > 4
    big = ctx.rdo_cand_batch(da, db, 64, 64, c, want_coeffs=False, want_satd=False)["sad"]
    tot = None
    for qx in (0, 32):
        for qy in (0, 32):
            q = c.copy()
            q["ox"] += qx; q["oy"] += qy; q["rx"] += qx; q["ry"] += qy
            s = ctx.rdo_cand_batch(da, db, 32, 32, q, want_coeffs=False, want_satd=False)["sad"]
            tot = s.clone() if tot is None else tot + s
    assert bool((tot == big).all())


def test_compat_shims_host_pointers(ctx, oracle):
    """Reference asm signatures (byte strides, host pointers)."""
    from rav1e_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(11)
    for bd in (8, 10):
        a, b = planes(bd, 128, 128, seed=60 + bd, pads=(16, 24))
        for (w, h) in ((4, 4), (8, 8), (16, 32), (64, 64), (128, 128)):
            x, y = int(rng.integers(0, 128 - w + 1)), int(rng.integers(0, 128 - h + 1))
            args = (a.block_ptr(x, y), a.stride * a.bpp, b.block_ptr(x, y), b.stride * b.bpp, w, h)
            oa = (a.block_ptr(x, y), a.stride, b.block_ptr(x, y), b.stride, w, h, int(bd > 8))
            if bd == 8:
                assert L.rav1e_sad_hip(*args) == oracle.r1o_get_sad(*oa)
                assert L.rav1e_satd_hip(*args) == oracle.r1o_get_satd(*oa)
            else:
                assert L.rav1e_sad_hbd_hip(*args) == oracle.r1o_get_sad(*oa)
                assert L.rav1e_satd_hbd_hip(*args, 1023) == oracle.r1o_get_satd(*oa)
        # put_8tap shim at the reference's bench MVs (benches/mc.rs:30-127)
        dt = np.uint8 if bd == 8 else np.uint16
        for (mx, my) in ((0, 0), (0, 4), (4, 0), (4, 4)):
            want = np.zeros((16, 16), dt)
            got = np.zeros((16, 16), dt)
            oracle.r1o_put_8tap(O.ptr(want), 16, a.block_ptr(40, 40), a.stride, 16, 16, mx, my,
                                0, 0, bd, int(bd > 8))
            if bd == 8:
                L.rav1e_put_8tap_hip(O.ptr(got), 16, a.block_ptr(40, 40), a.stride, 16, 16, mx,
                                     my, 0, 0)
            else:
                L.rav1e_put_8tap_hbd_hip(O.ptr(got), 32, a.block_ptr(40, 40), a.stride * 2, 16,
                                         16, mx, my, 0, 0, 1023)
            assert np.array_equal(got, want), (bd, mx, my)
    res = rng.integers(-255, 256, size=(8, 8)).astype(np.int16)
    want = np.zeros(64, np.int16)
    got = np.zeros(64, np.int16)
    oracle.r1o_forward_transform(O.ptr(res), O.ptr(want), 8, 1, 0, 8, 0)
    assert L.rav1e_fwd_txfm_hip(O.ptr(res), O.ptr(got), 8, 1, 0, 8, 2) == 0
    assert np.array_equal(got, want)


# ---------------------------------------------------------------- inverse tx
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_inv_txfm_golden_vectors(ctx):
    """480 (size, type, bit depth) cases produced by executing the reference's
    own inverse.rs text (tests/golden/gen_inv_tx_golden.py)."""
    G = np.load(os.path.join(GOLD, "inv_tx_golden.npz"))
    keys = [k for k in G.files if k.startswith("d2_") and k.endswith("_co")]
    assert len(keys) == 480
    for k in keys:
        _, ts, tt, bd, _ = k.split("_")
        ts, tt, bd = int(ts), int(tt), int(bd)
        co, pred, rec = G[k], G[k[:-3] + "_pred"], G[k[:-3] + "_rec"]
        if bd == 8:
            ok = np.abs(co).max(axis=1) < 32768       # T::Coeff = i16
            if not ok.any():
                continue
            got = ctx.inverse_transform_add_batch(_t(co[ok].astype(np.int16)),
                                                  _t(pred[ok].astype(np.uint8)), ts, tt, bd)
            assert np.array_equal(got.cpu().numpy(), rec[ok].astype(np.uint8)), k
        else:
            got = ctx.inverse_transform_add_batch(_t(co), _t(pred.astype(np.int16)), ts, tt, bd)
            assert np.array_equal(got.cpu().numpy().view(np.uint16), rec), k


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_inv_txfm_random_vs_oracle(ctx, oracle, bd):
    rng = np.random.default_rng(300 + bd)
    bpp = 1 if bd == 8 else 2
    ct = np.int16 if bd == 8 else np.int32
    pt = np.uint8 if bd == 8 else np.uint16
    for ts, (w, h) in enumerate(TX_SIZES):
        area = min(w, 32) * min(h, 32)
        types = [t for t in range(17) if oracle.r1o_valid_av1_transform(ts, t)]
        for tt in types:
            n = 37
            stride = w * h                       # the forward transform's block stride
            co = np.zeros((n, stride), ct)
            amp = 1 << (bd + 4)
            co[:, :area] = rng.integers(-amp, amp, (n, area))
            co[: n // 3, :area] *= (rng.random((n // 3, area)) < 0.1)
            if ct == np.int32:
                co[-1, :area] = rng.integers(-(1 << 24), 1 << 24, area)   # clamp paths
            pred = rng.integers(0, 1 << bd, (n, h, w)).astype(pt)
            want = np.zeros_like(pred)
            assert oracle.r1o_inv_txfm_add_batch(O.ptr(co), stride, O.ptr(pred), O.ptr(want), n, ts,
                                                 tt, bd, co.itemsize, bpp) == 0
            dp = _t(pred if bpp == 1 else pred.view(np.int16))
            got = ctx.inverse_transform_add_batch(_t(co), dp, ts, tt, bd).cpu().numpy()
            assert np.array_equal(got.view(pt), want), (bd, ts, tt)


def test_fwd_inv_roundtrip_on_device(ctx):
    """the reference's own round-trip property (src/transform/mod.rs:555-603),
    entirely on the GPU: forward_transform -> inverse_transform_add."""
    import torch
    from test_oracle_inv_tx import ROUNDTRIPS
    rng = np.random.default_rng(9)
    for ts, tt, tol in ROUNDTRIPS:
        w, h = TX_SIZES[ts]
        n = 64
        src = rng.integers(0, 256, (n, h, w))
        dst = rng.integers(0, 256, (n, h, w))
        res = _t((src - dst).astype(np.int16))
        freq = ctx.forward_transform_batch(res, ts, tt, 8)
        rec = ctx.inverse_transform_add_batch(freq, _t(dst.astype(np.uint8)), ts, tt, 8)
        err = np.abs(rec.cpu().numpy().astype(np.int32) - src).max()
        assert err <= tol, (ts, tt, err)


# ------------------------------------------------------------------ quantize
@pytest.mark.parametrize("fixture,ncases", [("quant_ref", 894), ("quant_golden", 570)])
def test_quantize_golden_vectors(ctx, fixture, ncases):
    """quant_ref.npz: outputs of the reference's own source text (gen_quant_ref.py executes
    src/quantize/mod.rs + tables.rs + scan_order.rs); quant_golden.npz: independent model."""
    G = np.load(os.path.join(GOLD, fixture + ".npz"))
    keys = [k for k in G.files if k.endswith("_co")]
    assert len(keys) == ncases
    for k in keys:
        _, ts, tt, bd, intra, qi, dcd, acd, _ = k.split("_")
        ts, tt, bd, intra, qi, dcd, acd = map(int, (ts, tt, bd, intra, qi, dcd, acd))
        o = ctx.quantize_batch(_t(G[k]), ts, tt, qi, bd, intra, dcd, acd)
        assert np.array_equal(o["eobs"].cpu().numpy().view(np.uint16), G[k[:-3] + "_eob"]), k
        assert np.array_equal(o["qcoeffs"].cpu().numpy(), G[k[:-3] + "_q"]), k
        assert np.array_equal(o["rcoeffs"].cpu().numpy(), G[k[:-3] + "_r"]), k


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_quantize_random_vs_oracle(ctx, oracle, bd):
    rng = np.random.default_rng(500 + bd)
    ct = np.int16 if bd == 8 else np.int32
    for ts, (w, h) in enumerate(TX_SIZES):
        area = min(w, 32) * min(h, 32)
        for tt in (0, 10, 11) if max(w, h) <= 16 else (0,):
            for qi, intra in ((60, 0), (160, 1), (255, 0)):
                n = 203                                  # ragged vs every group size
                stride = w * h
                acq = oracle.r1o_ac_q(qi, 0, bd)
                co = np.zeros((n, stride), ct)
                v = rng.integers(-3 * acq, 3 * acq + 1, (n, area))
                v[: n // 2] *= (rng.random((n // 2, area)) < 0.07)
                v[n // 2: 3 * n // 4] = rng.integers(-2, 3, (3 * n // 4 - n // 2, area)) + \
                    rng.choice([0, acq // 2, acq, -acq], (3 * n // 4 - n // 2, area))
                co[:, :area] = np.clip(v, np.iinfo(ct).min, np.iinfo(ct).max)
                co[0] = 0
                q = np.zeros((n, area), ct)
                r = np.zeros((n, area), ct)
                eobs = np.zeros(n, np.uint16)
                assert oracle.r1o_quantize_batch(O.ptr(co), stride, n, ts, tt, qi, bd, intra, 0, 0,
                                                 co.itemsize, O.ptr(q), O.ptr(eobs), O.ptr(r)) == 0
                o = ctx.quantize_batch(_t(co), ts, tt, qi, bd, intra)
                assert np.array_equal(o["eobs"].cpu().numpy().view(np.uint16), eobs), (bd, ts, tt, qi)
                assert np.array_equal(o["qcoeffs"].cpu().numpy(), q), (bd, ts, tt, qi)
                assert np.array_equal(o["rcoeffs"].cpu().numpy(), r), (bd, ts, tt, qi)
                r2 = ctx.dequantize_batch(o["qcoeffs"], ts, qi, bd).cpu().numpy()
                assert np.array_equal(r2, r)


def test_quantize_rejects_wht_and_bad_pairs(ctx):
    import torch
    from rav1e_amd.api import R1Error
    co = torch.zeros((4, 16), dtype=torch.int16, device="cuda")
    with pytest.raises(R1Error):
        ctx.quantize_batch(co, 0, 16, 100, 8, 0)
    with pytest.raises(R1Error):
        ctx.quantize_batch(torch.zeros((4, 4096), dtype=torch.int16, device="cuda"), 4, 1, 100, 8, 0)


# ----------------------------------------------- weighted SSE / cdef_dist (a3-a5)
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_dist_scaled_batch(ctx, oracle, bd):
    import ctypes as C
    a, b = planes(bd, seed=40 + bd, pads=(88, 120))
    # make ref a noisy copy of org so variances / sse are in a realistic regime
    rng = np.random.default_rng(700 + bd)
    noise = rng.integers(-12, 13, a.data.shape)
    b.data = np.clip(a.data[:, :b.data.shape[1]].astype(np.int32) if False else
                     rng.integers(0, 1 << bd, b.data.shape), 0, (1 << bd) - 1).astype(b.data.dtype)
    b.view()[:] = np.clip(a.view().astype(np.int32) + noise[a.yorigin:a.yorigin + a.height,
                                                            a.xorigin:a.xorigin + a.width],
                          0, (1 << bd) - 1)
    da, db = dev_plane(a), dev_plane(b)
    gw, gh = (a.width + 7) // 8, (a.height + 7) // 8
    scales = rng.integers(1 << 10, 1 << 17, (gh, gw + 3)).astype(np.uint32)   # stride != width
    import torch
    dscales = torch.from_numpy(scales.view(np.int32)).cuda()
    sizes = BLOCK_SIZES + [(12, 20), (24, 8), (40, 64), (8, 12), (4, 4), (120, 72)]
    for (w, h) in sizes:
        n = 97 if w * h <= 1024 else 19
        c = np.zeros(n, O.DIST_CAND)
        c["ox"] = rng.integers(0, a.width - w + 1, n) & ~3
        c["oy"] = rng.integers(0, a.height - h + 1, n) & ~3
        c["rx"] = c["ox"] + rng.integers(-2, 3, n)
        c["ry"] = c["oy"] + rng.integers(-2, 3, n)
        pa, pb = a.cstruct(), b.cstruct()
        for kind, xdec, ydec, sc in ((2, 0, 0, True), (2, 0, 0, False), (3, 0, 0, True),
                                     (3, 0, 0, False), (2, 1, 1, True)):
            if xdec:   # chroma-style lookup: luma position = plane position << 1
                if 2 * w > a.width or 2 * h > a.height:
                    continue
                c2 = c.copy()   # a chroma block at (x, y) covers luma (2x, 2y)..(2x+2w, 2y+2h)
                c2["ox"] = rng.integers(0, (a.width - 2 * w) // 2 + 1, n) & ~3
                c2["oy"] = rng.integers(0, (a.height - 2 * h) // 2 + 1, n) & ~3
                c2["rx"], c2["ry"] = c2["ox"], c2["oy"]
            else:
                c2 = c
            want = np.zeros(n, np.uint64)
            assert oracle.r1o_dist_scaled_batch(kind, C.byref(pa), C.byref(pb), w, h, O.ptr(c2), n,
                                                O.ptr(scales) if sc else None, scales.shape[1],
                                                xdec, ydec, O.ptr(want)) == 0
            got = ctx.dist_scaled_batch(kind, da, db, w, h, c2, dscales if sc else None, xdec, ydec)
            assert np.array_equal(got.cpu().numpy().view(np.uint64), want), (bd, w, h, kind, xdec, sc)


def test_dist_scaled_identity_is_zero(ctx):
    """sse / cdef_dist of a block against itself is 0 for every size."""
    a, _ = planes(8, seed=3)
    da = dev_plane(a)
    for (w, h) in BLOCK_SIZES:
        c = np.zeros(8, O.DIST_CAND)
        c["ox"] = c["rx"] = np.arange(8) * 8
        c["oy"] = c["ry"] = 16
        for kind in (2, 3):
            assert not ctx.dist_scaled_batch(kind, da, da, w, h, c).cpu().numpy().any()


# ------------------------------------------------------------ intra prediction
def _intra_cands(modes, variants, angles, iefs, aws, ahs):
    from rav1e_amd.api import INTRA_CAND
    c = np.zeros(len(modes), INTRA_CAND)
    c["mode"], c["variant"], c["angle"], c["ief"] = modes, variants, angles, iefs
    c["avail_w"], c["avail_h"] = aws, ahs
    return c


def test_predict_reference_known_answers(ctx):
    """src/predict.rs:1523-1618 (4x4: ten modes + 27 directional angles)."""
    from test_oracle_predict import ANGLE_EXPECTED, ANGLES, KAT, KAT_EDGE
    modes = [k[0] for k in KAT] + [3] * len(ANGLES)
    variants = [k[1] for k in KAT] + [3] * len(ANGLES)
    angles = [k[2] for k in KAT] + ANGLES
    n = len(modes)
    c = _intra_cands(modes, variants, angles, [0] * n, [4] * n, [4] * n)
    edges = _t(np.tile(KAT_EDGE, (n, 1)))
    lens = _t(np.array([[4, 4]] * len(KAT) + [[8, 8]] * len(ANGLES), np.uint8))
    got = ctx.predict_intra_batch(0, c, edges, lens, 8).cpu().numpy().reshape(n, 16)
    want = np.array([k[3] for k in KAT] + ANGLE_EXPECTED)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("fixture", ["predict_ref", "predict_golden"])
def test_predict_spec_model_vectors(ctx, fixture):
    """predict_ref.npz: 4992 outputs of the reference's own source text (gen_predict_ref.py
    executes src/predict.rs); predict_golden.npz: 4272 vectors of the AV1-spec-formulation
    model.  Every size, angle delta, edge filter / upsample path."""
    Z = np.load(os.path.join(GOLD, fixture + ".npz"))
    G = {k: Z[k] for k in Z.files if not k.startswith(("e_", "a_"))}
    for ts, (w, h) in enumerate(TX_SIZES):
        for bd in (8, 10, 12):
            idx = np.nonzero((G["ts"] == ts) & (G["bd"] == bd))[0]
            if not len(idx):
                continue
            c = _intra_cands(G["mode"][idx], G["variant"][idx], G["angle"][idx], G["ief"][idx],
                             G["avail_w"][idx], G["avail_h"][idx])
            e = G["edges"][idx]
            edges = _t(e.astype(np.uint8) if bd == 8 else e.view(np.int16))
            lens = _t(np.stack([G["left_len"][idx], G["above_len"][idx]], 1).astype(np.uint8))
            ac = np.zeros((len(idx), w * h), np.int16)
            for k, i in enumerate(idx):
                a, b = G["ac_off"][i], G["ac_off"][i + 1]
                if b > a:
                    ac[k] = G["ac"][a:b]
            got = ctx.predict_intra_batch(ts, c, edges, lens, bd, ac=_t(ac)).cpu().numpy()
            got = got.astype(np.uint16) if bd == 8 else got.view(np.uint16)
            for k, i in enumerate(idx):
                want = G["out"][G["off"][i]:G["off"][i] + w * h].reshape(h, w)
                assert np.array_equal(got[k], want), (ts, bd, int(G["mode"][i]), int(G["angle"][i]),
                                                      int(G["ief"][i]))


@pytest.mark.parametrize("bd", [8, 12])
def test_intra_edges_and_predict_vs_oracle(ctx, oracle, bd):
    """get_intra_edges on a reconstructed tile (random geometry incl. tile /
    frame borders and clipped blocks), then predict_intra from those edges,
    both against the oracle call by call."""
    from rav1e_amd.api import INTRA_EDGE_CAND
    rng = np.random.default_rng(900 + bd)
    hp = O.HostPlane(200, 136, bd, rng=rng)        # frame not a multiple of 64
    dp = dev_plane(hp)
    hbd = int(bd > 8)
    dt = np.uint16 if hbd else np.uint8
    tiles = [(0, 0, 128, 136), (128, 0, 128, 136)]  # second tile is clipped by the frame
    for ts in (0, 1, 2, 3, 4, 5, 8, 13, 16, 9):
        w, h = TX_SIZES[ts]
        for (tx0, ty0, tw_, th_) in tiles:
            rect_w, rect_h = min(tw_, hp.width - tx0), min(th_, hp.height - ty0)
            n = 120
            ec = np.zeros(n, INTRA_EDGE_CAND)
            gx, gy = max(1, (rect_w + w - 1) // w), max(1, (rect_h + h - 1) // h)
            ec["x"] = rng.integers(0, gx, n) * w
            ec["y"] = rng.integers(0, gy, n) * h
            ec["x"][:8], ec["y"][:8] = 0, 0
            ec["mode"] = rng.integers(-1, 14, n)
            ec["angle_delta"] = np.where((ec["mode"] >= 1) & (ec["mode"] <= 8),
                                         rng.integers(-3, 4, n), 0)
            ec["flags"] = rng.integers(0, 8, n)
            edges, lens = ctx.intra_edges_batch(dp, (tx0, ty0, tw_, th_), ts, ec)
            ge, gl = edges.cpu().numpy().view(dt), lens.cpu().numpy()
            tile_ptr = hp.block_ptr(tx0, ty0)
            want_e = np.zeros((n, 257), dt)
            want_l = np.zeros((n, 2), np.int32)
            for i in range(n):
                li = (O.C.c_int * 2)()
                f = int(ec["flags"][i])
                oracle.r1o_get_intra_edges(O.ptr(want_e[i]), li, tile_ptr, hp.stride,
                                           int(ec["x"][i]), int(ec["y"][i]), rect_w, rect_h, ts, bd,
                                           int(ec["mode"][i]), f & 1, int(ec["angle_delta"][i]),
                                           (f >> 1) & 1, (f >> 2) & 1, hbd)
                want_l[i] = li[0], li[1]
                il, ia = li[0], li[1]
                assert tuple(gl[i]) == (il, ia), (ts, i)
                assert np.array_equal(ge[i, 128 - il:129 + ia], want_e[i, 128 - il:129 + ia]), \
                    (ts, i, int(ec["mode"][i]), int(ec["x"][i]), int(ec["y"][i]), f)
            # predict from the device-built edges with the same (mode, position)
            modes = np.where(ec["mode"] < 0, 0, ec["mode"]).astype(np.int32)
            valid = ec["mode"] >= 0
            var = np.where((ec["x"] == 0) & (ec["y"] == 0), 0,
                           np.where(ec["y"] == 0, 1, np.where(ec["x"] == 0, 2, 3)))
            pm = modes.copy()
            pa = np.where(pm == 12, 1, 0)
            pm = np.where(pa & (var == 0), 0, np.where(pa & (var == 2), 1,
                          np.where(pa & (var == 1), 2, pm)))
            alpha = rng.integers(-16, 17, n)
            pm = np.where((pm == 13) & (alpha == 0), 0, pm)
            base_angle = np.array([0, 90, 180, 45, 135, 113, 157, 203, 67] + [0] * 5)[pm]
            angle = np.where(pm == 13, alpha, base_angle + 3 * ec["angle_delta"])
            ief = np.where(ec["flags"] & 1, rng.integers(1, 3, n), 0)
            aw = np.minimum(w, hp.width - (tx0 + ec["x"])).clip(1, 64)
            ah = np.minimum(h, hp.height - (ty0 + ec["y"])).clip(1, 64)
            ic = _intra_cands(pm, var, angle, ief, aw, ah)
            ac = rng.integers(-2000, 2000, (n, w * h)).astype(np.int16)
            got = ctx.predict_intra_batch(ts, ic, edges, lens, bd, ac=_t(ac)).cpu().numpy().view(dt)
            for i in np.nonzero(valid)[0]:
                out = np.zeros((h, w), dt)
                assert oracle.r1o_dispatch_predict_intra(
                    int(pm[i]), int(var[i]), O.ptr(out), w, ts, bd, O.ptr(ac[i]), int(angle[i]),
                    int(ief[i]), O.ptr(want_e[i]), int(want_l[i, 0]), int(want_l[i, 1]), int(aw[i]),
                    int(ah[i]), hbd) == 0
                assert np.array_equal(got[i], out), (ts, i, int(pm[i]), int(angle[i]), int(ief[i]))


@pytest.mark.parametrize("bd", [8, 10])
def test_cfl_ac_vs_oracle(ctx, oracle, bd):
    from rav1e_amd.api import CFL_AC_CAND
    rng = np.random.default_rng(77 + bd)
    hp = O.HostPlane(256, 128, bd, rng=rng)
    dp = dev_plane(hp)
    hbd = int(bd > 8)
    for (bw, bh) in ((4, 4), (8, 8), (16, 16), (32, 32), (8, 16), (16, 8), (4, 16), (32, 8)):
        for (xdec, ydec) in ((1, 1), (1, 0), (0, 0)):
            n = 50
            c = np.zeros(n, CFL_AC_CAND)
            c["x"] = rng.integers(0, (hp.width - (bw << xdec)) // 8 + 1, n) * 8
            c["y"] = rng.integers(0, (hp.height - (bh << ydec)) // 8 + 1, n) * 8
            c["w_pad"] = rng.integers(0, max(1, bw // 4), n)
            c["h_pad"] = rng.integers(0, max(1, bh // 4), n)
            got = ctx.cfl_ac_batch(dp, bw, bh, xdec, ydec, c).cpu().numpy()
            for i in range(n):
                want = np.zeros(bw * bh, np.int16)
                oracle.r1o_pred_cfl_ac(O.ptr(want), hp.block_ptr(int(c["x"][i]), int(c["y"][i])),
                                       hp.stride, bw, bh, int(c["w_pad"][i]), int(c["h_pad"][i]),
                                       xdec, ydec, hbd)
                assert np.array_equal(got[i], want), (bw, bh, xdec, ydec, i)


# ------------------------------------------------------------------------ CDEF
def _plane_from(arr, bd, pad=16):
    from rav1e_amd.api import Plane
    h, w = arr.shape
    hp = O.HostPlane(w, h, bd, pad, pad, rng=np.random.default_rng(1))   # garbage in the padding
    hp.view()[:] = arr
    return hp, Plane.from_numpy(hp.data, w, h, bd, pad, pad)


@pytest.mark.parametrize("fixture,ncases", [("cdef_ref", 12), ("cdef_golden", 10)])
def test_cdef_frames_spec_model(ctx, fixture, ncases):
    """whole-frame CDEF (3 planes, 4:2:0 / 4:2:2 / 4:4:4, 8/10/12-bit), plus the direction
    maps.  cdef_ref.npz: frames filtered by the reference's own source text
    (gen_cdef_ref.py executes src/cdef.rs up to cdef_filter_tile); cdef_golden.npz: the
    independent spec-formulation model."""
    from rav1e_amd.api import CDEF_DIR_CAND
    Z = np.load(os.path.join(GOLD, fixture + ".npz"))
    G = {k: Z[k] for k in Z.files}
    for c in range(ncases):
        k = "c%d" % c
        W, H, xdec, ydec, bd, damping = (int(v) for v in G[k + "_meta"])
        dt = np.uint16 if bd > 8 else np.uint8
        planes_ = [_plane_from(G[k + "_in%d" % p].astype(dt), bd) for p in range(3)]
        skip, ci = _t(G[k + "_skip"]), _t(G[k + "_ci"])
        for p in range(3):
            xd, yd = (0, 0) if p == 0 else (xdec, ydec)
            _, dst = _plane_from(np.zeros_like(G[k + "_in%d" % p]).astype(dt), bd)
            ctx.cdef_filter_frame_plane(planes_[0][1], planes_[p][1], dst, p, xd, yd, W, H, skip, ci,
                                        G[k + "_ystr"], G[k + "_uvstr"], damping, bd)
            got = dst.data.cpu().numpy().view(dt)[16:16 + (H >> yd), dst.xorigin:dst.xorigin + (W >> xd)]
            assert np.array_equal(got.astype(np.uint16), G[k + "_out%d" % p]), (c, p)
        # direction search as its own batch
        nby, nbx = H // 8, W // 8
        dc = np.zeros(nby * nbx, CDEF_DIR_CAND)
        dc["x"] = np.tile(np.arange(nbx) * 8, nby)
        dc["y"] = np.repeat(np.arange(nby) * 8, nbx)
        d, v = ctx.cdef_find_dir_batch(planes_[0][1], dc)
        d, v = d.cpu().numpy().reshape(nby, nbx), v.cpu().numpy().reshape(nby, nbx)
        ns = ~np.array([[G[k + "_skip"][2 * by:2 * by + 2, 2 * bx:2 * bx + 2].all()
                         for bx in range(nbx)] for by in range(nby)])
        assert np.array_equal(d[ns], G[k + "_dir"][ns]) and np.array_equal(v[ns], G[k + "_var"][ns])
        # the two-step form: one analysis of the frame (a thread per 8x8 block), three filters
        da, va = ctx.cdef_analyze_frame(planes_[0][1], W, H, skip.shape[1], skip.shape[0])
        da_h, va_h = da.cpu().numpy()[:nby, :nbx], va.cpu().numpy()[:nby, :nbx]
        assert np.array_equal(da_h[ns], G[k + "_dir"][ns]) and np.array_equal(va_h[ns], G[k + "_var"][ns])
        assert np.array_equal(da_h, d) and np.array_equal(va_h, v)      # skipped blocks too
        for p in range(3):
            xd, yd = (0, 0) if p == 0 else (xdec, ydec)
            _, dst = _plane_from(np.zeros_like(G[k + "_in%d" % p]).astype(dt), bd)
            ctx.cdef_filter_frame_plane_dirs(da, va, planes_[p][1], dst, p, xd, yd, W, H, skip, ci,
                                             G[k + "_ystr"], G[k + "_uvstr"], damping, bd)
            got = dst.data.cpu().numpy().view(dt)[16:16 + (H >> yd), dst.xorigin:dst.xorigin + (W >> xd)]
            assert np.array_equal(got.astype(np.uint16), G[k + "_out%d" % p]), (c, p, "dirs")


@pytest.mark.parametrize("bd", [8, 10])
def test_cdef_filter_block_vs_oracle(ctx, oracle, bd):
    """cdef_filter_block with every edge-flag combination, direction and tap parity."""
    from rav1e_amd.api import CDEF_BLOCK_CAND
    rng = np.random.default_rng(60 + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    img = np.clip(np.cumsum(rng.integers(-9, 10, (96, 128)), axis=1) + (1 << (bd - 1)), 0,
                  (1 << bd) - 1).astype(dt)
    hp, dp = _plane_from(img, bd)
    for (xdec, ydec) in ((0, 0), (1, 1), (1, 0)):
        xs, ys = 8 >> xdec, 8 >> ydec
        n = min(256, (128 // xs - 2) * (96 // ys - 2))
        c = np.zeros(n, CDEF_BLOCK_CAND)
        # distinct destination blocks so the comparison is well defined
        pos = rng.permutation((128 // xs - 2) * (96 // ys - 2))[:n]
        c["x"] = (pos % (128 // xs - 2) + 1) * xs
        c["y"] = (pos // (128 // xs - 2) + 1) * ys
        c["pri_strength"] = rng.integers(0, 16, n) << (bd - 8)
        c["sec_strength"] = rng.choice([0, 1, 2, 4], n) << (bd - 8)
        c["dir"] = rng.integers(0, 8, n)
        c["damping"] = rng.integers(3, 7, n) + (bd - 8)
        c["edges"] = np.arange(n) % 16
        out_h, out_d = _plane_from(np.zeros_like(img), bd)
        ctx.cdef_filter_block_batch(dp, out_d, xdec, ydec, c)
        got = out_d.data.cpu().numpy().view(dt)[16:16 + 96, out_d.xorigin:out_d.xorigin + 128]
        for i in range(n):
            want = np.zeros((ys, xs), dt)
            x, y = int(c["x"][i]), int(c["y"][i])
            oracle.r1o_cdef_filter_block(O.ptr(want), xs, hp.block_ptr(x, y), hp.stride,
                                         int(c["pri_strength"][i]), int(c["sec_strength"][i]),
                                         int(c["dir"][i]), int(c["damping"][i]), bd, xdec, ydec,
                                         int(c["edges"][i]), int(bd > 8))
            assert np.array_equal(got[y:y + ys, x:x + xs], want), (bd, xdec, ydec, i)


# ------------------------------------------- compat shims of the widened rows
def test_compat_shims_inverse_cdef_ipred(ctx, oracle):
    """Per-call shims with the reference's asm-style signatures (host pointers):
    inverse transform, CDEF direction / filter on the padded u16 tile, intra
    prediction from the top-left pointer of the edge buffer."""
    import ctypes as C
    from rav1e_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(4)
    # inverse_transform_add, 8-bit and 10-bit, a few sizes / types
    for ts, tt in ((1, 0), (2, 3), (9, 0), (4, 0)):
        w, h = TX_SIZES[ts]
        area = min(w, 32) * min(h, 32)
        for bd, ct, pt in ((8, np.int16, np.uint8), (10, np.int32, np.uint16)):
            co = (rng.integers(-2000, 2001, area) * (rng.random(area) < 0.2)).astype(ct)
            stride = w + 24                      # a plane row, bytes below
            dst = rng.integers(0, 1 << bd, (h, stride)).astype(pt)
            want = np.ascontiguousarray(dst[:, :w])
            assert oracle.r1o_inverse_transform_add(O.ptr(co), O.ptr(want), w, ts, tt, bd,
                                                    int(bd > 8), int(bd > 8)) == 0
            if bd == 8:
                rc = L.rav1e_inv_txfm_add_hip(O.ptr(dst), stride, O.ptr(co), area, ts, tt)
            else:
                rc = L.rav1e_inv_txfm_add_hbd_hip(O.ptr(dst), stride * 2, O.ptr(co), area, 1023, ts, tt)
            assert rc == 0 and np.array_equal(dst[:, :w], want), (ts, tt, bd)
    # cdef direction
    for bd, pt in ((8, np.uint8), (10, np.uint16)):
        img = rng.integers(0, 1 << bd, (8, 24)).astype(pt)
        v1, v2 = C.c_uint32(), C.c_uint32()
        want = oracle.r1o_cdef_find_dir(O.ptr(img), 24, C.byref(v1), bd - 8, int(bd > 8))
        got = (L.rav1e_cdef_dir_hip(O.ptr(img), 24, C.byref(v2)) if bd == 8 else
               L.rav1e_cdef_dir_hbd_hip(O.ptr(img), 48, C.byref(v2), 1023))
        assert (got, v2.value) == (want, v1.value)
    # cdef filter on the reference's padded u16 tile (sentinel = 0x8000)
    for (xdec, ydec) in ((0, 0), (1, 1)):
        xs, ys = 8 >> xdec, 8 >> ydec
        for bd, pt in ((8, np.uint8), (10, np.uint16)):
            tmp = rng.integers(0, 1 << bd, (12, 12)).astype(np.uint16)
            tmp[:2, :] = 0x8000                  # no top
            tmp[:, xs + 2:] = 0x8000             # no right
            src = tmp[2:2 + ys, 2:2 + xs].astype(pt)
            # oracle on the equivalent (plane, edges) formulation
            plane = np.zeros((16, 16), pt)
            plane[2:14, 2:14] = np.where(tmp == 0x8000, 0, tmp).astype(pt)
            want = np.zeros((ys, xs), pt)
            edges = 1 | 8                         # LEFT | BOTTOM
            oracle.r1o_cdef_filter_block(O.ptr(want), xs, C.c_void_p(plane.ctypes.data + (4 * 16 + 4) * plane.itemsize),
                                         16, 5 << (bd - 8), 2 << (bd - 8), 3, 5 + (bd - 8), bd, xdec,
                                         ydec, edges, int(bd > 8))
            got = np.zeros((ys, xs + 3), pt)
            tp = C.c_void_p(tmp.ctypes.data + (2 * 12 + 2) * 2)
            if bd == 8:
                L.rav1e_cdef_filter_hip(O.ptr(got), xs + 3, tp, 24, 5, 2, 3, 5, xdec, ydec)
            else:
                L.rav1e_cdef_filter_hbd_hip(O.ptr(got), (xs + 3) * 2, tp, 24, 5 << 2, 2 << 2, 3, 7, 1023,
                                            xdec, ydec)
            assert np.array_equal(got[:, :xs], want), (xdec, ydec, bd)
            del src
    # intra prediction from the top-left pointer
    from test_oracle_predict import KAT, KAT_EDGE
    for mode, variant, angle, want in KAT:
        out = np.zeros((4, 4), np.uint8)
        tl = C.c_void_p(KAT_EDGE.ctypes.data + 128)
        assert L.rav1e_ipred_hip(O.ptr(out), 4, tl, 4, 4, angle, mode, variant, 0, 4, 4, 4, 4, None, 8) == 0
        assert out.ravel().tolist() == want, (mode, variant)


# --------------------------------------------------- lookahead cost maps (N1)
@pytest.mark.parametrize("bd", [8, 10])
def test_lookahead_cost_maps(ctx, oracle, bd):
    """estimate_intra_costs / estimate_inter_costs / importance block difference
    for a whole frame vs the oracle's composition of get_intra_edges ->
    DC_PRED -> get_satd (src/api/lookahead.rs:30-268)."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(31 + bd)
    a = O.HostPlane(328, 184, bd, rng=rng)        # not multiples of 64
    b = O.HostPlane(328, 184, bd, rng=rng)
    yy, xx = np.mgrid[0:184, 0:328]
    a.view()[:] = np.clip((np.sin(xx / 9.0) + np.cos(yy / 5.0)) * 60 * (1 << (bd - 8)) +
                          (1 << (bd - 1)) + rng.integers(-8, 9, (184, 328)), 0, (1 << bd) - 1)
    da, db = dev_plane(a), dev_plane(b)
    hb, wb = 184 // 8, 328 // 8
    pa, pb = a.cstruct(), b.cstruct()
    want = np.zeros(hb * wb, np.uint32)
    oracle.r1o_estimate_intra_costs(C.byref(pa), bd, O.ptr(want))
    got = ctx.estimate_intra_costs(da).cpu().numpy().view(np.uint32).ravel()
    assert np.array_equal(got, want)
    mvs = rng.integers(-300, 301, (hb, wb, 2)).astype(np.int16)
    want = np.zeros(hb * wb, np.uint32)
    oracle.r1o_estimate_inter_costs(C.byref(pa), C.byref(pb), O.ptr(mvs), O.ptr(want))
    got = ctx.estimate_inter_costs(da, db, torch.from_numpy(mvs).cuda()).cpu().numpy().view(np.uint32)
    assert np.array_equal(got.ravel(), want)
    tot = oracle.r1o_importance_block_difference(C.byref(pa), C.byref(pb))
    assert ctx.importance_block_difference(da, db) == tot / (hb * wb)
    # every block is predicted with pred_dc_128 (the reference's tile rectangle starts at the block,
    # lookahead.rs:84-89): a flat frame costs the same everywhere, |64 * (77 - 128)| / 8 per block
    flat = O.HostPlane(64, 64, bd, fill=77 << (bd - 8))
    c = ctx.estimate_intra_costs(dev_plane(flat)).cpu().numpy()
    assert (c == ((64 * 51 << (bd - 8)) + 4) >> 3).all()


# ------------------------ N4 (first step): quantize + tx-domain distortion + rate
@pytest.mark.parametrize("bd", [8, 10])
def test_quantize_rdo_vs_oracle(ctx, oracle, bd):
    """r1_quantize_rdo_batch: eob / qcoeffs / rcoeffs as r1_quantize_batch, plus
    the transform-domain distortion of encode_tx_block and estimate_rate."""
    rng = np.random.default_rng(800 + bd)
    ct = np.int16 if bd == 8 else np.int32
    for ts in (0, 1, 2, 3, 4, 5, 9, 11, 17, 18):
        w, h = TX_SIZES[ts]
        area, full = min(w, 32) * min(h, 32), w * h
        for qi in (40, 130, 255):
            n = 77
            acq = oracle.r1o_ac_q(qi, 0, bd)
            co = np.clip(rng.integers(-4 * acq, 4 * acq + 1, (n, full)) *
                         (rng.random((n, full)) < 0.3), np.iinfo(ct).min, np.iinfo(ct).max).astype(ct)
            if ct == np.int16:
                co[0, :8] = [32767, -32768, 30000, -30000, 1, -1, 0, 5]   # i32 wrap in c*c
            q = np.zeros((n, area), ct)
            r = np.zeros((n, area), ct)
            eobs = np.zeros(n, np.uint16)
            dist = np.zeros(n, np.uint64)
            rate = np.zeros(n, np.uint64)
            assert oracle.r1o_quantize_rdo_batch(O.ptr(co), full, n, ts, 0, qi, bd, 0, 0, 0, co.itemsize,
                                                 O.ptr(q), O.ptr(eobs), O.ptr(r), O.ptr(dist),
                                                 O.ptr(rate)) == 0
            o = ctx.quantize_rdo_batch(_t(co), ts, 0, qi, bd, 0)
            assert np.array_equal(o["eobs"].cpu().numpy().view(np.uint16), eobs), (bd, ts, qi)
            assert np.array_equal(o["qcoeffs"].cpu().numpy(), q), (bd, ts, qi)
            assert np.array_equal(o["rcoeffs"].cpu().numpy(), r), (bd, ts, qi)
            assert np.array_equal(o["tx_dist"].cpu().numpy().view(np.uint64), dist), (bd, ts, qi)
            assert np.array_equal(o["est_rate"].cpu().numpy().view(np.uint64), rate), (bd, ts, qi)


# ------------------------ N4: the candidate carried through the quantizer in one launch
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_rdo_full_cand_vs_oracle(ctx, oracle, bd):
    """r1_rdo_full_cand_batch = put_8tap -> diff -> forward_transform -> quantize ->
    dequantize -> tx-domain distortion -> estimate_rate (encode_tx_block,
    RDOType::TxDistEstRate), per-candidate tx_type and scan order, every tx size."""
    import ctypes as C
    a, b = planes(bd, seed=60 + bd, pads=(88, 120))
    # a second reference close to the source: small residuals, short eobs
    near = planes(bd, seed=60 + bd, pads=(88, 120))[0]
    nz = np.random.default_rng(5).integers(-3, 4, near.data.shape)
    dtp = near.data.dtype
    near.data[...] = np.clip(near.data.astype(np.int64) + nz, 0, (1 << bd) - 1).astype(dtp)
    da, db, dn = dev_plane(a), dev_plane(b), dev_plane(near)
    rng = np.random.default_rng(900 + bd)
    ct = np.int16 if bd == 8 else np.int32
    for ts, (w, h) in enumerate(TX_SIZES):
        carea = min(w, 32) * min(h, 32)
        for hp, dp, qi, intra in ((b, db, 35, 0), (b, db, 200, 1), (near, dn, 20, 0), (near, dn, 110, 0)):
            n = 37 if w * h <= 1024 else 9
            c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 50, ts)
            if hp is near:   # co-located, integer or fractional: residual = interpolation error
                c["rx"], c["ry"] = c["ox"], c["oy"]
            pa, pb = a.cstruct(), hp.cstruct()
            wsad, wsatd = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
            weob = np.zeros(n, np.uint16)
            wdist, wrate = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
            wq = np.zeros((n, carea), ct)
            assert oracle.r1o_rdo_full_cand_batch(
                C.byref(pa), C.byref(pb), w, h, ts, O.ptr(c), n, qi, intra, 0, 0, O.ptr(wsad),
                O.ptr(wsatd), O.ptr(weob), O.ptr(wdist), O.ptr(wrate), O.ptr(wq)) == 0
            o = ctx.rdo_full_cand_batch(da, dp, w, h, c, qi, is_intra=intra, want_qcoeffs=True)
            key = (bd, w, h, qi)
            assert np.array_equal(o["sad"].cpu().numpy().view(np.uint32), wsad), key
            assert np.array_equal(o["satd"].cpu().numpy().view(np.uint32), wsatd), key
            assert np.array_equal(o["qcoeffs"].cpu().numpy(), wq), key
            assert np.array_equal(o["eob"].cpu().numpy().view(np.uint16), weob), key
            assert np.array_equal(o["tx_dist"].cpu().numpy().view(np.uint64), wdist), key
            assert np.array_equal(o["est_rate"].cpu().numpy().view(np.uint64), wrate), key
            # scalars only (what the mode decision consumes)
            o2 = ctx.rdo_full_cand_batch(da, dp, w, h, c, qi, is_intra=intra, want_sad=False,
                                         want_satd=False)
            assert np.array_equal(o2["tx_dist"].cpu().numpy().view(np.uint64), wdist), key
            assert np.array_equal(o2["eob"].cpu().numpy().view(np.uint16), weob), key


def test_rdo_full_cand_equals_staged_chain(ctx):
    """Size-independent property at frame scale: the fused launch equals the staged
    chain r1_rdo_cand_batch -> r1_quantize_rdo_batch on the GPU, per tx type."""
    import torch
    a, b = planes(8, w=640, h=384, seed=77)
    da, db = dev_plane(a), dev_plane(b)
    rng = np.random.default_rng(78)
    for ts in (1, 2, 3, 4, 9, 17):
        w, h = TX_SIZES[ts]
        n = 2000
        c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 40, ts)
        full = ctx.rdo_full_cand_batch(da, db, w, h, c, 90, want_qcoeffs=True, want_coeffs=True)
        st = ctx.rdo_cand_batch(da, db, w, h, c)
        assert torch.equal(full["coeffs"], st["coeffs"])
        assert torch.equal(full["satd"], st["satd"])
        for tt in np.unique(c["tx_type"]):
            idx = torch.from_numpy(np.nonzero(c["tx_type"] == tt)[0]).cuda()
            q = ctx.quantize_rdo_batch(st["coeffs"][idx].contiguous(), ts, int(tt), 90, 8, 0)
            assert torch.equal(full["qcoeffs"][idx], q["qcoeffs"]), (ts, tt)
            assert torch.equal(full["eob"][idx], q["eobs"]), (ts, tt)
            assert torch.equal(full["tx_dist"][idx], q["tx_dist"]), (ts, tt)
            assert torch.equal(full["est_rate"][idx], q["est_rate"]), (ts, tt)


# ------------------------ N2: hierarchical motion estimation of whole tiles
def _me_dev_pyr(pyr):
    return [dev_plane(p) for p in pyr]


def _me_stats_tensor(a):
    import torch
    return torch.from_numpy(a.view(np.int32).reshape(a.shape[0], a.shape[1], 2).copy()).cuda()


def _me_stats_numpy(t):
    return t.cpu().numpy().reshape(t.shape[0], -1).view(O.ME_STATS).reshape(t.shape[0], t.shape[1])


def _me_images(kind, w, h, bd, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 1 << bd, (h, w)), rng.integers(0, 1 << bd, (h, w))
    # smooth texture, the reference a shifted + noisy copy: real motion, ties in flat areas
    f = rng.standard_normal((h + 64, w + 64))
    for _ in range(3):
        f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
    f = ((f - f.min()) / (f.max() - f.min()) * ((1 << bd) - 1)).astype(np.int64)
    if kind == "flat":
        f = (f >> (bd - 3)) << (bd - 3)      # 8 grey levels: many equal costs
    org = f[32:32 + h, 32:32 + w]
    ref = f[32 + 5:32 + 5 + h, 32 - 9:32 - 9 + w] + rng.integers(-2, 3, (h, w))
    return org, np.clip(ref, 0, (1 << bd) - 1)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("kind", ["noise", "smooth", "flat"])
def test_estimate_tile_motion_vs_oracle(ctx, oracle, bd, kind):
    """r1_estimate_tile_motion_batch against oracle/me.c: frame sizes that are not
    multiples of 64 (cropped superblocks and blocks), previous-frame predictors,
    8- and 10-bit, with and without the full-search stage."""
    from rav1e_amd.api import me_lambdas
    for (w, h, full) in ((200, 136, 0), (320, 192, 0), (136, 72, 1)):
        org, ref = _me_images(kind, w, h, bd, 7 * bd + w)
        po, pr = O.me_pyramid(org, bd), O.me_pyramid(ref, bd)
        rng = np.random.default_rng(w)
        prev = np.zeros((h // 4, w // 4), O.ME_STATS)
        prev["row"] = rng.integers(-80, 81, prev.shape)
        prev["col"] = rng.integers(-80, 81, prev.shape)
        prev["normalized_sad"] = rng.integers(0, 1 << 22, prev.shape)
        init = np.zeros_like(prev)           # what the previous frame left in the array
        init["row"] = rng.integers(-40, 41, init.shape)
        init["col"] = rng.integers(-40, 41, init.shape)
        init["normalized_sad"] = rng.integers(0, 1 << 22, init.shape)
        lam = me_lambdas(30.0)
        want = init.copy()
        O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, want, prev,
                    allow_full_search=full)
        st = _me_stats_tensor(init)
        ctx.estimate_tile_motion([dict(org=_me_dev_pyr(po), ref=_me_dev_pyr(pr), stats=st,
                                       prev=_me_stats_tensor(prev), tile=(0, 0, w, h))],
                                 w // 4, h // 4, bd, lam, allow_full_search=bool(full))
        got = _me_stats_numpy(st)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (bd, kind, w, h, bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])


def test_estimate_tile_motion_jobs_are_tiles_and_references(ctx, oracle):
    """several jobs in one call: two tiles of one frame (sharing the stats array)
    and a second reference frame with its own array; no previous-frame stats."""
    from rav1e_amd.api import me_lambdas
    w, h, bd = 384, 200, 8
    org, ref1 = _me_images("smooth", w, h, bd, 11)
    _, ref2 = _me_images("noise", w, h, bd, 12)
    po, p1, p2 = O.me_pyramid(org, bd), O.me_pyramid(ref1, bd), O.me_pyramid(ref2, bd)
    lam = me_lambdas(12.0)
    tiles = [(0, 0, 192, h), (192, 0, 192, h)]
    want1 = np.zeros((h // 4, w // 4), O.ME_STATS)
    want2 = np.zeros_like(want1)
    for t in tiles:
        O.me_oracle(oracle, po, p1, w // 4, h // 4, t, bd, lam, want1)
        O.me_oracle(oracle, po, p2, w // 4, h // 4, t, bd, lam, want2, allow_hp=1)
    s1, s2 = _me_stats_tensor(np.zeros_like(want1)), _me_stats_tensor(np.zeros_like(want1))
    do, d1, d2 = _me_dev_pyr(po), _me_dev_pyr(p1), _me_dev_pyr(p2)
    jobs = [dict(org=do, ref=d1, stats=s1, tile=t) for t in tiles] + \
           [dict(org=do, ref=d2, stats=s2, tile=t) for t in tiles]
    ctx.estimate_tile_motion(jobs, w // 4, h // 4, bd, lam)
    assert np.array_equal(_me_stats_numpy(s1), want1)
    assert np.array_equal(_me_stats_numpy(s2), want2)


def test_estimate_tile_motion_from_concurrent_threads(ctx, oracle):
    """one context shared by several host threads, each enqueueing on its own stream (rav1e's
    per-tile workers): the calls use more ring slots than the context has, results stay exact"""
    import threading
    import torch
    from rav1e_amd.api import me_lambdas
    w, h, bd = 256, 136, 8
    lam = me_lambdas(20.0)
    cases = []
    for k in range(3):
        org, ref = _me_images("smooth", w, h, bd, 50 + k)
        po, pr = O.me_pyramid(org, bd), O.me_pyramid(ref, bd)
        want = np.zeros((h // 4, w // 4), O.ME_STATS)
        O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, want)
        cases.append((_me_dev_pyr(po), _me_dev_pyr(pr), want))
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            do, dr, want = cases[k]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(6):
                    st = torch.zeros((h // 4, w // 4, 2), dtype=torch.int32, device="cuda")
                    ctx.estimate_tile_motion([dict(org=do, ref=dr, stats=st, tile=(0, 0, w, h))],
                                             w // 4, h // 4, bd, lam)
                    stream.synchronize()
                    if not np.array_equal(_me_stats_numpy(st), want):
                        errors.append(k)
        except Exception as e:                     # noqa: BLE001 -- reported below
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("bd", [8, 10])
def test_estimate_motion_blocks_subpel_vs_oracle(ctx, oracle, bd):
    """r1_estimate_motion_batch (the RDO-time estimate_motion with pmv: full-pel from the
    tile's MEStats, SATD re-cost, sub-pel diamond through put_8tap) for every BlockSize up to
    64x64, SATD and SAD flavours, 1/8- and 1/4-pel precision, all four filters."""
    from rav1e_amd.api import me_lambdas, ME_RESULT
    w, h = 320, 192
    org, ref = _me_images("smooth", w, h, bd, 21 + bd)
    ref = np.clip(ref + np.random.default_rng(5).integers(-6, 7, ref.shape), 0, (1 << bd) - 1)
    po, pr = O.me_pyramid(org, bd), O.me_pyramid(ref, bd)
    lam = me_lambdas(25.0)
    stats = np.zeros((h // 4, w // 4), O.ME_STATS)
    O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, stats)   # realistic MEStats
    rng = np.random.default_rng(31 + bd)
    prev = np.zeros_like(stats)
    prev["row"] = rng.integers(-60, 61, prev.shape)
    prev["col"] = rng.integers(-60, 61, prev.shape)
    prev["normalized_sad"] = rng.integers(0, 1 << 20, prev.shape)
    sizes = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32),
             (32, 64), (64, 32), (64, 64), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
    c = np.zeros(6 * len(sizes), O.ME_BLOCK_CAND)
    for i in range(len(c)):
        bw, bh = sizes[i % len(sizes)]
        c["w"][i], c["h"][i] = bw, bh
        c["bx"][i] = rng.integers(0, (w - bw) // 4 + 1)
        c["by"][i] = rng.integers(0, (h - bh) // 4 + 1)
        c["corner"][i] = rng.choice([0, 1, 3, 5, 7])
        c["pmv"][i] = rng.integers(-40, 41, (2, 2))
    job = dict(org=_me_dev_pyr(po), ref=_me_dev_pyr(pr), stats=_me_stats_tensor(stats),
               prev=_me_stats_tensor(prev), tile=(0, 0, w, h))
    for use_satd, mode, hp in ((1, 0, 1), (0, 0, 1), (1, 2, 0), (1, 1, 1), (0, 3, 0)):
        want = O.me_block_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, stats, prev, c,
                                 use_satd=use_satd, filter_mode=mode, allow_hp=hp)
        got = ctx.estimate_motion_batch(job, c, w // 4, h // 4, bd, lam, use_satd=bool(use_satd),
                                        filter_mode=mode, allow_hp=bool(hp)).cpu().numpy().view(ME_RESULT)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (bd, use_satd, mode, hp, c[bad[0]], got[bad[0]], want[bad[0]])
    # a launch sized for 16x16 refuses the larger blocks (empty result), serves the rest
    got = ctx.estimate_motion_batch(job, c, w // 4, h // 4, bd, lam, max_w=16, max_h=16).cpu().numpy().view(ME_RESULT)
    big = (c["w"] > 16) | (c["h"] > 16)
    assert (got["cost"][big] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    want = O.me_block_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, stats, prev, c)
    assert np.array_equal(got[~big], want[~big])
    # ... through the wave-per-block kernel: the same flavour sweep on a batch of small blocks
    cs = np.tile(c[~big], 5)
    cs["bx"] = rng.integers(0, (w - 16) // 4 + 1, len(cs))
    cs["by"] = rng.integers(0, (h - 16) // 4 + 1, len(cs))
    cs["pmv"] = rng.integers(-40, 41, (len(cs), 2, 2))
    for use_satd, mode, hp in ((1, 0, 1), (0, 0, 1), (1, 2, 0), (1, 1, 1), (0, 3, 0)):
        want = O.me_block_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, stats, prev, cs,
                                 use_satd=use_satd, filter_mode=mode, allow_hp=hp)
        got = ctx.estimate_motion_batch(job, cs, w // 4, h // 4, bd, lam, use_satd=bool(use_satd),
                                        filter_mode=mode, allow_hp=bool(hp), max_w=16,
                                        max_h=16).cpu().numpy().view(ME_RESULT)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (bd, use_satd, mode, hp, cs[bad[0]], got[bad[0]], want[bad[0]])


# ------------------------ N3: deblocking filter + level search
def _deblock_planes(rng, w, h, bd, xdec, ydec, blocks):
    """per-plane (rec, src) images: per-8x8 DC steps + noise (what a coarse quantizer leaves)"""
    out = []
    for pli in range(3):
        xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
        pw, ph = w >> xd, h >> yd
        yy, xx = np.mgrid[0:ph, 0:pw]
        step = rng.integers(-6 << (bd - 8), (6 << (bd - 8)) + 1, (h // 8 + 1, w // 8 + 1))
        rec = (1 << (bd - 1)) + 2 * step[(yy << yd) // 8, (xx << xd) // 8] + rng.integers(-2, 3, (ph, pw))
        rec = np.clip(rec, 0, (1 << bd) - 1)
        src = np.clip(rec + rng.integers(-3 << (bd - 8), (3 << (bd - 8)) + 1, rec.shape), 0, (1 << bd) - 1)
        out.append((rec, src))
    return out


@pytest.mark.parametrize("fixture", ["deblock_ref", "deblock_golden"])
def test_deblock_golden_frames(ctx, fixture):
    """deblock_ref.npz: frames filtered and tallies summed by the reference's own source text
    (gen_deblock_ref.py executes src/deblock.rs); deblock_golden.npz: the specification-model
    frames and brute-force tallies."""
    import torch
    G = dict(np.load(os.path.join(GOLD, fixture + ".npz")))
    for name in sorted(k[:-5] for k in G if k.endswith("_meta")):
        w, h, cw, ch, bd, xdec, ydec = [int(v) for v in G[name + "_meta"]]
        blocks = torch.from_numpy(np.ascontiguousarray(G[name + "_blocks"]).view(np.uint8).reshape(
            G[name + "_blocks"].shape + (8,))).cuda()
        state = np.ascontiguousarray(G[name + "_state"])
        dt = np.uint8 if bd == 8 else np.uint16
        for pli in range(3):
            xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
            hp = O.plane_from_image(G["%s_p%d_rec" % (name, pli)], bd, 16, 16)
            hs = O.plane_from_image(G["%s_p%d_src" % (name, pli)], bd, 16, 16)
            dp, ds = dev_plane(hp), dev_plane(hs)
            t = ctx.deblock_sse_plane(dp, ds, pli, xd, yd, blocks, cw, ch).cpu().numpy()
            assert np.array_equal(np.cumsum(t[0])[:64], G["%s_p%d_tv" % (name, pli)]), (name, pli)
            assert np.array_equal(np.cumsum(t[1])[:64], G["%s_p%d_th" % (name, pli)]), (name, pli)
            ctx.deblock_plane(state, dp, pli, xd, yd, blocks, cw, ch)
            got = dp.data.cpu().numpy().view(dt)[hp.yorigin:hp.yorigin + hp.height,
                                                  hp.xorigin:hp.xorigin + hp.width]
            assert np.array_equal(got, G["%s_p%d_out" % (name, pli)]), (name, pli)


@pytest.mark.parametrize("cfg", [(8, 1, 1, False), (10, 1, 1, True), (12, 0, 0, True), (8, 1, 0, False)])
def test_deblock_frame_vs_oracle(ctx, oracle, cfg):
    """whole frames (720p-class, cropped width / height), three planes, against oracle/deblock.c:
    the filter in place (padding untouched), the tallies and the picked levels."""
    import ctypes as C
    import torch
    import deblock_util as D
    bd, xdec, ydec, deltas = cfg
    w, h, cw, ch = 1280, 720, 1276, 714
    rng = np.random.default_rng(70 + bd + xdec)
    blocks = D.random_blocks(rng, w // 4, h // 4, xdec, ydec, deltas=deltas)
    state = D.make_state([int(v) for v in rng.integers(8, 50, 4)], rng, deltas, deltas)
    dblocks = torch.from_numpy(blocks.view(np.uint8).reshape(blocks.shape + (8,)).copy()).cuda()
    dt = np.uint8 if bd == 8 else np.uint16
    for pli, (rec, src) in enumerate(_deblock_planes(rng, w, h, bd, xdec, ydec, blocks)):
        xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
        hp, hs = O.plane_from_image(rec, bd, 24, 24), O.plane_from_image(src, bd, 24, 24)
        dp, ds = dev_plane(hp), dev_plane(hs)
        # level search on the unfiltered reconstruction
        tv, th = np.zeros(65, np.int64), np.zeros(65, np.int64)
        pc, sc = hp.cstruct(), hs.cstruct()
        assert oracle.r1o_deblock_sse_plane(C.byref(pc), C.byref(sc), pli, xd, yd, blocks.ctypes.data,
                                            blocks.shape[1], blocks.shape[1], blocks.shape[0], cw, ch, bd,
                                            tv.ctypes.data, th.ctypes.data) == 0
        t = ctx.deblock_sse_plane(dp, ds, pli, xd, yd, dblocks, cw, ch)
        assert np.array_equal(t.cpu().numpy(), np.stack([tv, th])), (cfg, pli)
        lv = np.zeros(2, np.uint8)
        oracle.r1o_deblock_pick_levels(tv.ctypes.data, th.ctypes.data, pli, lv.ctypes.data)
        assert np.array_equal(ctx.deblock_pick_levels(t, pli), lv[:2] if pli == 0 else lv[:1])
        # filter
        assert oracle.r1o_deblock_plane(state.ctypes.data, C.byref(pc), pli, xd, yd, blocks.ctypes.data,
                                        blocks.shape[1], blocks.shape[1], blocks.shape[0], cw, ch, bd) == 0
        ctx.deblock_plane(state, dp, pli, xd, yd, dblocks, cw, ch)
        got = dp.data.cpu().numpy().view(dt)
        bad = np.argwhere(got != hp.data)
        assert len(bad) == 0, (cfg, pli, bad[:4])
        assert (hp.view() != rec).sum() > rec.size // 20       # the filter did something


# ------------------------ N1: intra mode pre-screen in one launch
@pytest.mark.parametrize("bd", [8, 10])
def test_intra_prescreen_vs_oracle(ctx, oracle, bd):
    """r1_intra_satd_batch: 13 luma modes per block from one edge set (get_intra_edges with
    IntraParam::None, as src/rdo.rs:1442-1458), SATD against the source; oracle = the reference's
    composition get_intra_edges -> dispatch_predict_intra -> get_satd, call by call."""
    import torch
    from rav1e_amd.api import INTRA_EDGE_CAND
    rng = np.random.default_rng(640 + bd)
    rec = O.HostPlane(192, 128, bd, rng=rng)
    srcp = O.HostPlane(192, 128, bd, rng=rng)
    drec, dsrc = dev_plane(rec), dev_plane(srcp)
    hbd = int(bd > 8)
    dt = np.uint16 if hbd else np.uint8
    MODES = list(range(13))                      # RAV1E_INTRA_MODES: DC .. PAETH
    BASE = [0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0]
    for ts in (0, 1, 2, 3, 4, 5, 8, 13, 9, 17):
        w, h = TX_SIZES[ts]
        nb = 40
        gx, gy = rec.width // w, rec.height // h
        bxs, bys = rng.integers(0, gx, nb) * w, rng.integers(0, gy, nb) * h
        bxs[:3], bys[:3] = [0, w, 0], [0, 0, h]
        ec = np.zeros(nb, INTRA_EDGE_CAND)
        ec["x"], ec["y"] = bxs, bys
        ec["mode"] = -1                           # IntraParam::None, no mode-specific trimming
        ec["flags"] = 1 | (rng.integers(0, 4, nb) << 1)
        edges, lens = ctx.intra_edges_batch(drec, (0, 0, rec.width, rec.height), ts, ec)
        he, hl = edges.cpu().numpy().view(dt), lens.cpu().numpy()
        var = np.where((bxs == 0) & (bys == 0), 0, np.where(bys == 0, 1, np.where(bxs == 0, 2, 3)))
        pm = np.tile(MODES, nb)
        v13 = np.repeat(var, 13)
        # PAETH without both neighbours falls back (PredictionMode::predict_intra, predict.rs:116-140)
        pm = np.where((pm == 12) & (v13 == 0), 0, np.where((pm == 12) & (v13 == 2), 1,
                      np.where((pm == 12) & (v13 == 1), 2, pm)))
        angle = np.array(BASE)[pm]
        ief = np.where((pm >= 1) & (pm <= 8), np.tile(rng.integers(1, 3, 13), nb), 0)
        ic = _intra_cands(pm, v13, angle, ief, [w] * len(pm), [h] * len(pm))
        pos = torch.from_numpy(np.stack([bxs, bys], 1).astype(np.int16)).cuda()
        got = ctx.intra_satd_batch(dsrc, ts, ic, 13, pos, edges, lens).cpu().numpy().view(np.uint32)
        for b in range(nb):
            for k in range(13):
                i = b * 13 + k
                out = np.zeros((h, w), dt)
                assert oracle.r1o_dispatch_predict_intra(
                    int(pm[i]), int(v13[i]), O.ptr(out), w, ts, bd, None, int(angle[i]), int(ief[i]),
                    O.ptr(he[b]), int(hl[b, 0]), int(hl[b, 1]), w, h, hbd) == 0
                want = oracle.r1o_get_satd(srcp.block_ptr(int(bxs[b]), int(bys[b])), srcp.stride,
                                           O.ptr(out), w, w, h, hbd)
                assert got[i] == want, (bd, ts, b, k, int(pm[i]))


# ------------------------ pixel-domain leg of the full candidate (default-configuration RDO)
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_rdo_pixel_cand_vs_oracle(ctx, oracle, bd):
    """r1_rdo_pixel_cand_batch = put_8tap -> diff -> forward_transform -> quantize -> dequantize
    -> inverse_transform_add -> sse_wxh / cdef_dist_wxh with the DistortionScale grid: every tx
    size, per-candidate tx types, both distortion kinds, with and without a scale grid."""
    import ctypes as C
    a, b = planes(bd, seed=80 + bd, pads=(88, 120))
    near = planes(bd, seed=80 + bd, pads=(88, 120))[0]
    nz = np.random.default_rng(6).integers(-4, 5, near.data.shape)
    near.data[...] = np.clip(near.data.astype(np.int64) + nz, 0, (1 << bd) - 1).astype(near.data.dtype)
    da, db, dn = dev_plane(a), dev_plane(b), dev_plane(near)
    rng = np.random.default_rng(950 + bd)
    ct = np.int16 if bd == 8 else np.int32
    dt = np.uint8 if bd == 8 else np.uint16
    scales = rng.integers(1 << 12, 1 << 16, ((a.height + 7) // 8, (a.width + 7) // 8)).astype(np.uint32)
    dscales = _t(scales.view(np.int32))
    for ts, (w, h) in enumerate(TX_SIZES):
        carea = min(w, 32) * min(h, 32)
        for hp, dp, qi, kind, use_sc in ((b, db, 60, 3, True), (near, dn, 25, 3, False),
                                         (near, dn, 140, 2, True), (b, db, 220, 2, False)):
            n = 29 if w * h <= 1024 else 7
            c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 50, ts)
            if hp is near:
                c["rx"], c["ry"] = c["ox"], c["oy"]
            pa, pb = a.cstruct(), hp.cstruct()
            wsad, wsatd = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
            weob, wdist = np.zeros(n, np.uint16), np.zeros(n, np.uint64)
            wq, wrec = np.zeros((n, carea), ct), np.zeros((n, h, w), dt)
            assert oracle.r1o_rdo_pixel_cand_batch(
                C.byref(pa), C.byref(pb), w, h, ts, O.ptr(c), n, qi, 0, 0, 0, kind,
                O.ptr(scales) if use_sc else None, scales.shape[1], 0, 0, O.ptr(wsad), O.ptr(wsatd),
                O.ptr(weob), O.ptr(wdist), O.ptr(wq), O.ptr(wrec), None) == 0
            o = ctx.rdo_pixel_cand_batch(da, dp, w, h, c, qi, kind, scales=dscales if use_sc else None,
                                         want_qcoeffs=True, want_rec=True)
            key = (bd, w, h, qi, kind)
            assert np.array_equal(o["satd"].cpu().numpy().view(np.uint32), wsatd), key
            assert np.array_equal(o["qcoeffs"].cpu().numpy(), wq), key
            assert np.array_equal(o["eob"].cpu().numpy().view(np.uint16), weob), key
            assert np.array_equal(o["rec"].cpu().numpy().view(dt), wrec), key
            assert np.array_equal(o["dist"].cpu().numpy().view(np.uint64), wdist), key
            o2 = ctx.rdo_pixel_cand_batch(da, dp, w, h, c, qi, kind, scales=dscales if use_sc else None,
                                          want_sad=False, want_satd=False)
            assert np.array_equal(o2["dist"].cpu().numpy().view(np.uint64), wdist), key


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_quantizer_at_the_coefficient_range_limits(ctx, oracle, bd):
    """The fused kernels divide |c << log_tx_scale| by ac_q on 24-bit multipliers (csrc/quant_common.hpp, NARROW for
    i16 and -- round 6 -- MID for i32 coefficients), exact while the operand stays below 2^18 / 2^22.  The largest
    coefficients a pixel residual can produce: source at one end of the pixel range against a prediction at the other
    (flat: everything lands on DC), stripes and checkerboards of the two (energy on the highest AC basis), every
    transform size, the smallest and the largest quantizer (qindex 0: the largest levels; 255), every valid type --
    the pixel-domain chain (eob, qcoeffs, reconstruction, distortion) and the transform-domain chain (tx_dist) against
    the oracle's general 32 / 64-bit arithmetic."""
    import ctypes as C
    from rav1e_amd.types import valid_av1_transform
    W, H = 192, 128
    mx = (1 << bd) - 1
    yy, xx = np.mgrid[0:H, 0:W]
    pats = {"flat": np.full((H, W), mx), "rows": np.where(yy & 1, mx, 0), "cols": np.where(xx & 1, mx, 0),
            "checker": np.where((xx ^ yy) & 1, mx, 0), "blocks4": np.where(((xx >> 2) ^ (yy >> 2)) & 1, mx, 0)}
    ct = np.int16 if bd == 8 else np.int32
    dt = np.uint8 if bd == 8 else np.uint16
    rng = np.random.default_rng(12 + bd)
    for name, img in pats.items():
        for flip in (False, True):
            src_img = mx - img if flip else img
            a = O.plane_from_image(src_img, bd, 88, 88)
            b = O.plane_from_image(mx - src_img, bd, 88, 88)      # the prediction: every residual sample is +-max
            da, db = dev_plane(a), dev_plane(b)
            pa, pb = a.cstruct(), b.cstruct()
            for ts, (w, h) in enumerate(TX_SIZES):
                carea = min(w, 32) * min(h, 32)
                valid = [t for t in range(16) if valid_av1_transform(ts, t)]
                n = len(valid)
                c = np.zeros(n, O.RDO_CAND)
                c["ox"] = c["rx"] = rng.integers(0, (W - w) // 2 + 1, n) * 2      # even: the pattern's phase is kept
                c["oy"] = c["ry"] = rng.integers(0, (H - h) // 2 + 1, n) * 2
                c["tx_type"] = valid
                for qi in (0, 255):
                    weob, wdist = np.zeros(n, np.uint16), np.zeros(n, np.uint64)
                    wq, wrec = np.zeros((n, carea), ct), np.zeros((n, h, w), dt)
                    assert oracle.r1o_rdo_pixel_cand_batch(C.byref(pa), C.byref(pb), w, h, ts, O.ptr(c), n, qi, 0, 0, 0, 3, None, 0,
                                                           0, 0, None, None, O.ptr(weob), O.ptr(wdist), O.ptr(wq), O.ptr(wrec),
                                                           None) == 0
                    o = ctx.rdo_pixel_cand_batch(da, db, w, h, c, qi, 3, want_qcoeffs=True, want_rec=True, want_sad=False,
                                                 want_satd=False)
                    key = (bd, name, flip, w, h, qi)
                    assert np.array_equal(o["qcoeffs"].cpu().numpy(), wq), key
                    assert np.array_equal(o["eob"].cpu().numpy().view(np.uint16), weob), key
                    assert np.array_equal(o["rec"].cpu().numpy().view(dt), wrec), key
                    assert np.array_equal(o["dist"].cpu().numpy().view(np.uint64), wdist), key
                    feob, fdist = np.zeros(n, np.uint16), np.zeros(n, np.uint64)
                    assert oracle.r1o_rdo_full_cand_batch(C.byref(pa), C.byref(pb), w, h, ts, O.ptr(c), n, qi, 0, 0, 0, None, None,
                                                          O.ptr(feob), O.ptr(fdist), None, None) == 0
                    f = ctx.rdo_full_cand_batch(da, db, w, h, c, qi, want_sad=False, want_satd=False)
                    assert np.array_equal(f["eob"].cpu().numpy().view(np.uint16), feob), key
                    assert np.array_equal(f["tx_dist"].cpu().numpy().view(np.uint64), fdist), key
            if name == "flat":
                assert int(np.abs(wq).max()) > 0


@pytest.mark.parametrize("bd", [8, 10])
def test_rdo_pred_cand_vs_oracle(ctx, oracle, bd):
    """r1_rdo_pred_cand_batch: the chains with the prediction taken from a dense buffer (intra
    predictions / compound averages): transform-domain and both pixel-domain distortions."""
    import ctypes as C
    a, _ = planes(bd, seed=90 + bd)
    da = dev_plane(a)
    rng = np.random.default_rng(970 + bd)
    ct = np.int16 if bd == 8 else np.int32
    dt = np.uint8 if bd == 8 else np.uint16
    for ts in (0, 1, 2, 3, 4, 5, 8, 9, 13, 17):
        w, h = TX_SIZES[ts]
        carea = min(w, 32) * min(h, 32)
        n = 31 if w * h <= 1024 else 6
        c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 0, ts)
        # predictions near the source (what an intra / compound predictor delivers)
        pred = np.zeros((n, h, w), dt)
        for i in range(n):
            blk = a.view()[c["oy"][i]:c["oy"][i] + h, c["ox"][i]:c["ox"][i] + w].astype(np.int64)
            pred[i] = np.clip(blk + rng.integers(-20, 21, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1)
        dpred = _t(pred.view(np.int16) if bd > 8 else pred)
        for kind, qi in ((0, 70), (2, 30), (3, 150)):
            pa = a.cstruct()
            wsad, wsatd = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
            weob, wdist = np.zeros(n, np.uint16), np.zeros(n, np.uint64)
            wq, wrec = np.zeros((n, carea), ct), np.zeros((n, h, w), dt)
            assert oracle.r1o_rdo_pixel_cand_batch(
                C.byref(pa), None, w, h, ts, O.ptr(c), n, qi, 1, 0, 0, kind, None, 0, 0, 0, O.ptr(wsad),
                O.ptr(wsatd), O.ptr(weob), O.ptr(wdist), O.ptr(wq), O.ptr(wrec) if kind else None,
                O.ptr(pred)) == 0
            o = ctx.rdo_pixel_cand_batch(da, None, w, h, c, qi, kind, is_intra=1, want_qcoeffs=True,
                                         want_rec=bool(kind), pred=dpred)
            key = (bd, w, h, kind)
            assert np.array_equal(o["sad"].cpu().numpy().view(np.uint32), wsad), key
            assert np.array_equal(o["satd"].cpu().numpy().view(np.uint32), wsatd), key
            assert np.array_equal(o["eob"].cpu().numpy().view(np.uint16), weob), key
            assert np.array_equal(o["qcoeffs"].cpu().numpy(), wq), key
            assert np.array_equal(o["dist"].cpu().numpy().view(np.uint64), wdist), key
            if kind:
                assert np.array_equal(o["rec"].cpu().numpy().view(dt), wrec), key


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_rdo_txsearch_vs_oracle(ctx, oracle, bd):
    """r1_rdo_txsearch_batch (the transform-type fan-out of rdo_tx_type_decision, src/rdo.rs:1701-1817):
    ONE launch evaluates every type of the mask on one prediction per candidate.  Every tx size up to
    32x32, inter and intra masks (RAV1E_TX_TYPES cut by the tx set) and the full 16-type set, fractional
    motion vectors, candidate counts that leave the last wave ragged, the three distortion kinds, from
    the reference plane and from a dense prediction buffer; 64-point sides go to the plain kernel."""
    import ctypes as C
    a, b = planes(bd, seed=180 + bd, pads=(88, 120))
    near = planes(bd, seed=180 + bd, pads=(88, 120))[0]
    nz = np.random.default_rng(8).integers(-5, 6, near.data.shape)
    near.data[...] = np.clip(near.data.astype(np.int64) + nz, 0, (1 << bd) - 1).astype(near.data.dtype)
    da, db, dn = dev_plane(a), dev_plane(b), dev_plane(near)
    rng = np.random.default_rng(1950 + bd)
    ct = np.int16 if bd == 8 else np.int32
    dt = np.uint8 if bd == 8 else np.uint16
    scales = rng.integers(1 << 12, 1 << 16, ((a.height + 7) // 8, (a.width + 7) // 8)).astype(np.uint32)
    dscales = _t(scales.view(np.int32))
    for ts, (w, h) in enumerate(TX_SIZES):
        carea = min(w, 32) * min(h, 32)
        side64 = max(w, h) == 64
        m_intra, m_inter = ctx.tx_type_mask(ts, False), ctx.tx_type_mask(ts, True)
        m_all = ctx.tx_type_mask(ts, True, rav1e_types_only=False)
        assert m_intra == oracle.r1o_tx_type_mask(ts, 0, 0, 1) and m_all == oracle.r1o_tx_type_mask(ts, 1, 0, 0)
        for hp, dp, qi, kind, use_sc, mask, intra in ((near, dn, 25, 3, True, m_inter, 0), (b, db, 70, 2, True, m_intra, 1),
                                                      (near, dn, 140, 0, False, m_all, 0), (b, db, 200, 3, False, m_intra, 1),
                                                      (near, dn, 60, 2, False, 1 << 9 if not side64 else 1, 0)):
            nt = bin(mask).count("1")
            n = (29 if w * h <= 256 else 11) if not side64 else 5
            c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 50, ts)
            c["tx_type"] = 0 if side64 else rng.integers(0, 16, n)      # ignored below 64-point sides
            if hp is near:
                c["rx"], c["ry"] = c["ox"], c["oy"]
            pa, pb = a.cstruct(), hp.cstruct()
            wsad, wsatd = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
            weob, wdist, wrate = np.zeros((n, nt), np.uint16), np.zeros((n, nt), np.uint64), np.zeros((n, nt), np.uint64)
            wq, wrec = np.zeros((n, nt, carea), ct), np.zeros((n, nt, h, w), dt)
            assert oracle.r1o_rdo_txsearch_batch(
                C.byref(pa), C.byref(pb), None, w, h, ts, O.ptr(c), n, mask, qi, intra, 0, 0, kind,
                O.ptr(scales) if use_sc else None, scales.shape[1], 0, 0, O.ptr(wsad), O.ptr(wsatd), O.ptr(weob),
                O.ptr(wdist), O.ptr(wrate) if kind == 0 else None, O.ptr(wq), O.ptr(wrec) if kind else None) == 0
            o = ctx.rdo_txsearch_batch(da, dp, w, h, c, mask, qi, kind, scales=dscales if use_sc else None,
                                       is_intra=intra, want_sad=True, want_satd=True, want_est_rate=kind == 0,
                                       want_qcoeffs=True, want_rec=bool(kind))
            key = (bd, w, h, qi, kind, hex(mask))
            assert np.array_equal(o["sad"].cpu().numpy().view(np.uint32), wsad), key
            assert np.array_equal(o["satd"].cpu().numpy().view(np.uint32), wsatd), key
            assert np.array_equal(o["eob"].cpu().numpy().view(np.uint16), weob), key
            assert np.array_equal(o["qcoeffs"].cpu().numpy(), wq), key
            assert np.array_equal(o["dist"].cpu().numpy().view(np.uint64), wdist), key
            if kind:
                assert np.array_equal(o["rec"].cpu().numpy().view(dt), wrec), key
            else:
                assert np.array_equal(o["est_rate"].cpu().numpy().view(np.uint64), wrate), key
            # scalars only, as the search consumes them
            o2 = ctx.rdo_txsearch_batch(da, dp, w, h, c, mask, qi, kind, scales=dscales if use_sc else None,
                                        is_intra=intra)
            assert np.array_equal(o2["dist"].cpu().numpy().view(np.uint64), wdist), key
            assert np.array_equal(o2["eob"].cpu().numpy().view(np.uint16), weob), key
        if side64:
            continue
        # the prediction from a dense buffer (the intra case of the reference: do_rdo_tx_type needs !is_inter)
        n = 19
        c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 0, ts)
        pred = np.zeros((n, h, w), dt)
        for i in range(n):
            blk = a.view()[c["oy"][i]:c["oy"][i] + h, c["ox"][i]:c["ox"][i] + w].astype(np.int64)
            pred[i] = np.clip(blk + rng.integers(-12, 13, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1)
        dpred = _t(pred.view(np.int16) if bd > 8 else pred)
        nt = bin(m_intra).count("1")
        pa = a.cstruct()
        for kind, qi in ((3, 90), (0, 50)):
            weob, wdist = np.zeros((n, nt), np.uint16), np.zeros((n, nt), np.uint64)
            wq = np.zeros((n, nt, carea), ct)
            assert oracle.r1o_rdo_txsearch_batch(
                C.byref(pa), None, O.ptr(pred), w, h, ts, O.ptr(c), n, m_intra, qi, 1, 0, 0, kind, O.ptr(scales),
                scales.shape[1], 0, 0, None, None, O.ptr(weob), O.ptr(wdist), None, O.ptr(wq), None) == 0
            o = ctx.rdo_txsearch_batch(da, None, w, h, c, m_intra, qi, kind, scales=dscales, is_intra=1,
                                       want_qcoeffs=True, pred=dpred)
            key = (bd, w, h, kind, "pred")
            assert np.array_equal(o["eob"].cpu().numpy().view(np.uint16), weob), key
            assert np.array_equal(o["qcoeffs"].cpu().numpy(), wq), key
            assert np.array_equal(o["dist"].cpu().numpy().view(np.uint64), wdist), key


def test_rdo_txsearch_rejects_what_the_reference_cannot_code(ctx):
    """argument checks of r1_rdo_txsearch_batch: a type the size has no kernel for (ADST at 32x32), anything but DCT_DCT
    on a 64-point side, an empty mask, bits beyond the 16 TxTypes, both / neither prediction source"""
    import torch
    a, b = planes(8, seed=5)
    da, db = dev_plane(a), dev_plane(b)
    c = rand_rdo_cands(np.random.default_rng(1), 4, a.width, a.height, 32, 32, 0, 3)
    for (w, h, mask) in ((32, 32, 0x2), (32, 32, 0x203), (64, 64, 0x201), (16, 16, 0), (16, 16, 0x10001), (32, 8, 0x4)):
        cc = rand_rdo_cands(np.random.default_rng(2), 4, a.width, a.height, w, h, 0, 0)
        with pytest.raises(RuntimeError):
            ctx.rdo_txsearch_batch(da, db, w, h, cc, mask, 60, 3)
    pred = torch.zeros((4, 32, 32), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        ctx.rdo_txsearch_batch(da, db, 32, 32, c, 0x201, 60, 3, pred=pred)       # both sources
    with pytest.raises(RuntimeError):
        ctx.rdo_txsearch_batch(da, None, 32, 32, c, 0x201, 60, 3)                 # neither
    o = ctx.rdo_txsearch_batch(da, db, 32, 32, c, 0x201, 60, 3)                   # DCT_DCT + IDTX: fine
    torch.cuda.synchronize()
    assert o["eob"].shape == (4, 2)


def test_rdo_txsearch_equals_single_type_launches(ctx):
    """Size-independent property at frame scale: slot j of the fan-out equals r1_rdo_pixel_cand_batch /
    r1_rdo_full_cand_batch run with tx_type = the j-th type, on thousands of candidates (many waves)."""
    import torch
    from rav1e_amd import rdo_glue as RG
    for bd in (8, 10):
        a, b = planes(bd, w=640, h=384, seed=277)
        b.data[...] = np.clip(a.data.astype(np.int64) + np.random.default_rng(3).integers(-9, 10, a.data.shape), 0,
                              (1 << bd) - 1).astype(a.data.dtype)
        da, db = dev_plane(a), dev_plane(b)
        rng = np.random.default_rng(278)
        for ts in (0, 1, 2, 3, 7, 16):
            w, h = TX_SIZES[ts]
            n = 3001
            c = rand_rdo_cands(rng, n, a.width, a.height, w, h, 8, ts)
            c["rx"] = np.clip(c["rx"], c["ox"] - 2, c["ox"] + 2)
            c["ry"] = np.clip(c["ry"], c["oy"] - 2, c["oy"] + 2)
            mask = ctx.tx_type_mask(ts, True)
            types = RG.tx_type_slots(mask)
            px = ctx.rdo_txsearch_batch(da, db, w, h, c, mask, 80, 3, want_qcoeffs=True)
            tx = ctx.rdo_txsearch_batch(da, db, w, h, c, mask, 80, 0, want_est_rate=True)
            for j, t in enumerate(types):
                c["tx_type"] = t
                one = ctx.rdo_pixel_cand_batch(da, db, w, h, c, 80, 3, want_sad=False, want_satd=False,
                                               want_qcoeffs=True)
                assert torch.equal(px["eob"][:, j], one["eob"]), (bd, ts, t)
                assert torch.equal(px["dist"][:, j], one["dist"]), (bd, ts, t)
                assert torch.equal(px["qcoeffs"][:, j], one["qcoeffs"]), (bd, ts, t)
                full = ctx.rdo_full_cand_batch(da, db, w, h, c, 80, want_sad=False, want_satd=False)
                assert torch.equal(tx["dist"][:, j], full["tx_dist"]), (bd, ts, t)
                assert torch.equal(tx["est_rate"][:, j], full["est_rate"]), (bd, ts, t)


# ------------------------ N3: loop restoration (self-guided filter)
def _lrf_run(ctx, cdef, debl, ydec, fh, us, sh, units, bd):
    import torch
    h, w = cdef.shape
    hc, hd = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(debl, bd, 16, 16)
    dc, dd, do = dev_plane(hc), dev_plane(hd), dev_plane(hc)
    du = torch.from_numpy(np.ascontiguousarray(units).view(np.uint8).reshape(units.shape + (4,)).copy()).cuda()
    ctx.lrf_sgrproj_plane(dc, dd, do, ydec, w, h, fh, us, du, sh)
    dt = np.uint8 if bd == 8 else np.uint16
    return do.data.cpu().numpy().view(dt)[hc.yorigin:hc.yorigin + h, hc.xorigin:hc.xorigin + w]


@pytest.mark.parametrize("fixture", ["lrf_golden.npz", "lrf_ref.npz"])
def test_lrf_golden_frames(ctx, fixture):
    """the independent-model frames of tests/golden/lrf_golden.npz and the same frames filtered by
    the reference's own lrf_filter_frame text (lrf_ref.npz, tests/golden/gen_lrf_ref.py)"""
    G = dict(np.load(os.path.join(GOLD, fixture)))
    for name in sorted(k[:-5] for k in G if k.endswith("_meta")):
        w, h, ydec, fh, us, sh, bd = [int(v) for v in G[name + "_meta"]]
        got = _lrf_run(ctx, G[name + "_cdef"], G[name + "_debl"], ydec, fh, us, sh, G[name + "_units"], bd)
        bad = np.argwhere(got != G[name + "_out"])
        assert len(bad) == 0, (name, bad[:5])


@pytest.mark.parametrize("cfg", [(8, 0, 64, 64, 1), (10, 1, 32, 32, 2), (12, 0, 128, 64, 1), (8, 0, 256, 64, 4)])
def test_lrf_frame_vs_oracle(ctx, oracle, cfg):
    """720p-class planes against oracle/lrf.c: every parameter set, unit sizes 32..256 (the last
    unit of a row stretches), noisy and smooth content (the reference's wrapping p * s included)."""
    import ctypes as C
    bd, ydec, us, sh, noise = cfg
    w, h = (1280 >> (1 if ydec else 0)) - 4, (720 >> ydec) - 2
    fh = (h << ydec)
    rng = np.random.default_rng(40 + bd + us)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 23.0) * np.cos(yy / 17.0) + 1) / 2 * ((1 << bd) - 1)
    debl = np.clip(base + rng.integers(-8 * noise, 8 * noise + 1, (h, w)) * (1 << (bd - 8)), 0,
                   (1 << bd) - 1).astype(np.int64)
    debl[: h // 3] = rng.integers(0, 1 << bd, (h // 3, w))          # pure noise band: z saturates / wraps
    cdef = np.clip(debl + rng.integers(-2, 3, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1)
    cols, rows = max((w + us // 2) // us, 1), max((h + us // 2) // us, 1)
    units = np.zeros((rows, cols), O.LRF_UNIT)
    units["filter"] = rng.choice([0, 3, 3, 3], (rows, cols))
    units["set"] = rng.integers(0, 16, (rows, cols))
    units["xqd"][..., 0] = rng.integers(-96, 32, (rows, cols))
    units["xqd"][..., 1] = rng.integers(-32, 96, (rows, cols))
    hc, hd = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(debl, bd, 16, 16)
    ho = O.plane_from_image(cdef, bd, 16, 16)
    cc, cd, co = hc.cstruct(), hd.cstruct(), ho.cstruct()
    assert oracle.r1o_lrf_filter_plane(C.byref(cc), C.byref(cd), C.byref(co), ydec, w, h, fh, us, cols,
                                       rows, sh, units.ctypes.data, bd) == 0
    got = _lrf_run(ctx, cdef, debl, ydec, fh, us, sh, units, bd)
    bad = np.argwhere(got != ho.view())
    assert len(bad) == 0, (cfg, bad[:5])
    assert (ho.view() != cdef).sum() > cdef.size // 100      # the filter did something


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_sgrproj_solve_vs_oracle(ctx, oracle, bd):
    """r1_sgrproj_solve_batch: xqd for every parameter set on restoration units of all shapes
    (64..256 wide / high, stretched last units, frame corner) against oracle/lrf.c."""
    import ctypes as C
    from rav1e_amd.api import SGR_SOLVE_UNIT
    rng = np.random.default_rng(60 + bd)
    h, w = 400, 600
    yy, xx = np.mgrid[0:h, 0:w]
    sc = 1 << (bd - 8)
    src = np.clip(((np.sin(xx / 19.0) * np.cos(yy / 13.0) + 1) / 2 * 200 + 20 + rng.integers(-6, 7, (h, w))) * sc,
                  0, (1 << bd) - 1).astype(np.int64)               # texture + fine detail
    cdef = np.clip(src + rng.integers(-3, 4, (h, w)) * sc, 0, (1 << bd) - 1)   # + coding noise
    cdef[300:, 500:] = rng.integers(0, 1 << bd, (100, 100))        # pure noise: saturated / wrapped a
    hc, hs = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(src, bd, 16, 16)
    rects = [(0, 0, 64, 64), (64, 64, 64, 56), (128, 0, 256, 256), (384, 128, 216, 200),
             (0, 256, 96, 144), (500, 300, 100, 100), (37 * 4, 200, 36, 35)]
    u = np.zeros(len(rects) * 16, SGR_SOLVE_UNIT)
    for i, (x, y, ww, hh) in enumerate(rects):
        for s in range(16):
            u[i * 16 + s] = (x, y, ww, hh, s, (i + s) & 3, (0, 0))      # every R1_SGR_EDGE_* combination
    got = ctx.sgrproj_solve_batch(dev_plane(hc), dev_plane(hs), u).cpu().numpy()
    cc, cs = hc.cstruct(), hs.cstruct()
    for i in range(len(u)):
        want = np.zeros(2, np.int8)
        oracle.r1o_sgrproj_solve(C.byref(cc), C.byref(cs), int(u["x"][i]), int(u["y"][i]), int(u["w"][i]),
                                 int(u["h"][i]), int(u["set"][i]), int(u["edges"][i]), bd, want.ctypes.data)
        assert np.array_equal(got[i], want), (bd, u[i], got[i], want)
    assert len(np.unique(got, axis=0)) > 12      # the weights actually vary (not all clamped)


@pytest.mark.parametrize("bd", [8, 10])
def test_sgrproj_solve_ref(ctx, bd):
    """r1_sgrproj_solve_batch on what sgrproj_solve of the reference's own text returned (lrf_ref.npz)"""
    from rav1e_amd.api import SGR_SOLVE_UNIT
    REF = np.load(os.path.join(GOLD, "lrf_ref.npz"))
    cdef, src = REF["solve%d_cdef" % bd].astype(np.int64), REF["solve%d_src" % bd].astype(np.int64)
    hc, hs = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(src, bd, 16, 16)
    cases = REF["solve%d_cases" % bd]
    u = np.zeros(len(cases), SGR_SOLVE_UNIT)
    for i, (x0, y0, uw, uh, set_, q0, q1) in enumerate(cases.tolist()):
        u[i] = (x0, y0, uw, uh, set_, 3, (0, 0))     # the vectors were made on slices of the whole frame: LEFT | ABOVE
    got = ctx.sgrproj_solve_batch(dev_plane(hc), dev_plane(hs), u).cpu().numpy()
    assert np.array_equal(got.astype(np.int64), cases[:, 5:7].astype(np.int64)), (bd, got[:4], cases[:4])


def _lrf_search_cases():
    L = np.load(os.path.join(GOLD, "loop_decision_ref.npz"))
    return ["s0", "s1", "s2", "s3"] + sorted(k[:-5] for k in L.files if k.startswith("ldl") and k.endswith("_meta"))


@pytest.mark.parametrize("case", _lrf_search_cases())
def test_lrf_search_ref(ctx, case):
    """r1_lrf_search_batch (the restoration leg of rdo_loop_decision but the rate) on what
    setup_integral_image + sgrproj_solve + sgrproj_stripe_filter + rdo_loop_plane_error of the
    reference's own text returned: (xqd, err) per (unit, set) and the no-filter error, three planes
    (lrf_search_ref.npz: the callees executed on slices of whole-frame planes -- everything around a unit exists,
    R1_SGR_EDGE_LEFT | ABOVE) and on what rdo_loop_decision ITSELF, executed whole, made of its restoration leg
    (loop_decision_ref.npz, cases ldl*: the units, their order, their visible sizes and which of them see pixels
    left of / above themselves are that function's)"""
    import torch
    from rav1e_amd.api import SGR_SOLVE_UNIT
    REF = np.load(os.path.join(GOLD, "loop_decision_ref.npz" if case.startswith("ld") else "lrf_search_ref.npz"))
    W, H, xdec, ydec, bd, lru_sb = [int(v) for v in REF[case + "_meta"]]
    rows, want = REF[case + "_rows"], REF[case + "_err"]
    edges = REF[case + "_edges"] if case + "_edges" in REF.files else np.full(len(rows), 3, np.uint8)
    scales = torch.from_numpy(REF[case + "_scales"].astype(np.int64).astype(np.int32)).cuda()
    for pli in range(3):
        hi = O.plane_from_image(REF[case + "_in%d" % pli].astype(np.int64), bd, 16, 16)
        hs = O.plane_from_image(REF[case + "_src%d" % pli].astype(np.int64), bd, 16, 16)
        sel = rows[:, 0] == pli
        r = rows[sel]
        u = np.zeros(len(r), SGR_SOLVE_UNIT)
        for i, ((_, x, y, w, h, set_, q0, q1), e) in enumerate(zip(r.tolist(), edges[sel].tolist())):
            u[i] = (x, y, w, h, set_, e, (0, 0))
        xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
        # units up to 64 x 64: the one-launch kernel; max 256: moments / solve / filter-and-error launches
        for mx in ([64, 256] if r[:, 3:5].max() <= 64 else [256]):
            xqd, err = ctx.lrf_search_batch(dev_plane(hi), dev_plane(hs), u, is_chroma=pli != 0, xdec=xd, ydec=yd,
                                            scales=scales, dist_scale=int(REF[case + "_dscale"][pli]), max_w=mx, max_h=mx)
            assert np.array_equal(xqd.cpu().numpy().astype(np.int64), r[:, 6:8].astype(np.int64)), (case, pli, mx)
            got = err.cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want[sel]), (case, pli, mx, np.argwhere(got != want[sel])[:4].ravel(), got[:4],
                                                    want[sel][:4])


def test_lrf_entry_points_reject_planes_their_32_bit_offsets_cannot_address(ctx):
    """the restoration kernels address pixels with 32-bit byte offsets and 24-bit row / stride factors
    (csrc/lrf.hip, px_off): a descriptor beyond that is R1_EINVAL before anything is launched"""
    from rav1e_amd.api import SGR_SOLVE_UNIT
    a, b = planes(8, seed=3)
    da, db = dev_plane(a), dev_plane(b)
    u = np.array([(0, 0, 64, 64, 3, 0, (0, 0))], SGR_SOLVE_UNIT)
    ctx.lrf_search_batch(da, db, u, max_w=64, max_h=64)                 # as it is: fine
    for field, value in (("stride", 1 << 24), ("alloc_height", 1 << 24)):
        keep = getattr(da, field)
        setattr(da, field, value)
        try:
            with pytest.raises(RuntimeError):
                ctx.lrf_search_batch(da, db, u, max_w=64, max_h=64)
            with pytest.raises(RuntimeError):
                ctx.sgrproj_solve_batch(da, db, u)
        finally:
            setattr(da, field, keep)
    keep = (da.stride, da.alloc_height)
    da.stride, da.alloc_height = 1 << 16, 1 << 16                       # 4 GiB of bytes
    try:
        with pytest.raises(RuntimeError):
            ctx.lrf_search_batch(db, da, u, max_w=64, max_h=64)
    finally:
        da.stride, da.alloc_height = keep


@pytest.mark.parametrize("bd", [8, 10])
def test_lrf_search_vs_oracle_frame_units(ctx, oracle, bd):
    """every 64x64 luma unit (and the 32x32 units of a 4:2:0 chroma plane) of a 520x264 frame -- the last
    column of units 8 wide, the last row 8 high -- x {no filter, four parameter sets}, against
    oracle/lrf.c::r1o_lrf_search_unit"""
    import ctypes as C
    import torch
    from rav1e_amd.api import SGR_SOLVE_UNIT
    rng = np.random.default_rng(77 + bd)
    W, H = 520, 264
    yy, xx = np.mgrid[0:H, 0:W]
    srcY = np.clip((np.sin(xx / 7.0) + np.cos(yy / 5.0) + 2) / 4 * ((1 << bd) - 1), 0, (1 << bd) - 1).astype(np.int64)
    inY = np.clip(srcY + rng.integers(-8, 9, (H, W)) * (1 << (bd - 8)), 0, (1 << bd) - 1)
    grid = rng.integers(1 << 12, 1 << 16, ((H + 7) // 8, (W + 7) // 8)).astype(np.uint32)
    dscales = torch.from_numpy(grid.astype(np.int64).astype(np.int32)).cuda()
    for chroma in (False, True):
        xd = yd = 1 if chroma else 0
        s_, i_ = (srcY[::2, ::2], inY[::2, ::2]) if chroma else (srcY, inY)
        hi, hs = O.plane_from_image(i_, bd, 16, 16), O.plane_from_image(s_, bd, 16, 16)
        us = 64 >> xd
        u = []
        for y in range(0, s_.shape[0], us):
            for x in range(0, s_.shape[1], us):
                for set_ in (255, 2, 11, 14, 9):
                    u.append((x, y, min(us, s_.shape[1] - x), min(us, s_.shape[0] - y), set_, (x // us + 2 * (y // us) + set_) & 3,
                              (0, 0)))                                # every R1_SGR_EDGE_* combination
        u = np.array(u, SGR_SOLVE_UNIT)
        xqd, err = ctx.lrf_search_batch(dev_plane(hi), dev_plane(hs), u, is_chroma=chroma, xdec=xd, ydec=yd,
                                        scales=dscales, dist_scale=21000, max_w=64, max_h=64)
        x2, e2 = ctx.lrf_search_batch(dev_plane(hi), dev_plane(hs), u, is_chroma=chroma, xdec=xd, ydec=yd,
                                      scales=dscales, dist_scale=21000, max_w=128, max_h=128)
        assert torch.equal(xqd, x2) and torch.equal(err, e2)            # the two launch plans agree
        xqd, err = xqd.cpu().numpy(), err.cpu().numpy().view(np.uint64)
        ci, cs = hi.cstruct(), hs.cstruct()
        wx, we = np.zeros((len(u), 2), np.int8), np.zeros(len(u), np.uint64)
        for i in range(len(u)):
            assert oracle.r1o_lrf_search_unit(C.byref(ci), C.byref(cs), int(u["x"][i]), int(u["y"][i]), int(u["w"][i]),
                                              int(u["h"][i]), int(u["set"][i]), int(u["edges"][i]), int(chroma), xd, yd, grid.ctypes.data,
                                              grid.shape[1], 21000, bd, wx[i].ctypes.data, we[i:].ctypes.data) == 0
        assert np.array_equal(xqd, wx), (bd, chroma)
        assert np.array_equal(err, we), (bd, chroma, np.argwhere(err != we)[:4].ravel())


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_activity_scales_vs_oracle(ctx, oracle, bd):
    """r1_activity_scales (ActivityMask::from_plane + fill_scales) on a frame whose size is not a
    multiple of 8 (the last blocks read the padding, as the reference's aligned rect does)."""
    import ctypes as C
    hp = O.HostPlane(1284, 722, bd, rng=np.random.default_rng(3 + bd))
    hp.view()[100:300, 200:500] = (1 << bd) // 3                       # flat area
    hp.view()[300:400] = (hp.view()[300:400] >> (bd - 2)) << (bd - 2)  # coarse levels
    pc = hp.cstruct()
    hb, wb = (722 + 7) // 8, (1284 + 7) // 8
    wvar, wsc = np.zeros((hb, wb), np.uint32), np.zeros((hb, wb), np.uint32)
    oracle.r1o_activity_scales(C.byref(pc), O.ptr(wvar), O.ptr(wsc))
    var, sc = ctx.activity_scales(dev_plane(hp))
    assert np.array_equal(var.cpu().numpy().view(np.uint32), wvar)
    assert np.array_equal(sc.cpu().numpy().view(np.uint32), wsc)


@pytest.mark.parametrize("bd", [8, 10])
def test_cfl_alpha_search_vs_oracle(ctx, oracle, bd):
    """r1_cfl_alpha_search_batch (rdo_cfl_alpha): the oracle predicts UV_CFL_PRED for all 33
    alphas, takes the plain SSE over the visible area and replays the reference's selection."""
    from rav1e_amd.api import CFL_ALPHA_CAND, CFL_AC_CAND, INTRA_EDGE_CAND
    rng = np.random.default_rng(410 + bd)
    luma = O.HostPlane(256, 128, bd, rng=rng)
    # chroma planes (4:2:0): reconstruction and source share structure with the luma
    lv = luma.view().astype(np.int64)
    sub = (lv[0::2, 0::2] + lv[0::2, 1::2] + lv[1::2, 0::2] + lv[1::2, 1::2]) // 4
    rec = O.plane_from_image(np.clip(sub // 2 + rng.integers(0, 1 << (bd - 1), sub.shape), 0, (1 << bd) - 1), bd, 44, 44)
    src = O.plane_from_image(np.clip(sub * 3 // 4 + rng.integers(0, 1 << (bd - 2), sub.shape), 0, (1 << bd) - 1), bd, 44, 44)
    dl, dr, dsrc = dev_plane(luma), dev_plane(rec), dev_plane(src)
    hbd = int(bd > 8)
    dt = np.uint16 if hbd else np.uint8
    for ts in (0, 1, 2, 3, 7, 8):
        w, h = TX_SIZES[ts]
        n = 40
        gx, gy = rec.width // w, rec.height // h
        bx, by = rng.integers(0, gx, n) * w, rng.integers(0, gy, n) * h
        bx[:3], by[:3] = [0, w, 0], [0, 0, h]
        ec = np.zeros(n, INTRA_EDGE_CAND)
        ec["x"], ec["y"], ec["mode"], ec["flags"] = bx, by, 13, 1
        edges, lens = ctx.intra_edges_batch(dr, (0, 0, rec.width, rec.height), ts, ec)
        he, hl = edges.cpu().numpy().view(dt), lens.cpu().numpy()
        ac_c = np.zeros(n, CFL_AC_CAND)
        ac_c["x"], ac_c["y"] = bx * 2, by * 2
        ac = ctx.cfl_ac_batch(dl, w, h, 1, 1, ac_c)
        hac = ac.cpu().numpy()
        var = np.where((bx == 0) & (by == 0), 0, np.where(by == 0, 1, np.where(bx == 0, 2, 3)))
        cc = np.zeros(n, CFL_ALPHA_CAND)
        cc["x"], cc["y"], cc["variant"] = bx, by, var
        cc["vis_w"] = np.where(rng.random(n) < 0.2, rng.integers(1, w + 1, n), w)
        cc["vis_h"] = np.where(rng.random(n) < 0.2, rng.integers(1, h + 1, n), h)
        alpha, cost = ctx.cfl_alpha_search_batch(dsrc, ts, cc, edges, lens, ac)
        alpha, cost = alpha.cpu().numpy(), cost.cpu().numpy()
        for i in range(n):
            vw, vh = int(cc["vis_w"][i]), int(cc["vis_h"][i])
            s = src.view()[by[i]:by[i] + vh, bx[i]:bx[i] + vw].astype(np.int64)
            costs = {}
            for a in range(-16, 17):
                out = np.zeros((h, w), dt)
                # alpha 0 is plain DC_PRED in the reference's dispatch (predict.rs:119-123)
                assert oracle.r1o_dispatch_predict_intra(13 if a else 0, int(var[i]), O.ptr(out), w, ts, bd,
                                                         O.ptr(hac[i]), a, 0, O.ptr(he[i]), int(hl[i, 0]),
                                                         int(hl[i, 1]), w, h, hbd) == 0
                d = s - out[:vh, :vw].astype(np.int64)
                costs[a] = int((d * d).sum())
            best, best_a, count = costs[0], 0, 2
            for a in range(1, 17):
                if costs[a] < best:
                    best, best_a, count = costs[a], a, count + 2
                if costs[-a] < best:
                    best, best_a, count = costs[-a], -a, count + 2
                if count < a:
                    break
            assert (int(alpha[i]), int(cost[i])) == (best_a, best), (bd, ts, i, int(alpha[i]), best_a)


def test_widened_entry_points_reject_bad_arguments(ctx):
    """where the reference would panic (asserts on geometry / bit depth / aliasing) the batch
    calls return R1_EINVAL and say why (r1_last_error)"""
    import torch
    import deblock_util as D
    from rav1e_amd.api import R1Error, Plane, me_lambdas, RDO_CAND
    a = Plane(128, 64, 8)
    st = torch.zeros((16, 32, 2), dtype=torch.int32, device="cuda")
    pyr = [a, a, a]
    with pytest.raises(R1Error):      # tile origin not superblock aligned
        ctx.estimate_tile_motion([dict(org=pyr, ref=pyr, stats=st, tile=(32, 0, 64, 64))], 32, 16, 8,
                                 me_lambdas(10.0))
    with pytest.raises(R1Error):      # tile outside the stats array
        ctx.estimate_tile_motion([dict(org=pyr, ref=pyr, stats=st, tile=(64, 0, 128, 64))], 32, 16, 8,
                                 me_lambdas(10.0))
    blocks = torch.zeros((16, 32, 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(R1Error):      # chroma decimation on the luma plane
        ctx.deblock_plane(D.make_state([10, 10, 10, 10]), a, 0, 1, 1, blocks, 128, 64)
    units = torch.zeros((1, 2, 4), dtype=torch.uint8, device="cuda")
    with pytest.raises(R1Error):      # in-place restoration: out must not alias the CDEF output
        ctx.lrf_sgrproj_plane(a, a, a, 0, 128, 64, 64, 64, units, 64)
    with pytest.raises(R1Error):      # unit size not a multiple of 32
        ctx.lrf_sgrproj_plane(a, a, Plane(128, 64, 8), 0, 128, 64, 64, 48, units, 64)
    c = np.zeros(4, RDO_CAND)
    with pytest.raises(R1Error):      # cdef_dist is a luma-only distortion
        ctx.rdo_pixel_cand_batch(a, a, 16, 16, c, 100, 3, xdec=1, ydec=1)


# ------------------------ N1: the selection step of the mode pre-screens
@pytest.mark.gpu
def test_prescreen_select_matches_the_reference_sorts(ctx):
    """r1_prescreen_select_batch against the reference's own statements run on the host:
    intra  `modes[num_modes_rdo / 2..].sort_by_key(|&a| satds[a])` + take (src/rdo.rs:1504-1509),
    inter  `sorted.sort_by_key(satd)` + take (src/rdo.rs:1352-1357); Rust's sort_by_key is
    stable, like Python's sorted().  Keys with many ties, every (group, keep_head, k)."""
    import torch
    from rav1e_amd.api import R1Error
    rng = np.random.default_rng(77)
    for group in (1, 2, 3, 13, 20, 64):
        n_groups = 257
        for hi in (3, 1 << 20):                           # hi = 3: almost everything ties
            keys = rng.integers(0, hi, (n_groups, group)).astype(np.uint32)
            dk = torch.from_numpy(keys.view(np.int32).reshape(-1).copy()).cuda()
            for k in sorted({1, min(3, group), min(7, group), group}):
                for head in sorted({0, k // 2, k}):
                    got = ctx.prescreen_select_batch(dk, group, head, k).cpu().numpy()
                    for g in range(n_groups):
                        modes = list(range(group))
                        modes[head:] = sorted(modes[head:], key=lambda a: keys[g, a])
                        assert list(got[g]) == modes[:k], (group, hi, k, head, g)
    e = torch.zeros(0, dtype=torch.int32, device="cuda")
    assert ctx.prescreen_select_batch(e, 13, 1, 3).shape == (0, 3)        # empty batch
    one = torch.zeros(13, dtype=torch.int32, device="cuda")
    for bad in ((65, 0, 1), (13, 0, 0), (13, 0, 14), (13, 4, 3), (13, -1, 3)):
        with pytest.raises(R1Error):
            ctx.prescreen_select_batch(one if bad[0] == 13 else torch.zeros(65, dtype=torch.int32, device="cuda"),
                                       *bad)


# ------------------------ N1: update_block_importances (f32 scatter-add in the reference's order)
@pytest.mark.gpu
def test_update_block_importances_bit_exact(ctx, oracle):
    """r1_update_block_importances against the oracle, compared as raw f32 bits: random motion
    fields (in range, far off-frame, block-aligned), many sources per destination (all vectors
    point at one block), 4K-sized maps; then chained over three references like the caller does."""
    import torch
    from test_oracle_lookahead import importance_case
    rng = np.random.default_rng(43)

    def run(w, h, length, intra, future, inter, mvs, ref_imp):
        want = ref_imp.copy()
        oracle.r1o_update_block_importances(O.ptr(intra), O.ptr(future), O.ptr(inter), O.ptr(mvs), w, h,
                                            length, O.ptr(want))
        got = ctx.update_block_importances(_t(intra.view(np.int32)), _t(future), _t(inter.view(np.int32)),
                                           _t(mvs), w, h, length, _t(ref_imp.copy()))
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), (w, h, length)
        return want

    for (w, h, mvr, length) in ((24, 16, 300, 1), (17, 9, 2000, 3), (1, 1, 64, 2), (480, 270, 700, 4),
                                (240, 135, 30000, 7)):
        run(w, h, length, *importance_case(rng, w, h, mvr))
    # every block points at (about) the same destination: long sequential chains
    w, h = 64, 40
    intra, future, inter, mvs, ref_imp = importance_case(rng, w, h, 20)
    yy, xx = np.divmod(np.arange(w * h), w)
    mvs[:, 0] += ((h // 2 - yy) * 64).astype(np.int16)
    mvs[:, 1] += ((w // 2 - xx) * 64).astype(np.int16)
    run(w, h, 2, intra, future, inter, mvs, ref_imp)
    # chained like compute_block_importances: the same reference map updated from three frames
    w, h = 120, 68
    acc = np.zeros(w * h, np.float32)
    dacc = _t(acc.copy())
    for _ in range(3):
        intra, future, inter, mvs, _unused = importance_case(rng, w, h, 500)
        oracle.r1o_update_block_importances(O.ptr(intra), O.ptr(future), O.ptr(inter), O.ptr(mvs), w, h,
                                            3, O.ptr(acc))
        ctx.update_block_importances(_t(intra.view(np.int32)), _t(future), _t(inter.view(np.int32)),
                                     _t(mvs), w, h, 3, dacc)
    assert np.array_equal(dacc.cpu().numpy().view(np.uint32), acc.view(np.uint32))
    # empty map and argument checks
    from rav1e_amd.api import R1Error
    e = torch.zeros(0, dtype=torch.int32, device="cuda")
    ctx.update_block_importances(e, e.float(), e, e.short(), 0, 0, 1, e.float())
    with pytest.raises(R1Error):
        ctx.update_block_importances(e, e.float(), e, e.short(), 4, 4, 0, e.float())


# ------------------------ N3: deblocking, all three planes per launch
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(8, 1, 1, False), (10, 1, 1, True), (12, 0, 0, True), (8, 1, 0, False)])
def test_deblock_frame_entry_points_vs_oracle(ctx, oracle, cfg):
    """r1_deblock_sse_frame (six tallies, one launch) and r1_deblock_frame (three planes, two
    launches) against oracle/deblock.c plane by plane; a zero chroma level switches that plane off"""
    import ctypes as C
    import torch
    import deblock_util as D
    bd, xdec, ydec, deltas = cfg
    w, h, cw, ch = 640, 384, 636, 378
    rng = np.random.default_rng(170 + bd + xdec)
    blocks = D.random_blocks(rng, w // 4, h // 4, xdec, ydec, deltas=deltas)
    dblocks = torch.from_numpy(blocks.view(np.uint8).reshape(blocks.shape + (8,)).copy()).cuda()
    dt = np.uint8 if bd == 8 else np.uint16
    imgs = _deblock_planes(rng, w, h, bd, xdec, ydec, blocks)
    for levels in ([int(v) for v in rng.integers(8, 50, 4)], [20, 0, 0, 17]):
        state = D.make_state(levels, rng, deltas, deltas)
        hp = [O.plane_from_image(rec, bd, 24, 24) for rec, _ in imgs]
        hs = [O.plane_from_image(src, bd, 24, 24) for _, src in imgs]
        dp, ds = [dev_plane(p) for p in hp], [dev_plane(p) for p in hs]
        want_t = np.zeros((3, 2, 65), np.int64)
        for pli in range(3):
            xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
            pc, sc = hp[pli].cstruct(), hs[pli].cstruct()
            assert oracle.r1o_deblock_sse_plane(C.byref(pc), C.byref(sc), pli, xd, yd, blocks.ctypes.data,
                                                blocks.shape[1], blocks.shape[1], blocks.shape[0], cw, ch, bd,
                                                want_t[pli, 0].ctypes.data, want_t[pli, 1].ctypes.data) == 0
        got_t = ctx.deblock_sse_frame(dp, ds, xdec, ydec, dblocks, cw, ch)
        assert np.array_equal(got_t.cpu().numpy(), want_t), (cfg, levels)
        for pli in range(3):
            xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
            pc = hp[pli].cstruct()
            assert oracle.r1o_deblock_plane(state.ctypes.data, C.byref(pc), pli, xd, yd, blocks.ctypes.data,
                                            blocks.shape[1], blocks.shape[1], blocks.shape[0], cw, ch, bd) == 0
        ctx.deblock_frame(state, dp, xdec, ydec, dblocks, cw, ch)
        for pli in range(3):
            got = dp[pli].data.cpu().numpy().view(dt)
            assert np.array_equal(got, hp[pli].data), (cfg, levels, pli)
        assert (hp[0].view() != imgs[0][0]).sum() > 0
        if levels[2] == 0:
            assert np.array_equal(hp[1].view(), imgs[1][0])     # switched off: untouched


@pytest.mark.parametrize("prep", [False, True])
def test_mc_mfma_variant_equals_dot4_path(ctx, oracle, prep):
    """r1_mc_batch_mfma (horizontal 8-tap pass as a banded-Toeplitz v_mfma_i32_16x16x32_i8,
    csrc/mc_mfma.hip) against the oracle and against r1_mc_put_batch / r1_mc_prep_batch."""
    import ctypes as C
    a, _ = planes(8, seed=77)
    da = dev_plane(a)
    rng = np.random.default_rng(1234 + int(prep))
    for s in (8, 16, 32, 64):
        n = 203 if s <= 16 else 37      # ragged: partial waves
        c = rand_mc_cands(rng, n, a.width, a.height, s, s, 60)
        c["mode_x"] = rng.integers(0, 4, n)
        c["mode_y"] = rng.integers(0, 4, n)
        pa = a.cstruct()
        if prep:
            want = np.zeros((n, s, s), np.int16)
            assert oracle.r1o_mc_prep_batch(C.byref(pa), s, s, O.ptr(c), n, O.ptr(want)) == 0
            ref = ctx.prep_8tap_batch(da, s, s, c).cpu().numpy()
        else:
            want = np.zeros((n, s, s), np.uint8)
            assert oracle.r1o_mc_put_batch(C.byref(pa), s, s, O.ptr(c), n, O.ptr(want)) == 0
            ref = ctx.put_8tap_batch(da, s, s, c).cpu().numpy()
        got = ctx.mc_batch_mfma(da, s, s, c, prep=prep).cpu().numpy()
        assert np.array_equal(ref, want), (s, "dot4 path")
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (s, prep, bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])


# ------------------------ a14: CDEF strength search (rdo_loop_decision's CDEF leg)
@pytest.mark.parametrize("cfg", [
    # w, h, bd, (xdec, ydec), planes, n_idx, area
    (320, 200, 8, (1, 1), 3, 8, (1, 1)),
    (264, 136, 10, (1, 1), 3, 8, (4, 4)),     # restoration units of 256: areas of 4 x 4 superblocks
    (192, 128, 8, (0, 0), 3, 4, (2, 1)),      # 4:4:4
    (192, 136, 12, (1, 0), 3, 8, (1, 2)),     # 4:2:2 (Cdef_Uv_Dir mapping)
    (136, 200, 10, (1, 1), 1, 2, (1, 1)),     # monochrome
])
def test_cdef_strength_search_vs_oracle(ctx, oracle, cfg):
    import ctypes as C
    from test_gpu_ref_vectors import run_cdef_search_gpu
    w, h, bd, (xdec, ydec), planes, n_idx, area = cfg
    rng = np.random.default_rng(w * 7 + h + bd)
    yy, xx = np.mgrid[0:h, 0:w]
    Y = np.clip(((np.sin(xx / 6.0) + np.cos((yy + 2 * xx) / 9.0)) * 45 + 128 + rng.integers(-5, 6, (h, w))), 0, 255)
    Y = Y.astype(np.int64) << (bd - 8)
    cw, ch = w >> xdec, h >> ydec
    U = rng.integers(0, 1 << bd, (ch, cw)) // 4 + (1 << (bd - 2))
    V = np.clip(Y[::1 << ydec, ::1 << xdec][:ch, :cw] // 2 + (30 << (bd - 8)), 0, (1 << bd) - 1)
    src_i = [Y, U, V]
    rec_i = [np.clip(s + rng.integers(-12 << (bd - 8), (12 << (bd - 8)) + 1, s.shape) * (rng.random(s.shape) < 0.3),
                     0, (1 << bd) - 1) for s in src_i]
    rec = [O.plane_from_image(a, bd, 16, 16) for a in rec_i]
    src = [O.plane_from_image(a, bd, 16, 16) for a in src_i]
    mi_cols, mi_rows = 2 * ((w + 7) // 8), 2 * ((h + 7) // 8)
    skip = (rng.random((mi_rows, mi_cols)) < 0.4).astype(np.uint8)
    skip[16:32, :16] = 1
    scales = rng.integers(1 << 11, 1 << 17, ((h + 7) // 8, (w + 7) // 8)).astype(np.uint32)
    for use_scales in (True, False):
        prm = O.CdefSearchParams()
        prm.y_strengths[:] = [int(v) for v in rng.integers(0, 64, 8)]
        prm.uv_strengths[:] = [int(v) for v in rng.integers(0, 64, 8)]
        prm.y_strengths[0], prm.uv_strengths[0] = 0, 0
        prm.damping, prm.bit_depth, prm.n_idx, prm.planes = int(rng.integers(3, 7)), bd, n_idx, planes
        prm.xdec, prm.ydec, prm.crop_w, prm.crop_h = xdec, ydec, w, h
        prm.area_sb_w, prm.area_sb_h = area
        prm.dist_scale[:] = [int(v) for v in rng.integers(1 << 12, 1 << 16, 3)]
        n_sbx, n_sby = (mi_cols + 15) // 16, (mi_rows + 15) // 16
        want_err = np.zeros((n_sby, n_sbx, 8), np.uint64)
        want_best = np.zeros((n_sby, n_sbx), np.int8)
        pr = (O.Plane * 3)(*[p.cstruct() for p in rec])
        ps = (O.Plane * 3)(*[p.cstruct() for p in src])
        sc = scales if use_scales else None
        assert oracle.r1o_cdef_strength_search(pr, ps, skip.ctypes.data, mi_cols, mi_cols, mi_rows,
                                               sc.ctypes.data if sc is not None else None,
                                               sc.shape[1] if sc is not None else 0, C.byref(prm),
                                               want_err.ctypes.data, want_best.ctypes.data) == 0
        got_err, got_best = run_cdef_search_gpu(ctx, rec, src, skip, sc, prm)
        bad = np.argwhere(got_err != want_err)
        assert len(bad) == 0, (cfg, use_scales, bad[:4], got_err[tuple(bad[0])], want_err[tuple(bad[0])])
        assert np.array_equal(got_best, want_best), (cfg, got_best, want_best)
        assert (want_best == -1).any() and (want_best >= 0).any()
