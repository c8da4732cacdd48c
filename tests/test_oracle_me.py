"""Motion-estimation oracle (oracle/me.c): the reference holds no vectors for
src/me.rs (PARITY UNPINNED, see the file header), so these tests pin what the
algorithm must satisfy structurally, whatever the search path taken."""
import os

import numpy as np
import pytest

import oracle_lib as O


def smooth_image(w, h, bd, seed):
    """band-limited texture: the hierarchical search needs structure at every scale"""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((h + 64, w + 64))
    for _ in range(3):   # separable blur
        f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
    f = (f - f.min()) / (f.max() - f.min())
    return (f * ((1 << bd) - 1)).astype(np.int64)


LAMBDAS = [40, 10, 2]   # by ssdec, the order of magnitude of speed-6 mid-quality settings


@pytest.mark.parametrize("bd", [8, 10])
def test_zero_motion_is_found(oracle, bd):
    w, h = 192, 128
    img = smooth_image(w, h, bd, 1)[:h, :w]
    pyr = O.me_pyramid(img, bd)
    stats = np.zeros((h // 4, w // 4), O.ME_STATS)
    O.me_oracle(oracle, pyr, pyr, w // 4, h // 4, (0, 0, w, h), bd, LAMBDAS, stats)
    assert not stats["row"].any() and not stats["col"].any()
    assert not stats["normalized_sad"].any()


@pytest.mark.parametrize("shift", [(8, -4), (-12, 8), (4, 16)])
def test_pure_translation_is_recovered(oracle, shift):
    """ref(x, y) = org(x + dx, y + dy) with dx, dy multiples of 4 (so the half and
    quarter resolution planes are shifted copies too): every block away from the
    frame edge must end on mv = (8 dy, 8 dx) with SAD 0."""
    dx, dy = shift
    w, h, bd = 256, 192, 8
    big = smooth_image(w, h, bd, 2)
    org = big[32:32 + h, 32:32 + w]
    ref = big[32 - dy:32 - dy + h, 32 - dx:32 - dx + w]    # ref(x + dx, y + dy) == org(x, y)
    po, pr = O.me_pyramid(org, bd), O.me_pyramid(ref, bd)
    stats = np.zeros((h // 4, w // 4), O.ME_STATS)
    O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, LAMBDAS, stats)
    inner = stats[8:-8, 8:-8]
    assert (inner["row"] == 8 * dy).all() and (inner["col"] == 8 * dx).all()
    assert not inner["normalized_sad"].any()
    # every 16x16 block carries one vector (save_me_stats fills the block)
    blk = stats[:, :].reshape(h // 16, 4, w // 16, 4)
    assert (blk == blk[:, :1, :, :1]).all()


def test_vectors_respect_the_mv_range_and_tiles_are_independent(oracle):
    """noise input (no true motion): results stay inside get_mv_range; a tile's
    result does not depend on what lies outside its stats rectangle."""
    w, h, bd = 192, 128, 8
    rng = np.random.default_rng(3)
    po = O.me_pyramid(rng.integers(0, 256, (h, w)), bd)
    pr = O.me_pyramid(rng.integers(0, 256, (h, w)), bd)
    prev = np.zeros((h // 4, w // 4), O.ME_STATS)
    prev["row"] = rng.integers(-64, 65, prev.shape)
    prev["col"] = rng.integers(-64, 65, prev.shape)
    prev["normalized_sad"] = rng.integers(0, 1 << 20, prev.shape)
    full = np.zeros((h // 4, w // 4), O.ME_STATS)
    O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, LAMBDAS, full, prev)
    bx = np.arange(w // 4)[None, :] * 32
    by = np.arange(h // 4)[:, None] * 32
    assert (full["col"] >= -bx - (128 + 64 * 8)).all() and (full["row"] >= -by - (128 + 64 * 8)).all()
    # two tiles: the right one alone, garbage in the left half of its stats array
    a, b = np.zeros_like(full), np.zeros_like(full)
    b["row"][:, : 128 // 4] = 77
    for s in (a, b):
        O.me_oracle(oracle, po, pr, w // 4, h // 4, (128, 0, 64, h), bd, LAMBDAS, s, prev)
    assert np.array_equal(a[:, 128 // 4:], b[:, 128 // 4:])
    assert (b["row"][:, : 128 // 4] == 77).all()        # and nothing outside the tile is written


def test_full_search_never_worsens_the_first_pass(oracle):
    w, h, bd = 128, 64, 8
    rng = np.random.default_rng(4)
    po = O.me_pyramid(rng.integers(0, 256, (h, w)), bd)
    pr = O.me_pyramid(rng.integers(0, 256, (h, w)), bd)
    a = np.zeros((h // 4, w // 4), O.ME_STATS)
    b = np.zeros_like(a)
    O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, LAMBDAS, a, allow_full_search=0)
    O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, LAMBDAS, b, allow_full_search=1)
    # the full search only ever replaces the first pass' result by a cheaper one; after
    # the refinement passes both runs are complete, block-constant maps
    for s_ in (a, b):
        blk = s_.reshape(h // 16, 4, w // 16, 4)
        assert (blk == blk[:, :1, :, :1]).all()


# ---- cross-check against the second restatement (tests/me_model.py) -----------------------
@pytest.mark.parametrize("case", [
    # (w, h, bd, image kind, tile, previous-frame stats, allow_hp, allow_full_search)
    (128, 64, 8, "smooth", None, False, 1, 0),
    (136, 72, 8, "noise", None, True, 1, 1),          # cropped superblocks, full search stage
    (200, 136, 10, "smooth", None, True, 0, 0),       # 10-bit, quarter-pel rates
    (192, 128, 8, "flat", (64, 0, 128, 128), True, 1, 0),   # a tile inside the frame, many ties
])
def test_tile_motion_equals_the_independent_model(oracle, case):
    """oracle/me.c against tests/me_model.py: every MEStats entry (mv and normalised SAD) of the
    three-pass search, bit for bit"""
    import me_model
    w, h, bd, kind, tile, use_prev, hp, full = case
    rng = np.random.default_rng(w + h)
    if kind == "noise":
        org, ref = rng.integers(0, 1 << bd, (h, w)), rng.integers(0, 1 << bd, (h, w))
    else:
        f = rng.standard_normal((h + 64, w + 64))
        for _ in range(3):
            f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
            f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
        f = ((f - f.min()) / (f.max() - f.min()) * ((1 << bd) - 1)).astype(np.int64)
        if kind == "flat":
            f = (f >> (bd - 3)) << (bd - 3)
        org = f[32:32 + h, 32:32 + w]
        ref = np.clip(f[32 + 6:32 + 6 + h, 32 - 11:32 - 11 + w] + rng.integers(-2, 3, (h, w)), 0, (1 << bd) - 1)
    po, pr = O.me_pyramid(org, bd), O.me_pyramid(ref, bd)
    tile = tile or (0, 0, w, h)
    prev = None
    if use_prev:
        prev = np.zeros((h // 4, w // 4), O.ME_STATS)
        prev["row"] = rng.integers(-90, 91, prev.shape)
        prev["col"] = rng.integers(-90, 91, prev.shape)
        prev["normalized_sad"] = rng.integers(0, 1 << 21, prev.shape)
    init = np.zeros((h // 4, w // 4), O.ME_STATS)
    init["row"] = rng.integers(-30, 31, init.shape)
    init["col"] = rng.integers(-30, 31, init.shape)
    init["normalized_sad"] = rng.integers(0, 1 << 21, init.shape)
    lam = [int(v) for v in rng.integers(5, 60, 3)]
    want = init.copy()
    O.me_oracle(oracle, po, pr, w // 4, h // 4, tile, bd, lam, want, prev, allow_hp=hp,
                allow_full_search=full)
    m = me_model.Model(po, pr, w // 4, h // 4, bd, lam, allow_hp=hp, allow_full_search=full)
    got = m.estimate_tile_motion(init.copy(), tile, prev)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (case, len(bad), bad[:4], got[tuple(bad[0])], want[tuple(bad[0])])
    assert (want != init).any()


@pytest.mark.parametrize("shift", [(4, -2), (4, 2), (2, 3), (-2, 3), (-2, -3), (0, 4), (0, -4), (8, 4), (-12, 9),
                                    (5, 0), (-7, 0), (0, 3), (0, -9), (1, 1), (-6, -2), (3, -5)])
def test_tile_motion_pattern_entries_against_the_model(oracle, shift):
    """delta-shaped SAD landscapes: noise, the reference an exact copy displaced by `shift`
    quarter-resolution pixels.  In the first (extensive) pass nothing but the search patterns
    can find the displacement -- the long cross arms, the 16-entry uneven hexagon at its six
    scales (with its duplicated entry: (-2, -3) is not in it), the hexagon and square
    refinements -- so a wrong entry, order or scale in either restatement changes the result."""
    import me_model
    w, h, bd = 320, 192, 8
    rng = np.random.default_rng(1000 + 31 * shift[0] + shift[1])
    big = rng.integers(0, 256, (h + 256, w + 256))
    if (shift[0] + shift[1]) % 2:
        # every other case on a smooth texture + noise: the SAD landscape has a slope, the
        # searches walk it step by step (hexagon iterations, square refinement, diamond)
        f = rng.standard_normal(big.shape)
        for _ in range(4):
            f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
            f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
        big = ((f - f.min()) / (f.max() - f.min()) * 255).astype(np.int64) + rng.integers(-3, 4, big.shape)
        big = np.clip(big, 0, 255)
    org = big[128:128 + h, 128:128 + w]
    ref = big[128 + 4 * shift[0]:128 + 4 * shift[0] + h, 128 + 4 * shift[1]:128 + 4 * shift[1] + w].copy()
    # the first superblock column is an undisplaced copy: its blocks end with SAD 0, which makes
    # the early-exit threshold of their right-hand neighbours small (me.rs:768-769) -- those see
    # a useless predictor (0, 0), fail the threshold and have to run the uneven multi-hexagon search
    ref[:, :64] = org[:, :64]
    po, pr = O.me_pyramid(org, bd), O.me_pyramid(ref, bd)
    lam = [40, 10, 3]
    want = np.zeros((h // 4, w // 4), O.ME_STATS)
    O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, want)
    got = me_model.Model(po, pr, w // 4, h // 4, bd, lam).estimate_tile_motion(np.zeros_like(want), (0, 0, w, h))
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (shift, len(bad), bad[:4], got[tuple(bad[0])], want[tuple(bad[0])])


@pytest.mark.parametrize("cfg", [(8, 1, 1), (8, 0, 1), (10, 1, 0), (8, 1, 0)])
def test_rdo_time_estimate_motion_equals_the_independent_model(oracle, cfg):
    """the RDO-time estimate_motion (predicted MVs in the rate, SATD re-cost, sub-pel diamond
    through put_8tap) of oracle/me.c against tests/me_model.py: mv, sad and cost of every block"""
    import me_model
    bd, use_satd, allow_hp = cfg
    w, h = 192, 128
    rng = np.random.default_rng(7 + bd + 2 * use_satd + allow_hp)
    f = rng.standard_normal((h + 64, w + 64))
    for _ in range(3):
        f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
    f = ((f - f.min()) / (f.max() - f.min()) * ((1 << bd) - 1)).astype(np.int64)
    org = f[32:32 + h, 32:32 + w]
    ref = np.clip(f[32 - 3:32 - 3 + h, 32 + 5:32 + 5 + w] + rng.integers(-3, 4, (h, w)), 0, (1 << bd) - 1)
    po, pr = O.me_pyramid(org, bd), O.me_pyramid(ref, bd)
    lam = [30, 8, 2]
    stats = np.zeros((h // 4, w // 4), O.ME_STATS)
    O.me_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, stats, allow_hp=allow_hp)
    cands = []
    for (bw, bh) in ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (64, 32), (4, 4), (4, 8)):
        for _ in range(6):
            bx = int(rng.integers(0, (w - bw) // 4 + 1))
            by = int(rng.integers(0, (h - bh) // 4 + 1))
            cands.append((bx, by, bw, bh, int(rng.integers(0, 4)) * 2 + 1,
                          [[int(v) for v in rng.integers(-40, 41, 2)] for _ in range(2)]))
    c = np.zeros(len(cands), O.ME_BLOCK_CAND)
    for i, (bx, by, bw, bh, corner, pmv) in enumerate(cands):
        c[i]["bx"], c[i]["by"], c[i]["w"], c[i]["h"], c[i]["corner"], c[i]["pmv"] = bx, by, bw, bh, corner, pmv
    want = O.me_block_oracle(oracle, po, pr, w // 4, h // 4, (0, 0, w, h), bd, lam, stats, None, c,
                             use_satd=use_satd, filter_mode=0, allow_hp=allow_hp)
    taps = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mc_golden.npz"))["filters"]
    m = me_model.Model(po, pr, w // 4, h // 4, bd, lam, allow_hp=allow_hp)
    for i, (bx, by, bw, bh, corner, pmv) in enumerate(cands):
        got = me_model.estimate_motion_block(m, taps, stats, None, (0, 0, w, h), bx, by, bw, bh,
                                             (bool(corner & 2), bool(corner & 4)),
                                             (tuple(pmv[0]), tuple(pmv[1])), use_satd)
        wnt = (int(want[i]["row"]), int(want[i]["col"]), int(want[i]["sad"]), int(want[i]["cost"]))
        assert got == wnt, (cfg, i, cands[i], got, wnt)
