"""Motion-estimation oracle (oracle/me.c): the reference holds no vectors for
src/me.rs (PARITY UNPINNED, see the file header), so these tests pin what the
algorithm must satisfy structurally, whatever the search path taken."""
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
