"""oracle/me.c against tests/golden/me_ref.npz: MEStats maps and RDO-time block searches
produced by EXECUTING the reference's own src/me.rs text (tests/golden/gen_me_ref.py,
tools/rustlite) -- the pin of the motion-estimation oracle to the reference."""
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF = np.load(os.path.join(HERE, "golden", "me_ref.npz"))
CASES = sorted(k[:-5] for k in REF.files if k.endswith("_meta"))


def load_case(name):
    """-> dict(meta fields, org pyramid, per-reference (pyramid, prev, want))"""
    w, h, bd, tx, ty, tw, th, hp, full, scale, n_refs, _ = [int(v) for v in REF[name + "_meta"]]
    pads = (88, 44, 22)
    org = [O.plane_from_image(REF["%s_org%d" % (name, s)].astype(np.int64), bd, pads[s], pads[s]) for s in range(3)]
    refs = []
    for k in range(n_refs):
        pyr = [O.plane_from_image(REF["%s_ref%d_%d" % (name, k, s)].astype(np.int64), bd, pads[s], pads[s])
               for s in range(3)]
        refs.append((pyr, to_stats(REF["%s_prev%d" % (name, k)]), to_stats(REF["%s_stats%d" % (name, k)])))
    return dict(w=w, h=h, bd=bd, tile=(tx, ty, tw, th), hp=hp, full=full, scale=scale, org=org, refs=refs,
                lam=[int(v) for v in REF[name + "_lambda"]])


def to_stats(a):
    s = np.zeros(a.shape[:2], O.ME_STATS)
    s["row"], s["col"], s["normalized_sad"] = a[..., 0], a[..., 1], a[..., 2]
    return s


@pytest.mark.parametrize("name", CASES)
def test_tile_motion_equals_the_executed_reference(oracle, name):
    c = load_case(name)
    for k, (pyr, prev, want) in enumerate(c["refs"]):
        got = np.zeros_like(want)
        O.me_oracle(oracle, c["org"], pyr, (c["w"] + 3) // 4, (c["h"] + 3) // 4, c["tile"], c["bd"], c["lam"], got,
                    prev, allow_hp=c["hp"], allow_full_search=c["full"], me_range_scale=c["scale"])
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (name, k, len(bad), bad[:4], got[tuple(bad[0])], want[tuple(bad[0])])


@pytest.mark.parametrize("name", [n for n in CASES if n + "_blk" in REF.files])
def test_rdo_time_estimate_motion_equals_the_executed_reference(oracle, name):
    c = load_case(name)
    pyr, prev, stats = c["refs"][0]
    blk = REF[name + "_blk"]
    cands = np.zeros(len(blk), O.ME_BLOCK_CAND)
    cands["bx"], cands["by"], cands["w"], cands["h"], cands["corner"] = blk[:, 0], blk[:, 1], blk[:, 2], blk[:, 3], blk[:, 4]
    cands["pmv"] = blk[:, 5:9].reshape(-1, 2, 2)
    use_satd, fmode = [int(v) for v in REF[name + "_blkcfg"]]
    got = O.me_block_oracle(oracle, c["org"], pyr, (c["w"] + 3) // 4, (c["h"] + 3) // 4, c["tile"], c["bd"], c["lam"],
                            stats.copy(), prev, cands, use_satd=use_satd, filter_mode=fmode, allow_hp=c["hp"])
    want = REF[name + "_blkout"]
    for i in range(len(blk)):
        g = (int(got["row"][i]), int(got["col"][i]), int(got["sad"][i]), int(got["cost"][i]))
        assert g == tuple(int(v) for v in want[i]), (name, i, blk[i], g, want[i])
