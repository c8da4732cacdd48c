"""The lookahead cost maps and the block-importance propagation of the oracle against
tests/golden/lookahead_ref.npz -- vectors produced by executing the reference's own text
(src/api/lookahead.rs:30-267, src/api/internal.rs:912-1068; tests/golden/gen_lookahead_ref.py)."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "lookahead_ref.npz")


def cases():
    G = np.load(GOLD)
    for k in G["keys"]:
        bd, w, h, _ = k.split("_")
        yield G, str(k), int(bd), int(w), int(h)


def planes(G, k, bd):
    org = O.plane_from_image(G["org_" + k], bd, 16, 16)
    ref = O.plane_from_image(G["ref_" + k], bd, 16, 16)
    # the generator's reference plane: the same edge replication
    assert np.array_equal(ref.data[ref.yorigin - 16:ref.yorigin + ref.height + 16,
                                   ref.xorigin - 16:ref.xorigin + ref.width + 16], G["refpad_" + k])
    return org, ref


def test_lookahead_maps_reproduce_the_executed_reference(oracle):
    n = 0
    for G, k, bd, w, h in cases():
        org, ref = planes(G, k, bd)
        po, pr = org.cstruct(), ref.cstruct()
        hb, wb = h // 8, w // 8
        intra = np.zeros(hb * wb, np.uint32)
        oracle.r1o_estimate_intra_costs(C.byref(po), bd, O.ptr(intra))
        assert np.array_equal(intra.reshape(hb, wb), G["intra_" + k]), k
        tot = oracle.r1o_importance_block_difference(C.byref(po), C.byref(pr))
        assert tot / (wb * hb) == float(G["blockdiff_" + k][0]), k            # `as f64 / as f64`
        mvs = np.ascontiguousarray(G["mv_" + k])
        inter = np.zeros(hb * wb, np.uint32)
        oracle.r1o_estimate_inter_costs(C.byref(po), C.byref(pr), O.ptr(mvs), O.ptr(inter))
        assert int(inter.astype(np.uint64).sum()) / (wb * hb) == float(G["inter_mean_" + k][0]), k
        fut = np.ascontiguousarray(G["future_" + k])
        for ln in (1, 4):
            acc = np.ascontiguousarray(G["refimp_in_%d_%s" % (ln, k)]).copy()
            oracle.r1o_update_block_importances(O.ptr(intra), O.ptr(fut), O.ptr(inter), O.ptr(mvs), wb, hb,
                                                ln, O.ptr(acc))
            want = G["refimp_out_%d_%s" % (ln, k)]
            assert np.array_equal(acc.view(np.uint32), want.view(np.uint32).ravel().reshape(acc.shape)), (k, ln)
            n += 1
    assert n == 12


def test_update_block_importances_adversarial_positions(oracle):
    """the adversarial set (gen_lookahead_ref.py): reference positions on / either side of
    importance-block boundaries, negative positions (-1, -63, -64, -65, -127, -128), positions at and
    beyond the right / bottom edge, len in {1, 2, 7}; intra costs given, inter costs = get_satd"""
    G = np.load(GOLD)
    n = 0
    for k in G["adv_keys"]:
        k = str(k)
        bd, w, h = [int(v) for v in k.split("_")[:3]]
        org, ref = planes(G, k, bd)
        po, pr = org.cstruct(), ref.cstruct()
        hb, wb = h // 8, w // 8
        mvs = np.ascontiguousarray(G["mv_" + k])
        # the set must contain what it claims: negative and beyond-the-edge reference positions
        rx = np.arange(wb)[None, :] * 64 + mvs[:, :, 1].astype(np.int64)
        ry = np.arange(hb)[:, None] * 64 + mvs[:, :, 0].astype(np.int64)
        assert {-1, -64}.issubset(set(rx.ravel()) | set(ry.ravel())) and rx.max() >= wb * 64 - 63 and ry.max() >= hb * 64 - 63
        inter = np.zeros(hb * wb, np.uint32)
        oracle.r1o_estimate_inter_costs(C.byref(po), C.byref(pr), O.ptr(mvs), O.ptr(inter))
        intra = np.ascontiguousarray(G["intra_" + k]).ravel().copy()
        fut = np.ascontiguousarray(G["future_" + k])
        for ln in (1, 2, 7):
            acc = np.ascontiguousarray(G["refimp_in_%d_%s" % (ln, k)]).copy()
            oracle.r1o_update_block_importances(O.ptr(intra), O.ptr(fut), O.ptr(inter), O.ptr(mvs), wb, hb,
                                                ln, O.ptr(acc))
            want = G["refimp_out_%d_%s" % (ln, k)]
            assert np.array_equal(acc.view(np.uint32), want.view(np.uint32).ravel().reshape(acc.shape)), (k, ln)
            n += 1
    assert n == 18


def test_search_to_cost_to_importance_chain_reproduces_the_reference(oracle):
    """lookahead_chain_ref.npz: the reference's motion search (me_ref.npz statistics, executed) feeding
    its cost loop and update_block_importances (executed by gen_lookahead_chain_ref.py).  The oracle runs
    the WHOLE chain itself -- its own search, its statistics into its cost loop -- and must land on the
    same mean inter cost and importances."""
    import test_oracle_me_ref as MR
    CH = np.load(os.path.join(os.path.dirname(__file__), "golden", "lookahead_chain_ref.npz"))
    n = 0
    for name in CH["keys"]:
        name = str(name)
        c = MR.load_case(name)
        pyr, prev, want = c["refs"][0]
        stats = np.zeros_like(want)
        O.me_oracle(oracle, c["org"], pyr, (c["w"] + 3) // 4, (c["h"] + 3) // 4, c["tile"], c["bd"], c["lam"], stats,
                    prev, allow_hp=c["hp"], allow_full_search=c["full"], me_range_scale=c["scale"])
        hb, wb = c["h"] // 8, c["w"] // 8
        # the importance blocks' vectors: every second MEStats entry in both directions (lookahead.rs:236-244)
        mvs = np.ascontiguousarray(np.stack([stats["row"][0:2 * hb:2, 0:2 * wb:2], stats["col"][0:2 * hb:2, 0:2 * wb:2]],
                                            axis=-1).astype(np.int16))
        po, pr = c["org"][0].cstruct(), pyr[0].cstruct()
        inter = np.zeros(hb * wb, np.uint32)
        oracle.r1o_estimate_inter_costs(C.byref(po), C.byref(pr), O.ptr(mvs), O.ptr(inter))
        assert int(inter.astype(np.uint64).sum()) / (wb * hb) == float(CH["inter_mean_" + name][0]), name
        intra = np.zeros(hb * wb, np.uint32)
        oracle.r1o_estimate_intra_costs(C.byref(po), c["bd"], O.ptr(intra))
        assert np.array_equal(intra.reshape(hb, wb), CH["intra_" + name]), name
        fut = np.ascontiguousarray(CH["future_" + name])
        for ln in (1, 3):
            acc = np.ascontiguousarray(CH["imp_in_%d_%s" % (ln, name)]).copy()
            oracle.r1o_update_block_importances(O.ptr(intra), O.ptr(fut), O.ptr(inter), O.ptr(mvs), wb, hb, ln, O.ptr(acc))
            assert np.array_equal(acc.view(np.uint32), CH["imp_out_%d_%s" % (ln, name)].view(np.uint32)), (name, ln)
            n += 1
    assert n == 16
