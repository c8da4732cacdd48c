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
