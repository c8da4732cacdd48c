"""The transform-type search (rdo_tx_type_decision, src/rdo.rs:1701-1817) and compute_distortion's
chroma leg: the CPU oracle and the host glue against rdo_txsearch_ref.npz -- vectors produced by
executing the reference's own text (tests/golden/gen_rdo_txsearch_ref.py)."""
import ctypes as C

import numpy as np

import oracle_lib as O
import rdo_glue_cases as RC


def test_tx_type_mask_is_the_loop_filter_of_the_executed_reference(oracle):
    """av1_tx_used[get_tx_set(..)] & RAV1E_TX_TYPES for every TxSize x is_inter x use_reduced_set: the
    oracle's restatement, the product's host function (no device work) and the glue's slot order"""
    from rav1e_amd import _lib
    from rav1e_amd import rdo_glue as RG
    G = np.load(RC.GOLD_TXSEARCH)
    L = _lib.load()
    assert tuple(int(t) for t in G["rav1e_tx_types"]) == RG.RAV1E_TX_TYPES
    for ts in range(19):
        for inter in (0, 1):
            for red in (0, 1):
                for allt in (0, 1):
                    want = int(G["ts_mask"][ts, inter, red, allt])
                    assert int(oracle.r1o_tx_type_mask(ts, inter, red, 1 - allt)) == want, (ts, inter, red, allt)
                    assert int(L.r1_tx_type_mask(ts, inter, red, 1 - allt)) == want, (ts, inter, red, allt)
                # the slot order of the fan-out = the loop order of rdo_tx_type_decision over RAV1E_TX_TYPES
                m = int(G["ts_mask"][ts, inter, red, 0])
                assert RG.tx_type_slots(m) == [t for t in RG.RAV1E_TX_TYPES if (m >> t) & 1]
    assert int(L.r1_tx_type_mask(19, 0, 0, 1)) == 0 and int(L.r1_tx_type_mask(-1, 0, 0, 1)) == 0


def oracle_dist_scaled(oracle):
    def dist_scaled(kind, bd, src, rec, x, y, vw, vh, grid):
        pa, pb = src.cstruct(), rec.cstruct()
        c = np.zeros(1, O.DIST_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"] = x, y, x, y
        out = np.zeros(1, np.uint64)
        sc = None if grid is None else O.ptr(grid)
        assert oracle.r1o_dist_scaled_batch(kind, C.byref(pa), C.byref(pb), vw, vh, O.ptr(c), 1, sc,
                                            0 if grid is None else grid.shape[1], 0, 0, O.ptr(out)) == 0
        return int(out[0])
    return dist_scaled


def test_type_search_every_visited_type_on_one_prediction(oracle):
    """r1o_rdo_txsearch_batch: eob, quantized coefficients, reconstruction and the four distortions
    of every TxType the loop visits, bit depths 8 / 10 / 12, inter and intra quantizer offsets, grid
    phases, blocks cut by the frame edge (distortion over the visible part through the reconstruction)"""
    G = np.load(RC.GOLD_TXSEARCH)

    def txsearch(bd, ts, mask, qidx, is_intra, src, pred, ox, oy, kind, grid):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        hbd = int(bd > 8)
        nt = bin(mask).count("1")
        c = np.zeros(1, O.RDO_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"], c["tx_type"] = ox, oy, ox, oy, 15   # the field is ignored
        pa, pb = src.cstruct(), pred.cstruct()
        eob, dist = np.zeros(nt, np.uint16), np.zeros(nt, np.uint64)
        qc = np.zeros((nt, min(w, 32) * min(h, 32)), np.int32 if hbd else np.int16)
        rec = np.zeros((nt, h, w), np.uint16 if hbd else np.uint8)
        sc = None if grid is None else O.ptr(grid)
        assert oracle.r1o_rdo_txsearch_batch(C.byref(pa), C.byref(pb), None, w, h, ts, O.ptr(c), 1, mask, qidx,
                                             is_intra, 0, 0, kind, sc, 0 if grid is None else grid.shape[1], 0, 0,
                                             None, None, O.ptr(eob), O.ptr(dist), None, O.ptr(qc), O.ptr(rec)) == 0
        return eob, dist, qc, rec
    n = RC.check_txsearch(G, txsearch, oracle_dist_scaled(oracle))
    assert n == sum(len(G["tsr_types_" + str(k)]) for k in G["tsr_keys"]) * 4 and n > 5000


def test_next_transform_depth_of_an_inter_block_is_the_type_search_on_its_quadrants(oracle):
    """write_tx_tree with tx_size < bsize, executed whole (90 cases: 16x16 -> 4 x 8x8 ... 64x64 -> 4 x 32x32, rectangles,
    8x8 -> 4 x 4x4, bit depths 8 / 10 / 12, blocks cut by the frame edge), against r1o_rdo_txsearch_batch in its
    dense-prediction form on the transform blocks rdo_glue.tx_split_blocks lists"""
    G = np.load(RC.GOLD_TXSEARCH)

    def txsearch_pred(bd, ts, mask, qidx, src, preds, pos, kind, grid):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        hbd = int(bd > 8)
        nt, n = bin(mask).count("1"), len(pos)
        c = np.zeros(n, O.RDO_CAND)
        c["ox"], c["oy"] = [p[0] for p in pos], [p[1] for p in pos]
        pa = src.cstruct()
        pr = np.ascontiguousarray(preds.astype(np.uint16 if hbd else np.uint8))
        eob, dist = np.zeros((n, nt), np.uint16), np.zeros((n, nt), np.uint64)
        qc = np.zeros((n, nt, min(w, 32) * min(h, 32)), np.int32 if hbd else np.int16)
        rec = np.zeros((n, nt, h, w), np.uint16 if hbd else np.uint8)
        sc = None if grid is None else O.ptr(grid)
        assert oracle.r1o_rdo_txsearch_batch(C.byref(pa), None, O.ptr(pr), w, h, ts, O.ptr(c), n, mask, qidx, 0, 0, 0, kind, sc,
                                             0 if grid is None else grid.shape[1], 0, 0, None, None, O.ptr(eob), O.ptr(dist),
                                             None, O.ptr(qc), O.ptr(rec)) == 0
        return eob, dist, qc, rec
    n = RC.check_txsplit(G, txsearch_pred, oracle_dist_scaled(oracle))
    assert n == sum(len(G["txs_types_" + str(k)]) for k in G["txs_keys"]) * 4 and n > 1500


def test_compute_distortion_with_chroma(oracle):
    """compute_distortion (src/rdo.rs:254-347) with is_chroma_block and !luma_only on 4:2:0 / 4:2:2 /
    4:4:4 planes: rav1e_amd.rdo_glue.compute_distortion over the oracle's sse_wxh / cdef_dist_wxh"""
    G = np.load(RC.GOLD_TXSEARCH)

    def make_dist(bd, srcs, recs, grid, xdec, ydec):
        def dist_wxh(kind, p, x, y, w, h):
            pa, pb = srcs[p].cstruct(), recs[p].cstruct()
            c = np.zeros(1, O.DIST_CAND)
            c["ox"], c["oy"], c["rx"], c["ry"] = x, y, x, y
            out = np.zeros(1, np.uint64)
            xd, yd = (xdec, ydec) if p else (0, 0)
            sc = None if grid is None else O.ptr(grid)
            assert oracle.r1o_dist_scaled_batch(kind, C.byref(pa), C.byref(pb), w, h, O.ptr(c), 1, sc,
                                                0 if grid is None else grid.shape[1], xd, yd, O.ptr(out)) == 0
            return int(out[0])
        return dist_wxh
    assert RC.check_compute_distortion(G, make_dist) == 5 * 104


def test_restoration_geometry_is_the_executed_restoration_state():
    """RestorationState::new (src/lrf.rs:1321-1480) executed for 260-odd frame configurations (sizes, chroma
    samplings, quantizers on both sides of its thresholds, large / small units, 128x128 superblocks, tilings, random
    sizes): rav1e_amd.rdo_glue.restoration_plane_configs reproduces unit size, sb shifts, stripe height and the
    last-unit rule of cols / rows of all three planes, and restoration_search_units the visible size of the last
    unit column / row (rdo.rs:2645-2654) -- the geometry the restoration search's unit lists are built from
    (tools/frame_stages.py, tools/bench_lrf_search.py) and that gen_lrf_search_ref.py used to state by hand."""
    import os
    from rav1e_amd import rdo_glue as RG
    G = np.load(os.path.join(os.path.dirname(RC.GOLD), "lrf_geometry_ref.npz"))
    n = 0
    for inp, cfg, vis in zip(G["geo_in"], G["geo_cfg"], G["geo_vis"]):
        w, h, xd, yd, q, large, rest, sb128, tc, tr, tw, th = [int(v) for v in inp]
        if (cfg[:, 1:3] < 0).any():
            continue      # a unit smaller than the superblock: the reference's own usize subtraction underflows there
        got = RG.restoration_plane_configs(w, h, xd, yd, q, bool(large), bool(rest), bool(sb128), (tc, tr, tw, th))
        for pli in range(3):
            g = got[pli]
            assert [g["unit_size"], g["sb_h_shift"], g["sb_v_shift"], g["stripe_height"], g["cols"], g["rows"]] == \
                [int(v) for v in cfg[pli]], (tuple(inp), pli, g, cfg[pli])
            dx, dy = (xd, yd) if pli else (0, 0)
            units = RG.restoration_search_units(g, w, h, dx, dy)
            if units:
                assert max(u[0] for u in units) // g["unit_size"] <= g["cols"] - 1
                last_col = max(u[0] for u in units)
                last_row = max(u[1] for u in units)
                vw = [u[2] for u in units if u[0] == last_col][0]
                vh = [u[3] for u in units if u[1] == last_row][0]
                assert (vw, vh) == (int(vis[pli][0]), int(vis[pli][1])), (tuple(inp), pli, vw, vh, vis[pli])
                # the list is a partition of what it covers: no unit wider than the nominal size, none empty
                assert all(0 < u[2] <= g["unit_size"] and 0 < u[3] <= g["unit_size"] for u in units)
        n += 1
    assert n > 200
    # the 4K frame of BASELINE config 4 at qindex 100: 64-pixel luma units, 32-pixel chroma units (what the benches use)
    c4 = RG.restoration_plane_configs(3840, 2160, 1, 1, 100)
    assert (c4[0]["unit_size"], c4[1]["unit_size"], c4[0]["cols"], c4[0]["rows"]) == (64, 32, 60, 34)
    assert len(RG.restoration_search_units(c4[0], 3840, 2160)) == 60 * 34
