"""GPU parity against REFERENCE-DERIVED vectors: tests/golden/*_ref.npz hold outputs
computed by executing the reference's own Rust source text (tools/rustlite,
tests/golden/gen_*_ref.py).  The HIP kernels are called through the C ABI on the
same inputs and must reproduce them bit for bit -- no oracle in between."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dev_plane(hp):
    from rav1e_amd.api import Plane
    return Plane.from_numpy(hp.data, hp.width, hp.height, hp.bit_depth, hp.xpad, hp.ypad)


def block_plane(arr, bd, pad):
    h, w = arr.shape
    hp = O.HostPlane(w, h, bd, pad, pad)
    hp.view()[:] = arr
    return hp


def test_dist_ref_sad_satd(ctx):
    G = np.load(os.path.join(GOLD, "dist_ref.npz"))
    c = np.zeros(1, O.DIST_CAND)
    for i, k in enumerate(G["d_keys"]):
        bd, w, h, _ = map(int, k.split("_"))
        if (w & (w - 1)) or (h & (h - 1)):
            continue    # the batch entry point takes the 22 BlockSizes; odd sizes stay on the host path
        a, b = block_plane(G["d_org_" + k], bd, 8), block_plane(G["d_ref_" + k], bd, 16)
        da, db = dev_plane(a), dev_plane(b)
        sad = ctx.dist_batch(0, da, db, w, h, c).cpu().numpy().view(np.uint32)[0]
        satd = ctx.dist_batch(1, da, db, w, h, c).cpu().numpy().view(np.uint32)[0]
        assert (sad, satd) == (G["d_sad"][i], G["d_satd"][i]), k


def test_dist_ref_cdef_dist_kernel(ctx):
    """cdef_dist_kernel for w, h in {4, 8}: cdef_dist_wxh of one kernel with the default
    DistortionScale (1 << 14) is the kernel value itself."""
    G = np.load(os.path.join(GOLD, "dist_ref.npz"))
    c = np.zeros(1, O.DIST_CAND)
    for i, k in enumerate(G["k_keys"]):
        bd, w, h, _ = map(int, k.split("_"))
        if w % 4 or h % 4:
            continue    # r1_dist_scaled_batch takes multiples of 4 (coded frame sizes are padded to 8)
        a, b = block_plane(G["k_org_" + k], bd, 8), block_plane(G["k_ref_" + k], bd, 8)
        got = ctx.dist_scaled_batch(3, dev_plane(a), dev_plane(b), w, h, c).cpu().numpy().view(np.uint64)[0]
        assert got == G["k_out"][i], k


def test_dist_ref_wxh_glue_on_planes(ctx):
    """cdef_dist_wxh / sse_wxh + distortion_scale() lookups, luma and 4:2:0 chroma."""
    import torch
    G = np.load(os.path.join(GOLD, "dist_ref.npz"))
    cache = {}
    for k in G["f_keys"]:
        bd, kind, w, h, xdec, use_grid = map(int, k.split("_"))
        if (bd, xdec) not in cache:
            org, ref = G["f_org_%d" % bd], G["f_ref_%d" % bd]
            H, W = org.shape
            a = block_plane(org[:H >> xdec, :W >> xdec], bd, 16)
            b = block_plane(ref[:H >> xdec, :W >> xdec], bd, 24)
            sc = torch.from_numpy(np.ascontiguousarray(G["f_scales_%d" % bd]).view(np.int32)).cuda()
            cache[(bd, xdec)] = (dev_plane(a), dev_plane(b), sc)
        da, db, sc = cache[(bd, xdec)]
        cands = G["f_cands_" + k]
        c = np.zeros(len(cands), O.DIST_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"] = cands.T
        got = ctx.dist_scaled_batch(kind, da, db, w, h, c, sc if use_grid else None, xdec, xdec)
        assert np.array_equal(got.cpu().numpy().view(np.uint64), G["f_out_" + k]), k


def test_predict_ref_intra_edges(ctx):
    """get_intra_edges vectors of the reference's own text (src/partition.rs:639-898)."""
    from rav1e_amd.api import INTRA_EDGE_CAND, Plane
    Z = np.load(os.path.join(GOLD, "predict_ref.npz"))
    G = {k: Z[k] for k in Z.files if k.startswith("e_")}
    cases = np.unique(G["e_case"])
    for cs in cases:
        idx = np.nonzero(G["e_case"] == cs)[0]
        bd = int(G["e_bd"][idx[0]])
        rect_w, rect_h = int(G["e_rect_w"][idx[0]]), int(G["e_rect_h"][idx[0]])
        tile = G["e_tile_%d" % cs]
        th, tw = tile.shape
        dt = np.uint16 if bd > 8 else np.uint8
        # plane whose visible size is the tile's visible rectangle; the rest of the tile
        # array (pixels past the frame edge) lands in the padding
        hp = O.HostPlane(rect_w, rect_h, bd, 96, 48)
        hp.data[hp.yorigin:hp.yorigin + th, hp.xorigin:hp.xorigin + tw] = tile.astype(dt)
        dp = Plane.from_numpy(hp.data, rect_w, rect_h, bd, 96, 48)
        for ts in np.unique(G["e_ts"][idx]):
            sub = idx[G["e_ts"][idx] == ts]
            ec = np.zeros(len(sub), INTRA_EDGE_CAND)
            ec["x"], ec["y"] = G["e_x"][sub], G["e_y"][sub]
            ec["mode"], ec["angle_delta"] = G["e_mode"][sub], G["e_angle_delta"][sub]
            ec["flags"] = G["e_enable_ief"][sub] | (G["e_has_tr"][sub] << 1) | (G["e_has_bl"][sub] << 2)
            edges, lens = ctx.intra_edges_batch(dp, (0, 0, tw, th), int(ts), ec)
            ge, gl = edges.cpu().numpy().view(dt), lens.cpu().numpy()
            for k, i in enumerate(sub):
                ll, al = int(G["e_left_len"][i]), int(G["e_above_len"][i])
                assert tuple(gl[k]) == (ll, al), (cs, i)
                assert np.array_equal(ge[k, 128 - ll:129 + al].astype(np.uint16),
                                      G["e_edges"][i][128 - ll:129 + al]), (cs, i)


def test_predict_ref_cfl_ac(ctx):
    """pred_cfl_ac vectors of the reference's own text (src/predict.rs:1020-1063)."""
    from rav1e_amd.api import CFL_AC_CAND
    Z = np.load(os.path.join(GOLD, "predict_ref.npz"))
    G = {k: Z[k] for k in Z.files if k.startswith("a_")}
    for i in range(len(G["a_bd"])):
        bd, bw, bh = int(G["a_bd"][i]), int(G["a_bw"][i]), int(G["a_bh"][i])
        xdec, ydec = int(G["a_xdec"][i]), int(G["a_ydec"][i])
        lw, lh = max(bw << xdec, 8), max(bh << ydec, 8)
        luma = G["a_luma"][G["a_luma_off"][i]:G["a_luma_off"][i + 1]].reshape(lh, lw)
        hp = block_plane(luma.astype(np.uint16 if bd > 8 else np.uint8), bd, 16)
        c = np.zeros(1, CFL_AC_CAND)
        c["w_pad"], c["h_pad"] = int(G["a_w_pad"][i]), int(G["a_h_pad"][i])
        got = ctx.cfl_ac_batch(dev_plane(hp), bw, bh, xdec, ydec, c).cpu().numpy()[0]
        off = int(G["a_off"][i])
        assert np.array_equal(got.ravel(), G["a_out"][off:off + bw * bh]), i


def test_activity_ref_scales(ctx):
    """ActivityMask::from_plane + fill_scales vectors of the reference's own text."""
    A = np.load(os.path.join(GOLD, "activity_ref.npz"))
    for k in A["keys"]:
        bd, w, h = map(int, k.split("_"))
        hp = O.plane_from_image(A["img_" + k], bd, 16, 16)
        var, sc = ctx.activity_scales(dev_plane(hp))
        assert np.array_equal(var.cpu().numpy().view(np.uint32), A["var_" + k]), k
        assert np.array_equal(sc.cpu().numpy().view(np.uint32), A["scale_" + k]), k


def test_comm_c_abi_single_rank(ctx):
    """csrc/comm.hip through the C ABI on one GPU: RCCL initialises (world 1), the tile
    all-gather packs / gathers / leaves the plane intact, a halo plan without neighbours is a
    no-op.  (Multi-rank geometry: tests/test_distributed.py on gloo; the 8-GPU run is the driver's.)"""
    import torch
    from rav1e_amd import tiles
    from rav1e_amd import workload as W
    hp = O.HostPlane(640, 384, 8, rng=np.random.default_rng(5))
    dp = dev_plane(hp)
    comm = tiles.Comm(ctx, 0, 1)
    rects = W.tile_rects(1, 640, 384)
    before = dp.data.clone()
    comm.allgather_tiles(dp, rects)
    assert comm.exchange_tile_halos(dp, rects) == 0
    send = torch.arange(4096, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    comm.allgather(send, recv)
    torch.cuda.synchronize()
    assert torch.equal(before, dp.data) and torch.equal(send, recv)
    # the peer-store entry points through the communicator (world 1: no peers, the stores and the hand-shake are
    # no-ops, the argument checks and the plumbing are not): r1_comm_open_peer_planes, r1_comm_push_frame, the ring
    pp = [tiles.PeerPlanes(ctx, dp, 0, 1, comm=comm), tiles.PeerPlanes(ctx, dev_plane(hp), 0, 1, comm=comm)]
    pp[0].push_frame(rects)
    ring = tiles.TileRing([pp[0].plane, pp[1].plane], pp, rects, 0, tiles.visible(dp).clone())
    assert ring.check()
    for _ in range(5):
        ring.advance()
        assert ring.check()
    torch.cuda.synchronize()
    for p in pp:
        p.close()
    comm.close()


def test_lookahead_ref_maps_and_importances(ctx):
    """N1 against the executed reference text (lookahead_ref.npz: estimate_intra_costs,
    estimate_importance_block_difference, the cost loop of estimate_inter_costs,
    update_block_importances -- src/api/lookahead.rs:30-267, src/api/internal.rs:912-1068)."""
    import torch
    G = np.load(os.path.join(GOLD, "lookahead_ref.npz"))
    for k in G["keys"]:
        k = str(k)
        bd, w, h = [int(v) for v in k.split("_")[:3]]
        hb, wb = h // 8, w // 8
        org, ref = O.plane_from_image(G["org_" + k], bd, 16, 16), O.plane_from_image(G["ref_" + k], bd, 16, 16)
        do, dr = dev_plane(org), dev_plane(ref)
        intra = ctx.estimate_intra_costs(do)
        assert np.array_equal(intra.cpu().numpy().view(np.uint32), G["intra_" + k]), k
        assert ctx.importance_block_difference(do, dr) == float(G["blockdiff_" + k][0]), k
        mvs = torch.from_numpy(np.ascontiguousarray(G["mv_" + k])).cuda()
        inter = ctx.estimate_inter_costs(do, dr, mvs)
        tot = int(inter.cpu().numpy().view(np.uint32).astype(np.uint64).sum())
        assert tot / (wb * hb) == float(G["inter_mean_" + k][0]), k
        fut = torch.from_numpy(np.ascontiguousarray(G["future_" + k])).cuda()
        for ln in (1, 4):
            acc = torch.from_numpy(np.ascontiguousarray(G["refimp_in_%d_%s" % (ln, k)]).copy()).cuda()
            ctx.update_block_importances(intra.reshape(-1), fut.reshape(-1), inter.reshape(-1),
                                         mvs.reshape(-1, 2), wb, hb, ln, acc.reshape(-1))
            want = G["refimp_out_%d_%s" % (ln, k)]
            assert np.array_equal(acc.cpu().numpy().view(np.uint32), want.view(np.uint32)), (k, ln)


def test_lookahead_ref_adversarial_importance_positions(ctx):
    """update_block_importances on the adversarial set of lookahead_ref.npz (negative / boundary /
    beyond-the-edge reference positions, len in {1, 2, 7}); see test_oracle_lookahead_ref.py"""
    import torch
    G = np.load(os.path.join(GOLD, "lookahead_ref.npz"))
    n = 0
    for k in G["adv_keys"]:
        k = str(k)
        bd, w, h = [int(v) for v in k.split("_")[:3]]
        hb, wb = h // 8, w // 8
        org, ref = O.plane_from_image(G["org_" + k], bd, 16, 16), O.plane_from_image(G["ref_" + k], bd, 16, 16)
        do, dr = dev_plane(org), dev_plane(ref)
        mvs = torch.from_numpy(np.ascontiguousarray(G["mv_" + k])).cuda()
        inter = ctx.estimate_inter_costs(do, dr, mvs)
        intra = torch.from_numpy(np.ascontiguousarray(G["intra_" + k]).view(np.int32)).cuda()
        fut = torch.from_numpy(np.ascontiguousarray(G["future_" + k])).cuda()
        for ln in (1, 2, 7):
            acc = torch.from_numpy(np.ascontiguousarray(G["refimp_in_%d_%s" % (ln, k)]).copy()).cuda()
            ctx.update_block_importances(intra.reshape(-1), fut.reshape(-1), inter.reshape(-1),
                                         mvs.reshape(-1, 2), wb, hb, ln, acc.reshape(-1))
            want = G["refimp_out_%d_%s" % (ln, k)]
            assert np.array_equal(acc.cpu().numpy().view(np.uint32), want.view(np.uint32)), (k, ln)
            n += 1
    assert n == 18


# ---- a13: inverse_transform_add, the WHOLE function executed from the text (gen_inv_tx_ref.py) --
def test_inv_tx_ref_whole_function(ctx):
    """480 (tx_size, tx_type, bit depth) cases x 3-5 blocks; the 2-D driver of
    src/transform/inverse.rs:1633-1705 was executed as written (no hand-stated driver)."""
    import torch
    G = np.load(os.path.join(GOLD, "inv_tx_ref.npz"))
    keys = [k for k in G.files if k.endswith("_co")]
    assert len(keys) == 480
    for k in keys:
        _, ts, tt, bd, _ = k.split("_")
        ts, tt, bd = int(ts), int(tt), int(bd)
        co, pred, rec = G[k], G[k[:-3] + "_pred"], G[k[:-3] + "_rec"]
        dp = pred if bd == 8 else pred.view(np.int16)
        got = ctx.inverse_transform_add_batch(torch.from_numpy(co).cuda(), torch.from_numpy(dp).cuda(), ts, tt, bd)
        assert np.array_equal(got.cpu().numpy().view(pred.dtype), rec), k


# ---- N4 glue against the executed reference text (gen_rdo_glue_ref.py) ----------------------
def test_rdo_glue_ref_tx_block_rate_and_distortion(ctx):
    """encode_tx_block's TxDistEstRate evaluation (src/encoder.rs:1404-1661) as executed from the
    reference's text: r1_rdo_full_cand_batch on the zero-motion candidate reproduces the
    transform-domain distortion and estimate_rate's value of every case."""
    import rdo_glue_cases as RC
    from rav1e_amd.api import RDO_CAND
    G = np.load(RC.GOLD)

    def full_cand(bd, ts, tt, qidx, src, pred):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        c = np.zeros(1, RDO_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"], c["tx_type"] = 8, 8, 8, 8, tt
        o = ctx.rdo_full_cand_batch(dev_plane(src), dev_plane(pred), w, h, c, qidx, is_intra=0)
        return int(o["tx_dist"].cpu().numpy().view(np.uint64)[0]), int(o["est_rate"].cpu().numpy().view(np.uint64)[0])
    assert RC.check_tx_blocks(G, full_cand) == 156


def test_rdo_pixel_ref_encode_tx_block_pixel_leg(ctx):
    """rav1e's default tune (BASELINE config 4): encode_tx_block with
    use_tx_domain_distortion = false (dequantize -> inverse_transform_add into the reconstruction,
    src/encoder.rs:1588-1614) followed by compute_distortion (src/rdo.rs:254-347), executed from
    the reference's text (gen_rdo_pixel_ref.py): r1_rdo_pixel_cand_batch reproduces eob, the
    quantized coefficients, the reconstruction and the distortion (sse_wxh and cdef_dist_wxh,
    with and without the temporal-RDO scale grid) of all 482 blocks."""
    import torch
    import rdo_glue_cases as RC
    from rav1e_amd.api import RDO_CAND
    G = np.load(RC.GOLD_PIXEL)
    cache = {}

    def pixel_cand(bd, ts, tt, qidx, src, pred, kind, scales, stride):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        if cache.get("key") != (id(src), id(pred)):
            cache.update(key=(id(src), id(pred)), planes=(dev_plane(src), dev_plane(pred)), hold=(src, pred))
        ds, dp = cache["planes"]
        c = np.zeros(1, RDO_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"], c["tx_type"] = 8, 8, 8, 8, tt
        sc = None if scales is None else torch.from_numpy(scales.view(np.int32)).cuda()
        o = ctx.rdo_pixel_cand_batch(ds, dp, w, h, c, qidx, kind, scales=sc, is_intra=0, want_qcoeffs=True,
                                     want_rec=True)
        pt = np.uint8 if bd == 8 else np.uint16
        return (int(o["eob"].cpu().numpy().view(np.uint16)[0]), int(o["dist"].cpu().numpy().view(np.uint64)[0]),
                o["qcoeffs"].cpu().numpy()[0], o["rec"].cpu().numpy().view(pt)[0])
    assert RC.check_pixel_blocks(G, pixel_cand) == 482 * 4


def _gpu_dist_scaled(ctx):
    import torch
    from rav1e_amd.api import DIST_CAND

    def dist_scaled(kind, bd, src, rec, x, y, vw, vh, grid, xdec=0, ydec=0):
        c = np.zeros(1, DIST_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"] = x, y, x, y
        sc = None if grid is None else torch.from_numpy(grid.view(np.int32)).cuda()
        return int(ctx.dist_scaled_batch(kind, dev_plane(src), dev_plane(rec), vw, vh, c, sc, xdec, ydec)
                   .cpu().numpy().view(np.uint64)[0])
    return dist_scaled


def test_rdo_txsearch_ref_every_visited_type_on_one_prediction(ctx):
    """rdo_tx_type_decision's loop body as executed from the reference's text (gen_rdo_txsearch_ref.py):
    r1_rdo_txsearch_batch in ONE launch reproduces eob, quantized coefficients, reconstruction and the
    four distortions (sse_wxh / cdef_dist_wxh, with and without the temporal-RDO grid) of EVERY TxType the
    loop visits (RAV1E_TX_TYPES cut by av1_tx_used[get_tx_set]), bit depths 8 / 10 / 12, intra and inter
    quantizer offsets, moving grid phases; for blocks cut by the frame edge (frame 102 x 78) the visible-
    part distortion of compute_distortion comes from the reconstruction through r1_dist_scaled_batch."""
    import torch
    import rdo_glue_cases as RC
    from rav1e_amd.api import RDO_CAND
    G = np.load(RC.GOLD_TXSEARCH)
    cache = {}

    def txsearch(bd, ts, mask, qidx, is_intra, src, pred, ox, oy, kind, grid):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        if bd not in cache:
            cache[bd] = (dev_plane(src), dev_plane(pred), torch.from_numpy(grid_of[bd].view(np.int32)).cuda())
        ds, dp, dg = cache[bd]
        c = np.zeros(1, RDO_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"], c["tx_type"] = ox, oy, ox, oy, 14     # the field is ignored
        o = ctx.rdo_txsearch_batch(ds, dp, w, h, c, mask, qidx, kind, scales=None if grid is None else dg,
                                   is_intra=is_intra, want_qcoeffs=True, want_rec=True)
        pt = np.uint8 if bd == 8 else np.uint16
        return (o["eob"].cpu().numpy().view(np.uint16)[0], o["dist"].cpu().numpy().view(np.uint64)[0],
                o["qcoeffs"].cpu().numpy()[0], o["rec"].cpu().numpy().view(pt)[0])
    grid_of = {bd: np.ascontiguousarray(G["tsr_scales_%d" % bd]) for bd in (8, 10, 12)}
    n = RC.check_txsearch(G, txsearch, _gpu_dist_scaled(ctx))
    assert n == sum(len(G["tsr_types_" + str(k)]) for k in G["tsr_keys"]) * 4 and n > 5000


def test_rdo_txsearch_ref_next_transform_depth_of_an_inter_block(ctx):
    """rdo_tx_size_type's next depth for an inter block (src/rdo.rs:745-815) = write_tx_tree with tx_size < bsize on the
    SAME prediction, EXECUTED WHOLE in the fixture (part 4: 90 cases x every visited type): r1_rdo_txsearch_batch in its
    dense-prediction form, ONE launch per (block, distortion kind) over the block's transform blocks, reproduces eob and
    coefficients of every transform block, the block's reconstruction and its four distortions; where the transform
    blocks are whole 8x8 tiles the block's distortion is the sum of the launch's own outputs."""
    import torch
    import rdo_glue_cases as RC
    from rav1e_amd.api import RDO_CAND
    G = np.load(RC.GOLD_TXSEARCH)
    cache = {}

    def txsearch_pred(bd, ts, mask, qidx, src, preds, pos, kind, grid):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        if bd not in cache:
            cache[bd] = (dev_plane(src), torch.from_numpy(np.ascontiguousarray(G["tsr_scales_%d" % bd]).view(np.int32)).cuda())
        ds, dg = cache[bd]
        c = np.zeros(len(pos), RDO_CAND)
        c["ox"], c["oy"] = [p[0] for p in pos], [p[1] for p in pos]
        pt = np.uint8 if bd == 8 else np.uint16
        dp = torch.from_numpy(np.ascontiguousarray(preds.astype(pt)).view(np.uint8 if bd == 8 else np.int16)).cuda()
        o = ctx.rdo_txsearch_batch(ds, None, w, h, c, mask, qidx, kind, scales=None if grid is None else dg, is_intra=0,
                                   want_qcoeffs=True, want_rec=True, pred=dp)
        return (o["eob"].cpu().numpy().view(np.uint16), o["dist"].cpu().numpy().view(np.uint64), o["qcoeffs"].cpu().numpy(),
                o["rec"].cpu().numpy().view(pt))
    n = RC.check_txsplit(G, txsearch_pred, _gpu_dist_scaled(ctx))
    assert n == sum(len(G["txs_types_" + str(k)]) for k in G["txs_keys"]) * 4 and n > 1500


def test_rdo_txsearch_ref_compute_distortion_with_chroma(ctx):
    """compute_distortion (src/rdo.rs:254-347) with is_chroma_block and !luma_only as executed from the
    reference's text on 4:2:0 / 4:2:2 / 4:4:4 planes, bit depths 8 / 10 / 12: rav1e_amd.rdo_glue's
    composition over r1_dist_scaled_batch (cdef_dist_wxh / sse_wxh on luma, sse_wxh on the decimated
    chroma planes, blocks cut by the frame edge down to kernels that are not multiples of 4)."""
    import rdo_glue_cases as RC
    G = np.load(RC.GOLD_TXSEARCH)
    ds = _gpu_dist_scaled(ctx)

    def make_dist(bd, srcs, recs, grid, xdec, ydec):
        def dist_wxh(kind, p, x, y, w, h):
            xd, yd = (xdec, ydec) if p else (0, 0)
            return ds(kind, bd, srcs[p], recs[p], x, y, w, h, grid, xd, yd)
        return dist_wxh
    assert RC.check_compute_distortion(G, make_dist) == 5 * 104


def test_rdo_glue_ref_compute_tx_distortion(ctx):
    """compute_tx_distortion (src/rdo.rs:349-434): rav1e_amd.rdo_glue's composition over
    r1_dist_scaled_batch (sse_wxh)"""
    import rdo_glue_cases as RC
    from rav1e_amd.api import DIST_CAND
    G = np.load(RC.GOLD)

    def make_sse(bd, srcs, recs):
        ds, dr = [dev_plane(p) for p in srcs], [dev_plane(p) for p in recs]

        def sse(p, x, y, w, h):
            c = np.zeros(1, DIST_CAND)
            c["ox"], c["oy"], c["rx"], c["ry"] = x, y, x, y
            dec = 1 if p else 0
            return int(ctx.dist_scaled_batch(2, ds[p], dr[p], w, h, c, None, dec, dec).cpu().numpy().view(np.uint64)[0])
        return sse
    assert RC.check_compute_tx_distortion(G, make_sse) == 2 * 11 * 8


def test_rdo_glue_ref_cfl_alpha(ctx):
    """rdo_cfl_alpha (src/rdo.rs:1593-1688) as executed from the reference's text: luma_ac ->
    get_intra_edges -> the 33-alpha search, through r1_cfl_ac_batch / r1_intra_edges_batch /
    r1_cfl_alpha_search_batch with rav1e_amd.rdo_glue's descriptor arithmetic."""
    import rdo_glue_cases as RC
    from rav1e_amd.api import CFL_AC_CAND, CFL_ALPHA_CAND, INTRA_EDGE_CAND
    G = np.load(RC.GOLD)
    cache = {}

    def alpha_search(bd, xdec, ydec, srcs, recs, uv_ts, pli, cx, cy, lx, ly, w_pad, h_pad, vw, vh, variant):
        key = id(srcs)
        if key not in cache:
            cache.clear()
            cache[key] = ([dev_plane(p) for p in srcs], [dev_plane(p) for p in recs], srcs)   # srcs held: id stays unique
        ds, dr, _ = cache[key]
        tw, th = RC.TX_W[uv_ts], RC.TX_H[uv_ts]
        ec = np.zeros(1, INTRA_EDGE_CAND)
        ec["x"], ec["y"], ec["mode"], ec["flags"] = cx, cy, 13, 1
        rec = recs[pli]
        edges, lens = ctx.intra_edges_batch(dr[pli], (0, 0, rec.width, rec.height), uv_ts, ec)
        ac_c = np.zeros(1, CFL_AC_CAND)
        ac_c["x"], ac_c["y"], ac_c["w_pad"], ac_c["h_pad"] = lx, ly, w_pad, h_pad
        ac = ctx.cfl_ac_batch(dr[0], tw, th, xdec, ydec, ac_c)
        cc = np.zeros(1, CFL_ALPHA_CAND)
        cc["x"], cc["y"], cc["variant"], cc["vis_w"], cc["vis_h"] = cx, cy, variant, vw, vh
        alpha, _ = ctx.cfl_alpha_search_batch(ds[pli], uv_ts, cc, edges, lens, ac)
        return int(alpha.cpu().numpy()[0])
    assert RC.check_cfl_alpha(G, alpha_search) == 4 * 8 * 2


def test_rdo_glue_ref_predict_inter_compound(ctx):
    """predict_inter_compound (src/predict.rs:339-382): get_mv_params on the host, prep_8tap x 2
    and mc_avg on the device"""
    import rdo_glue_cases as RC
    from rav1e_amd.api import MC_CAND
    G = np.load(RC.GOLD)
    cache = {}

    def compound(bd, filt, refs, w, h, p0, p1):
        key = id(refs[0])
        if key not in cache:
            cache.clear()
            cache[key] = (refs, [dev_plane(p) for p in refs])   # holds refs: the id cannot be reused
        tmps = []
        for dp, (x, y, cf, rf) in zip(cache[key][1], (p0, p1)):
            c = np.zeros(1, MC_CAND)
            c["rx"], c["ry"], c["col_frac"], c["row_frac"], c["mode_x"], c["mode_y"] = x, y, cf, rf, filt, filt
            tmps.append(ctx.prep_8tap_batch(dp, w, h, c))
        out = ctx.mc_avg_batch(tmps[0], tmps[1], w, h, bd)
        return out.cpu().numpy().view(np.uint8 if bd == 8 else np.uint16)
    assert RC.check_compound(G, compound) == 3 * 2 * 21


# ---- N2: motion estimation against the executed src/me.rs text (gen_me_ref.py) ----------
def _me_ref_cases():
    M = np.load(os.path.join(GOLD, "me_ref.npz"))
    return sorted(k[:-5] for k in M.files if k.endswith("_meta"))


def _me_stats_tensor(a):
    import torch
    s = np.zeros(a.shape[:2], O.ME_STATS)
    s["row"], s["col"], s["normalized_sad"] = a[..., 0], a[..., 1], a[..., 2]
    return torch.from_numpy(s.view(np.int32).reshape(s.shape[0], s.shape[1], 2).copy()).cuda(), s


def test_lookahead_chain_search_to_cost_to_importance(ctx):
    """lookahead_chain_ref.npz (gen_lookahead_chain_ref.py): the reference's motion search feeding its cost
    loop and update_block_importances.  The device runs the whole chain on its own: r1_estimate_tile_motion_batch
    writes the statistics, a strided VIEW of that very buffer is what r1_estimate_inter_costs /
    r1_update_block_importances take as vectors -- no host round trip -- and the mean inter cost and the
    importances equal the reference's."""
    import torch
    M = np.load(os.path.join(GOLD, "me_ref.npz"))
    CH = np.load(os.path.join(GOLD, "lookahead_chain_ref.npz"))
    pads = (88, 44, 22)
    n = 0
    for name in CH["keys"]:
        name = str(name)
        w, h, bd, tx, ty, tw, th, hp, full, scale, n_refs, _ = [int(v) for v in M[name + "_meta"]]
        lam = [int(v) for v in M[name + "_lambda"]]
        org = [dev_plane(O.plane_from_image(M["%s_org%d" % (name, s)].astype(np.int64), bd, pads[s], pads[s]))
               for s in range(3)]
        ref = [dev_plane(O.plane_from_image(M["%s_ref0_%d" % (name, s)].astype(np.int64), bd, pads[s], pads[s]))
               for s in range(3)]
        prev_t, _ = _me_stats_tensor(M["%s_prev0" % name])
        st, _ = _me_stats_tensor(np.zeros_like(M["%s_stats0" % name]))
        cols, rows = (w + 3) // 4, (h + 3) // 4
        mode = ctx.estimate_frame_motion([dict(org=org, ref=ref, stats=st, prev=prev_t, tile=(tx, ty, tw, th))], cols, rows,
                                         bd, lam, allow_hp=bool(hp), allow_full_search=bool(full), me_range_scale=scale)
        assert mode in (0, 1)
        hb, wb = h // 8, w // 8
        # MEStats = (row | col << 16, normalized_sad): the first int32 of every second entry, as int16 pairs
        mvs = st[0:2 * hb:2, 0:2 * wb:2, 0].contiguous().view(torch.int16).reshape(hb, wb, 2)
        inter = ctx.estimate_inter_costs(org[0], ref[0], mvs)
        tot = int(inter.cpu().numpy().view(np.uint32).astype(np.uint64).sum())
        assert tot / (wb * hb) == float(CH["inter_mean_" + name][0]), name
        intra = ctx.estimate_intra_costs(org[0])
        assert np.array_equal(intra.cpu().numpy().view(np.uint32), CH["intra_" + name]), name
        fut = torch.from_numpy(np.ascontiguousarray(CH["future_" + name])).cuda()
        for ln in (1, 3):
            acc = torch.from_numpy(np.ascontiguousarray(CH["imp_in_%d_%s" % (ln, name)]).copy()).cuda()
            ctx.update_block_importances(intra.reshape(-1), fut.reshape(-1), inter.reshape(-1), mvs.reshape(-1, 2),
                                         wb, hb, ln, acc.reshape(-1))
            assert np.array_equal(acc.cpu().numpy().view(np.uint32), CH["imp_out_%d_%s" % (ln, name)].view(np.uint32)), (name, ln)
            n += 1
    assert n == 16


@pytest.mark.parametrize("launch_mode", [1, 2, 3])
@pytest.mark.parametrize("name", _me_ref_cases())
def test_me_ref_tile_motion_and_block_searches(ctx, name, launch_mode):
    """r1_estimate_tile_motion_batch (all references of a case as the jobs of ONE call) and
    r1_estimate_motion_batch on what estimate_tile_motion / estimate_motion of the reference's
    own text produced: every MEStats entry, every (mv, sad, cost)."""
    from rav1e_amd.api import ME_RESULT
    M = np.load(os.path.join(GOLD, "me_ref.npz"))
    w, h, bd, tx, ty, tw, th, hp, full, scale, n_refs, _ = [int(v) for v in M[name + "_meta"]]
    pads = (88, 44, 22)
    lam = [int(v) for v in M[name + "_lambda"]]
    org = [dev_plane(O.plane_from_image(M["%s_org%d" % (name, s)].astype(np.int64), bd, pads[s], pads[s]))
           for s in range(3)]
    jobs, wants = [], []
    for k in range(n_refs):
        ref = [dev_plane(O.plane_from_image(M["%s_ref%d_%d" % (name, k, s)].astype(np.int64), bd, pads[s], pads[s]))
               for s in range(3)]
        want_t, want = _me_stats_tensor(M["%s_stats%d" % (name, k)])
        prev_t, _ = _me_stats_tensor(M["%s_prev%d" % (name, k)])
        st, _ = _me_stats_tensor(np.zeros_like(M["%s_stats%d" % (name, k)]))
        jobs.append(dict(org=org, ref=ref, stats=st, prev=prev_t, tile=(tx, ty, tw, th)))
        wants.append((want_t, want))
    cols, rows = (w + 3) // 4, (h + 3) // 4
    # launch_mode 1: one launch per superblock diagonal; 2 / 3: the persistent row walkers, XCD-pinned / not
    ctx.estimate_tile_motion(jobs, cols, rows, bd, lam, allow_hp=bool(hp), allow_full_search=bool(full),
                             me_range_scale=scale, launch_mode=launch_mode)
    ok, first_failed, calls = ctx.me_status(wait=True)   # no dependency wait ran out of patience
    assert ok and first_failed == 0 and calls >= 1, (ok, first_failed, calls)
    for k, (want_t, want) in enumerate(wants):
        got = jobs[k]["stats"].cpu().numpy().reshape(rows, -1).view(O.ME_STATS).reshape(rows, cols)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (name, k, len(bad), bad[:4], got[tuple(bad[0])], want[tuple(bad[0])])
    if name + "_blk" not in M.files:
        return
    blk = M[name + "_blk"]
    c = np.zeros(len(blk), O.ME_BLOCK_CAND)
    c["bx"], c["by"], c["w"], c["h"], c["corner"] = blk[:, 0], blk[:, 1], blk[:, 2], blk[:, 3], blk[:, 4]
    c["pmv"] = blk[:, 5:9].reshape(-1, 2, 2)
    use_satd, fmode = [int(v) for v in M[name + "_blkcfg"]]
    job = dict(jobs[0], stats=wants[0][0])
    got = ctx.estimate_motion_batch(job, c, cols, rows, bd, lam, use_satd=bool(use_satd), filter_mode=fmode,
                                    allow_hp=bool(hp)).cpu().numpy().view(ME_RESULT)
    want = M[name + "_blkout"]
    for i in range(len(blk)):
        g = (int(got["row"][i]), int(got["col"][i]), int(got["sad"][i]), int(got["cost"][i]))
        assert g == tuple(int(v) for v in want[i]), (name, i, blk[i], g, want[i])


# ---- a14: the CDEF strength search against the executed reference (gen_cdef_search_ref.py) ----
def _cdef_search_file(name):
    # ldc* / ldb*: the CDEF leg as rdo_loop_decision ITSELF, executed whole, ran it (gen_loop_decision_ref.py)
    return os.path.join(GOLD, "loop_decision_ref.npz" if name.startswith("ld") else "cdef_search_ref.npz")


def _cdef_search_cases():
    S, L = np.load(os.path.join(GOLD, "cdef_search_ref.npz")), np.load(os.path.join(GOLD, "loop_decision_ref.npz"))
    return sorted(k[:-5] for k in S.files if k.endswith("_meta")) + \
        sorted(k[:-5] for k in L.files if k.startswith(("ldc", "ldb")) and k.endswith("_meta"))


def run_cdef_search_gpu(ctx, rec, src, skip, scales, prm):
    import torch
    planes = prm.planes
    err, best = ctx.cdef_strength_search(
        [dev_plane(p) for p in rec[:planes]], [dev_plane(p) for p in src[:planes]],
        torch.from_numpy(skip).cuda(), list(prm.y_strengths), list(prm.uv_strengths), prm.damping,
        prm.bit_depth, prm.n_idx, prm.xdec, prm.ydec, prm.crop_w, prm.crop_h,
        area_sb=(prm.area_sb_w, prm.area_sb_h),
        scales=torch.from_numpy(scales.view(np.int32)).cuda() if scales is not None else None,
        dist_scale=list(prm.dist_scale))
    return err.cpu().numpy().view(np.uint64), best.cpu().numpy()


@pytest.mark.parametrize("name", _cdef_search_cases())
def test_cdef_search_ref(ctx, name):
    """r1_cdef_strength_search on what cdef_filter_superblock + rdo_loop_plane_error of the
    reference's own text produced: every (superblock, index) error and every pick."""
    S = np.load(_cdef_search_file(name))
    rec, src, skip, scales, prm, want_err, want_best = O.cdef_search_case(S, name)
    got_err, got_best = run_cdef_search_gpu(ctx, rec, src, skip, scales, prm)
    assert np.array_equal(got_best, want_best), (name, got_best, want_best)
    bad = np.argwhere(got_err != want_err)
    assert len(bad) == 0, (name, bad[:4], got_err[tuple(bad[0])], want_err[tuple(bad[0])])
