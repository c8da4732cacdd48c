"""The N > 1 path on CPU: two processes, gloo backend (127.0.0.1 rendezvous).
Checks the tile sharding (disjoint, complete) and the per-frame exchange step
(all-gather of reconstructed rows rebuilds the whole plane on every rank)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rav1e_amd import tiles, workload as W
    fw, fh, k = 1920, 1080, 2
    mine = tiles.shard_candidates(fw, fh, k, rank, world)
    # every rank can rebuild the whole list: sharding is a partition of it
    whole = W.speed6_ladder(fw, fh, k)
    counts = torch.tensor([len(mine[s]) for s in W.LADDER], dtype=torch.int64)
    tot = counts.clone()
    dist.all_reduce(tot)
    ok_counts = [int(t) == len(whole[s]) for t, s in zip(tot, W.LADDER)]
    # membership: my candidates are exactly the whole-frame ones inside my tile
    rect = W.tile_rects(world, fw, fh)[rank]
    ok_member = True
    for s in W.LADDER:
        w_ = whole[s]
        inside = (w_["ox"] >= rect[0]) & (w_["ox"] < rect[2]) & (w_["oy"] >= rect[1]) & (w_["oy"] < rect[3])
        ok_member &= np.array_equal(w_[inside], mine[s])
    # exchange step: each rank knows only its rows of the "reconstructed" plane
    lay = W.plane_layout(fw, fh, 8)
    full = torch.from_numpy(W.random_plane_array(fw, fh, 8, seed=5))
    local = torch.zeros_like(full)
    rows, lo, hi = tiles.owned_rows(lay["alloc_height"], rank, world)
    local[lo:hi] = full[lo:hi]
    send, gathered = tiles.make_exchange_buffers(local, rank, world)
    tiles.exchange_rows(send, gathered)
    ok_plane = torch.equal(gathered[: lay["alloc_height"]], full)
    q.put((rank, all(ok_counts), bool(ok_member), bool(ok_plane)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_tile_sharding_and_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, okc, okm, okp in res:
        assert okc and okm and okp, (rank, okc, okm, okp)


def test_tile_rects_match_reference_layout():
    """--tiles 8 on 4K -> 4 tile columns x 2 tile rows of 15x17 superblocks
    (src/encoder.rs:248-277, src/tiling/tiler.rs:56-150)."""
    from rav1e_amd import workload as W
    r = W.tile_rects(8, 3840, 2160)
    assert len(r) == 8 and W.tile_split(8, 3840, 2160) == (4, 2)
    assert r[0] == (0, 0, 960, 1088) and r[7] == (2880, 1088, 3840, 2160)
    cover = np.zeros((2160 // 8, 3840 // 8), np.int32)
    for (x0, y0, x1, y1) in r:
        cover[y0 // 8:y1 // 8, x0 // 8:x1 // 8] += 1
    assert (cover == 1).all()
    r1080 = W.tile_rects(8, 1920, 1080)
    assert W.tile_split(8, 1920, 1080) == (4, 2) and r1080[3][2] == 1920


def _worker_widened(rank, world, port, q):
    """N2 / N3 across ranks, CPU oracle as the per-rank engine: tile ME jobs merged by one
    all-reduce; deblocking by row slabs with a 16-row halo equals the whole-frame pass."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import deblock_util as D
    import oracle_lib as O
    from rav1e_amd import tiles, workload as W
    L = O.lib()
    # ---- motion estimation: 2 tiles x 2 references dealt over the ranks ----
    w, h, bd = 256, 128, 8
    rng = np.random.default_rng(1)
    org = O.me_pyramid(rng.integers(0, 256, (h, w)), bd)
    refs = [O.me_pyramid(rng.integers(0, 256, (h, w)), bd) for _ in range(2)]
    rects = W.tile_rects(2, w, h)
    lam = [30, 8, 2]
    whole = np.zeros((2, h // 4, w // 4), O.ME_STATS)
    for r in range(2):
        for (x0, y0, x1, y1) in rects:
            O.me_oracle(L, org, refs[r], w // 4, h // 4, (x0, y0, x1 - x0, y1 - y0), bd, lam, whole[r])
    mine = np.zeros_like(whole)
    for (t, r) in tiles.me_jobs_for_rank(len(rects), 2, rank, world):
        x0, y0, x1, y1 = rects[t]
        O.me_oracle(L, org, refs[r], w // 4, h // 4, (x0, y0, x1 - x0, y1 - y0), bd, lam, mine[r])
    merged = torch.from_numpy(mine.view(np.int32).reshape(2, h // 4, w // 4, 2).copy())
    tiles.merge_me_stats(merged)
    ok_me = np.array_equal(merged.numpy().reshape(2, h // 4, -1).view(O.ME_STATS).reshape(whole.shape), whole)
    # ---- deblocking by row slabs ----
    fw, fh = 192, 256
    blocks = D.random_blocks(np.random.default_rng(2), fw // 4, fh // 4, 1, 1)
    state = D.make_state([30, 26, 0, 0])
    img = np.random.default_rng(3).integers(100, 140, (fh, fw))
    ref_p = O.plane_from_image(img, bd, 16, 16)
    pc = ref_p.cstruct()
    assert L.r1o_deblock_plane(state.ctypes.data, C.byref(pc), 0, 0, 0, blocks.ctypes.data, fw // 4,
                               fw // 4, fh // 4, fw, fh, bd) == 0
    lo, hi, rlo, rhi = tiles.postfilter_slab(fh, rank, world)
    sub = O.plane_from_image(img[rlo:rhi], bd, 16, 16)
    sc = sub.cstruct()
    bsub = np.ascontiguousarray(blocks[rlo // 4:rhi // 4])
    assert L.r1o_deblock_plane(state.ctypes.data, C.byref(sc), 0, 0, 0, bsub.ctypes.data, fw // 4,
                               fw // 4, bsub.shape[0], fw, rhi - rlo, bd) == 0
    out = torch.zeros((fh, fw), dtype=torch.int32)
    out[lo:hi] = torch.from_numpy(sub.view()[lo - rlo:hi - rlo].astype(np.int32))
    dist.all_reduce(out)
    ok_db = np.array_equal(out.numpy(), ref_p.view().astype(np.int32)) and (ref_p.view() != img).any()
    q.put((rank, bool(ok_me), bool(ok_db)))
    dist.barrier()
    dist.destroy_process_group()


def test_me_jobs_and_deblock_slabs_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_widened, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_me, ok_db in res:
        assert ok_me and ok_db, (rank, ok_me, ok_db)


def _worker_halo(rank, world, port, q):
    """N3 with the reconstruction resident per tile: 2-D tiles, point-to-point exchange of the
    tile borders, deblock + CDEF on the expanded tile (CPU oracle as the per-rank engine), then
    one merge of FINAL pixels -- equal to filtering the whole frame."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import deblock_util as D
    import oracle_lib as O
    from rav1e_amd import tiles, workload as W
    L = O.lib()
    fw, fh, bd = 320, 256, 8
    rects = W.tile_rects(world, fw, fh)
    rng = np.random.default_rng(11)                       # same on every rank
    blocks = D.random_blocks(rng, fw // 4, fh // 4, 1, 1)
    state = D.make_state([28, 24, 0, 0])
    yy, xx = np.mgrid[0:fh, 0:fw]
    img = (128 + 40 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 6 * rng.integers(-3, 4, (fh // 8 + 1, fw // 8 + 1))[yy // 8, xx // 8]
           + rng.integers(-2, 3, (fh, fw))).clip(0, 255).astype(np.int64)
    skip = np.ascontiguousarray(blocks["flags"] & 1).astype(np.uint8)
    ystr, uvstr = np.array([2 * 4 + 1] + [0] * 7, np.uint8), np.array([5] + [0] * 7, np.uint8)

    def post_filter(pix, blk, sk):
        """deblock in place, then CDEF, on a (sub-)frame given as an image"""
        h, w = pix.shape
        p = O.plane_from_image(pix, bd, 16, 16)
        pc = p.cstruct()
        assert L.r1o_deblock_plane(state.ctypes.data, C.byref(pc), 0, 0, 0, blk.ctypes.data, blk.shape[1],
                                   blk.shape[1], blk.shape[0], w, h, bd) == 0
        out = O.HostPlane(w, h, bd, 16, 16)
        ci = np.zeros(((h + 63) // 64, (w + 63) // 64), np.uint8)
        oc = out.cstruct()
        L.r1o_cdef_filter_tile_plane(C.byref(pc), C.byref(pc), C.byref(oc), 0, 0, 0, w, h, O.ptr(sk),
                                     sk.shape[1], sk.shape[1], sk.shape[0], O.ptr(ci), ci.shape[1],
                                     O.ptr(ystr), O.ptr(uvstr), 4, bd)
        return out.view().astype(np.int32)

    want = post_filter(img, blocks, skip)
    # this rank knows its own tile only
    x0, y0, x1, y1 = rects[rank]
    local = torch.zeros((fh, fw), dtype=torch.int32)
    local[y0:y1, x0:x1] = torch.from_numpy(img[y0:y1, x0:x1].astype(np.int32))
    tiles.exchange_tile_halos(local, rects, rank)
    ex0, ey0, ex1, ey1 = tiles.expanded_rect(rects[rank], tiles.POSTFILTER_HALO, fw, fh)
    ok_halo = np.array_equal(local.numpy()[ey0:ey1, ex0:ex1], img[ey0:ey1, ex0:ex1])
    outside = local.numpy().copy()
    outside[ey0:ey1, ex0:ex1] = 0
    ok_halo = ok_halo and not outside.any()               # nothing beyond the ring was sent
    sub = post_filter(local.numpy()[ey0:ey1, ex0:ex1].astype(np.int64),
                      np.ascontiguousarray(blocks[ey0 // 4:ey1 // 4, ex0 // 4:ex1 // 4]),
                      np.ascontiguousarray(skip[ey0 // 4:ey1 // 4, ex0 // 4:ex1 // 4]))
    final = torch.zeros((fh, fw), dtype=torch.int32)
    final[y0:y1, x0:x1] = torch.from_numpy(sub[y0 - ey0:y1 - ey0, x0 - ex0:x1 - ex0].copy())
    dist.all_reduce(final)                                # disjoint tiles: the sum is the merge
    ok_final = np.array_equal(final.numpy(), want) and (want != img).mean() > 0.05
    sends, recvs = tiles.tile_halo_plan(rects, rank, tiles.POSTFILTER_HALO, fw, fh)
    sent = sum((r[2] - r[0]) * (r[3] - r[1]) for _, r in sends)
    q.put((rank, bool(ok_halo), bool(ok_final), sent, len(sends), len(recvs)))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_halo_exchange_and_postfilter_gloo():
    """four ranks, 2 x 2 tiles (every rank has an edge neighbour in both directions and a corner
    neighbour)"""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_halo, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_halo, ok_final, sent, ns, nr in res:
        assert ok_halo and ok_final, (rank, ok_halo, ok_final)
        assert ns == 3 and nr == 3 and sent < 320 * 256 // 2, (rank, ns, nr, sent)


def _verify_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from rav1e_amd import tiles, workload as W
    fw, fh = 1920, 1080
    rects = W.tile_rects(world, fw, fh)
    data = torch.from_numpy(W.random_plane_array(fw, fh, 8, seed=3)).clone()
    before = data.clone()
    plane = SimpleNamespace(data=data, width=fw, height=fh, xorigin=88, yorigin=88)
    vis = data[88:88 + fh, 88:88 + fw]

    def gather(swap=False):
        for r in range(world):
            x0, y0, x1, y1 = rects[r]
            t = vis[y0:y1, x0:x1].contiguous()
            dist.broadcast(t, src=r)
            if swap and r != rank and world > 1:
                t = t + 1            # a tile that arrives with the wrong owner's bytes
            vis[y0:y1, x0:x1] = t
    good = tiles.verify_exchange(plane, rects, rank, world,
                                 lambda: tiles.exchange_tile_halos(vis, rects, rank), gather)
    restored = torch.equal(data, before)
    # a halo leg that does nothing and a gather that delivers the wrong bytes are both caught
    bad = tiles.verify_exchange(plane, rects, rank, world, lambda: None, lambda: gather(swap=True))
    q.put((rank, good, bad, restored and torch.equal(data, before)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_verify_exchange_self_check_gloo(world):
    """bench.py's pre-run self-check (tiles.verify_exchange): tagged tiles through the halo exchange
    and the tile gather; passes on the real exchange, fails on a no-op halo leg and on a gather that
    lands foreign bytes; the plane is restored either way."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, good, bad, restored in res:
        assert good == {"halo": True, "gather": True}, (rank, good)
        assert bad == {"halo": False, "gather": False}, (rank, bad)
        assert restored, rank


def test_tile_halo_plan_is_symmetric():
    """what a rank sends to a peer is exactly what that peer expects to receive from it (4K, 8 tiles)"""
    from rav1e_amd import tiles, workload as W
    rects = W.tile_rects(8, 3840, 2160)
    plans = [tiles.tile_halo_plan(rects, r, 64, 3840, 2160) for r in range(8)]
    for r in range(8):
        for peer, rect in plans[r][0]:
            assert (r, rect) in plans[peer][1]
        for peer, rect in plans[r][1]:
            assert (r, rect) in plans[peer][0]
        assert sum((a[2] - a[0]) * (a[3] - a[1]) for _, a in plans[r][0]) < 400_000   # vs 8.3 M pixels


# ---- the RCCL path itself, two ranks on two GPUs (skipped on 1-GPU boxes) ------------------
def _worker_rccl(rank, world, port, q):
    try:
        import numpy as np
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)   # carries the unique id only
        from rav1e_amd import tiles
        from rav1e_amd import workload as W
        from rav1e_amd.api import Context, Plane
        fw, fh = 512, 256
        rects = [(0, 0, 256, 256), (256, 0, 512, 256)]
        truth = W.random_plane_array(fw, fh, 8, 77)               # what every rank must end up with
        mine = np.zeros_like(truth)
        x0, y0, x1, y1 = rects[rank]
        geo = Plane(fw, fh, 8, 88, 88, device="cpu")
        xo, yo = geo.xorigin, geo.yorigin          # the visible area's origin inside the allocation
        mine[yo + y0:yo + y1, xo + x0:xo + x1] = truth[yo + y0:yo + y1, xo + x0:xo + x1]
        ctx = Context(rank)
        dp = Plane.from_numpy(mine, fw, fh, 8, 88, 88)
        comm = tiles.Comm(ctx, rank, world)
        lib = comm.lib.r1_comm_library().decode()
        n = comm.exchange_tile_halos(dp, rects)
        torch.cuda.synchronize()
        got = dp.data.cpu().numpy()
        ex = tiles.expanded_rect(rects[rank], tiles.POSTFILTER_HALO, fw, fh)
        ok_halo = np.array_equal(got[yo + ex[1]:yo + ex[3], xo + ex[0]:xo + ex[2]],
                                 truth[yo + ex[1]:yo + ex[3], xo + ex[0]:xo + ex[2]])
        comm.allgather_tiles(dp, rects)
        torch.cuda.synchronize()
        got = dp.data.cpu().numpy()
        ok_all = np.array_equal(got[yo:yo + fh, xo:xo + fw], truth[yo:yo + fh, xo:xo + fw])
        # bench.py's pre-run self-check on the same communicator, and the plane back as it was
        v = tiles.verify_exchange(dp, rects, rank, world, lambda: comm.exchange_tile_halos(dp, rects),
                                  lambda: comm.allgather_tiles(dp, rects))
        ok_all = ok_all and v == {"halo": True, "gather": True} and np.array_equal(dp.data.cpu().numpy(), got)
        # the same exchange as direct peer stores through the communicator (planes mapped over IPC,
        # r1_comm_barrier as the hand-shake)
        pp = tiles.PeerPlanes(ctx, dp, rank, world, comm=comm)
        v = tiles.verify_exchange(dp, rects, rank, world, lambda: pp.push_halos(rects),
                                  lambda: pp.push_tile(rects), pre=dist.barrier)
        ok_all = ok_all and v == {"halo": True, "gather": True} and np.array_equal(dp.data.cpu().numpy(), got)
        dist.barrier()
        pp.close()
        comm.close()
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, n, ok_halo, ok_all, lib, None))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, 0, False, False, "", traceback.format_exc()[-1500:]))


@pytest.mark.gpu
def test_rccl_halo_exchange_and_tile_allgather_two_gpus():
    """r1_comm_exchange_halos + r1_comm_allgather_tiles with world = 2 over RCCL: every rank ends
    with its tile plus the 64-px ring after the p2p exchange and with the whole frame after the
    all-gather.  Needs two GPUs; the 1-GPU boxes of the round run skip it."""
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_worker_rccl, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, n, ok_halo, ok_all, lib, err in res:
        assert err is None, "rank %d: %s" % (rank, err)
        assert n == 2 and ok_halo and ok_all, (rank, n, ok_halo, ok_all)
        assert "librccl" in lib


# ---- direct peer stores: two processes sharing ONE GPU map each other's plane over IPC ----------
def _worker_peer_stores(rank, world, port, q, bd):
    try:
        import numpy as np
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from rav1e_amd import tiles
        from rav1e_amd import workload as W
        from rav1e_amd.api import Context, Plane
        fw, fh = 650, 333                        # odd sizes: rows with unaligned heads and tails
        rects = W.tile_rects(world, fw, fh) if world > 2 else [(0, 0, 323, 333), (323, 0, 650, 333)]
        truth = W.random_plane_array(fw, fh, bd, 78)
        mine = np.zeros_like(truth)
        x0, y0, x1, y1 = rects[rank]
        geo = Plane(fw, fh, bd, 88, 88, device="cpu")
        xo, yo = geo.xorigin, geo.yorigin          # the visible area's origin inside the allocation
        mine[yo + y0:yo + y1, xo + x0:xo + x1] = truth[yo + y0:yo + y1, xo + x0:xo + x1]
        ctx = Context(0)
        # a filler of a rank-dependent size first: the planes sit at different offsets of their
        # allocator blocks in the two processes
        filler = torch.empty(4096 * (1 + 3 * rank), dtype=torch.uint8, device="cuda")
        dp = Plane.from_numpy(mine, fw, fh, bd, 88, 88)
        pp = tiles.PeerPlanes(ctx, dp, rank, world, comm=None)
        dist.barrier()                            # both planes filled before anybody stores
        n = pp.push_halos(rects, halo=37)
        got = dp.data.cpu().numpy()
        ex = tiles.expanded_rect(rects[rank], 37, fw, fh)
        ok_halo = np.array_equal(got[yo + ex[1]:yo + ex[3], xo + ex[0]:xo + ex[2]],
                                 truth[yo + ex[1]:yo + ex[3], xo + ex[0]:xo + ex[2]])
        # and nothing outside the ring arrived
        outside = got[yo:yo + fh, xo:xo + fw].copy()
        outside[ex[1]:ex[3], ex[0]:ex[2]] = 0
        ok_halo = ok_halo and not outside.any()
        dist.barrier()
        pp.push_tile(rects)
        got = dp.data.cpu().numpy()
        ok_all = np.array_equal(got[yo:yo + fh, xo:xo + fw], truth[yo:yo + fh, xo:xo + fw])
        # the padding around the visible area is nobody's tile: untouched
        pad = got.copy()
        pad[yo:yo + fh, xo:xo + fw] = 0
        ok_all = ok_all and not pad.any()
        v = tiles.verify_exchange(dp, rects, rank, world, lambda: pp.push_halos(rects),
                                  lambda: pp.push_tile(rects), pre=dist.barrier)
        ok_all = ok_all and v == {"halo": True, "gather": True} and np.array_equal(dp.data.cpu().numpy(), got)
        # a second plane while the first is still mapped: small planes share an allocator block, so
        # the peer's allocation is already mapped here (the library maps an allocation once)
        truth2 = W.random_plane_array(fw, fh, bd, 79)
        mine2 = np.zeros_like(truth2)
        mine2[yo + y0:yo + y1, xo + x0:xo + x1] = truth2[yo + y0:yo + y1, xo + x0:xo + x1]
        dq = Plane.from_numpy(mine2, fw, fh, bd, 88, 88)
        pq = tiles.PeerPlanes(ctx, dq, rank, world, comm=None)
        dist.barrier()
        pq.push_tile(rects)
        got2 = dq.data.cpu().numpy()
        ok_all = ok_all and np.array_equal(got2[yo:yo + fh, xo:xo + fw], truth2[yo:yo + fh, xo:xo + fw])
        ok_all = ok_all and np.array_equal(dp.data.cpu().numpy(), got)       # the first plane: untouched
        dist.barrier()
        pp.close()
        # the shared mapping outlives the first close
        pq.push_tile(rects)
        ok_all = ok_all and np.array_equal(dq.data.cpu().numpy(), got2)
        dist.barrier()
        pq.close()
        del filler
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, n, ok_halo, ok_all, None))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, 0, False, False, traceback.format_exc()[-1500:]))


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10])
def test_peer_stores_two_processes_one_gpu(bd):
    """r1_ipc_export / r1_ipc_open / r1_push_rects: two processes on the SAME GPU (RCCL would refuse
    the pair; the mapping and the store kernel do not care which GPU the peer's memory is on) map
    each other's plane and store their halo rectangles, then their tiles, into it.  Odd sizes and
    different allocator offsets exercise the unaligned row edges."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_worker_peer_stores, args=(r, 2, port, q, bd)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, n, ok_halo, ok_all, err in res:
        assert err is None, "rank %d: %s" % (rank, err)
        assert n == 1 and ok_halo and ok_all, (rank, n, ok_halo, ok_all)


def _worker_ring(rank, world, port, q, bd, steps, devices, overlap=False):
    """tiles.TileRing -- the code bench.py --gpus N steps -- for `steps` ring steps with the check after
    EVERY step; devices: one GPU index per rank (the same one twice: two processes on one GPU)."""
    try:
        import numpy as np
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(devices[rank])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from rav1e_amd import tiles
        from rav1e_amd import workload as W
        from rav1e_amd.api import Context, Plane
        fw, fh = 1290, 714
        rects = W.tile_rects(world, fw, fh)
        host = W.random_plane_array(fw, fh, bd, 91)
        ctx = Context(devices[rank])
        ring = [Plane.from_numpy(host, fw, fh, bd, 88, 88) for _ in range(2)]
        peers = [tiles.PeerPlanes(ctx, pl, rank, world, comm=None) for pl in ring]
        dist.barrier()
        tr = tiles.TileRing(ring, peers, rects, rank, tiles.visible(ring[0]).clone())
        ok0 = tr.check()                                   # t = 0: both planes are the original
        bad_at = -1
        side = torch.cuda.Stream() if overlap else None
        for i in range(steps):
            if overlap:       # halo stores on a side stream beside the interior's work (TileRing.advance_overlapped)
                tr.advance_overlapped(side)
            else:
                tr.advance()
            if not tr.check() and bad_at < 0:
                bad_at = i
        # the check is not vacuous: a step whose tile store is skipped on ONE rank leaves a stale tag in
        # the OTHER rank's copy (and a late store -- the previous ring cycle's bytes -- does too)
        nxt = tr.cur ^ 1
        torch.bitwise_xor(tr.tiles[tr.cur], tiles.ring_delta(tr.t), out=tr.tiles[nxt])
        if rank != 0:
            tr.peers[nxt].push_halos(rects)
            tr.peers[nxt].push_tile(rects)                 # rank 0 "forgets" its stores
        else:
            torch.cuda.synchronize()
            dist.barrier()
            dist.barrier()                                 # the two hand-shakes the peers run
        tr.cur, tr.t = nxt, tr.t + 1
        stale_seen = not tr.check()
        flags = torch.tensor([1.0 if stale_seen else 0.0])
        dist.all_reduce(flags, op=dist.ReduceOp.SUM)       # every rank but the forgetful one must see it
        dist.barrier()
        for pp in peers:
            pp.close()
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok0, bad_at, int(flags.item()), None))
    except Exception:   # noqa: BLE001
        import traceback
        q.put((rank, False, 0, 0, traceback.format_exc()[-1500:]))


def _run_ring(devices, bd, steps, overlap=False):
    import torch.multiprocessing as mp
    world = len(devices)
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_worker_ring, args=(r, world, port, q, bd, steps, devices, overlap)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, ok0, bad_at, stale, err in res:
        assert err is None, "rank %d: %s" % (rank, err)
        assert ok0 and bad_at == -1, (rank, ok0, "first bad step", bad_at)
        assert stale == world - 1, (rank, stale)


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10])
def test_tile_ring_240_steps_two_processes_one_gpu(bd):
    """The timed loop's exchange as a stress test, not one exchange: tiles.TileRing (stand-in
    reconstruction into the other plane of the ring, halo stores, tile stores, hand-shake) for 240 steps
    between two processes sharing one GPU, every rank's copy of EVERY tile checked against
    original ^ ring_tag(t) after every step; then a step in which one rank skips its stores must be seen
    by the other."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _run_ring([0, 0], bd, 240)


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10])
def test_tile_ring_overlapped_exchange_two_processes_one_gpu(bd):
    """the ring with the exchange's halo leg on a side stream, gated by an event behind the tile's border and running
    beside its interior (tiles.TileRing.advance_overlapped): 240 steps, every rank's copy of every tile after every
    step, and the dropped-store detection, as for the serial schedule"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _run_ring([0, 0], bd, 240, overlap=True)


@pytest.mark.gpu
def test_tile_ring_overlapped_exchange_world_1_with_pool(ctx):
    """world 1 through the C-ABI communicator: the plane POOL opened in one call (r1_comm_plane_pool_open), the
    overlapped step's control flow (events, side stream, empty store lists), the ring check"""
    import torch
    from rav1e_amd import tiles, workload as W
    from rav1e_amd.api import Plane
    fw, fh = 640, 360
    host = W.random_plane_array(fw, fh, 8, 17)
    ring = [Plane.from_numpy(host, fw, fh, 8, 88, 88) for _ in range(2)]
    comm = tiles.Comm(ctx, 0, 1)
    peers = tiles.PeerPlanes.open_pool(ctx, ring, 0, 1, comm=comm)
    assert len(peers) == 2 and peers[0].ptrs[0] == ring[0].data.data_ptr() and peers[1].ptrs[0] == ring[1].data.data_ptr()
    tr = tiles.TileRing(ring, peers, W.tile_rects(1, fw, fh), 0, tiles.visible(ring[0]).clone())
    side = torch.cuda.Stream()
    for _ in range(12):
        tr.advance_overlapped(side)
        assert tr.check()
    for pp in peers:
        pp.close()
    comm.close()


@pytest.mark.gpu
def test_tile_ring_240_steps_two_gpus():
    """the same loop across two physical GPUs (separate L2s / Infinity Caches: the cross-device visibility
    of peer stores after the hand-shake).  Needs two GPUs; the 1-GPU boxes of the round skip it."""
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_ring([0, 1], 8, 240)


def test_ring_tags_have_a_period_longer_than_the_ring():
    """ring_tag / ring_delta: composing the deltas reproduces the tags, four distinct values, so neither a
    missing store (tag t - 1 ... wait: the other plane holds tag t - 1) nor one that is a whole two-plane
    cycle late (tag t - 2) can pass for tag t"""
    from rav1e_amd import tiles
    v = 0
    for t in range(64):
        assert v == tiles.ring_tag(t)
        assert tiles.ring_tag(t) != tiles.ring_tag(t + 1) and tiles.ring_tag(t) != tiles.ring_tag(t + 2)
        v ^= tiles.ring_delta(t)


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10])
def test_push_rects_into_a_second_plane(bd):
    """r1_push_rects inside one process: 40 random rectangles (three launches of <= 16) of one plane
    stored into two other planes of the same geometry; everything outside the rectangles stays."""
    import ctypes as C
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from rav1e_amd import tiles, workload as W
    from rav1e_amd.api import Context, Plane
    fw, fh = 777, 211
    rng = np.random.default_rng(5)
    src = W.random_plane_array(fw, fh, bd, 3)
    ctx = Context(0)
    sp = Plane.from_numpy(src, fw, fh, bd, 88, 88)
    dsts = [Plane.from_numpy(np.zeros_like(src), fw, fh, bd, 88, 88) for _ in range(2)]
    xo, yo = sp.xorigin, sp.yorigin
    ptrs = (C.c_void_p * 2)(*[d.data.data_ptr() for d in dsts])
    x = np.zeros(40, tiles.PUSH_RECT)
    want = [np.zeros_like(src), np.zeros_like(src)]
    for i in range(40):
        x0, y0 = int(rng.integers(0, fw - 1)), int(rng.integers(0, fh - 1))
        x1, y1 = int(rng.integers(x0 + 1, min(fw, x0 + 300) + 1)), int(rng.integers(y0 + 1, min(fh, y0 + 40) + 1))
        peer = int(rng.integers(0, 2))
        x[i] = (peer, x0, y0, x1, y1)
        want[peer][yo + y0:yo + y1, xo + x0:xo + x1] = src[yo + y0:yo + y1, xo + x0:xo + x1]
    p = sp.cstruct()
    lib = ctx.lib
    rc = lib.r1_push_rects(ctx.h, C.byref(p), ptrs, 2, x.ctypes.data, 40, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.r1_last_error()
    torch.cuda.synchronize()
    for d, w in zip(dsts, want):
        assert np.array_equal(d.data.cpu().numpy(), w)
    # a rectangle outside the visible area, an unknown peer, the plane itself as its own peer: refused
    bad = x[:1].copy()
    bad["x1"] = fw + 1
    assert lib.r1_push_rects(ctx.h, C.byref(p), ptrs, 2, bad.ctypes.data, 1, None) == -1
    bad = x[:1].copy()
    bad["peer"] = 2
    assert lib.r1_push_rects(ctx.h, C.byref(p), ptrs, 2, bad.ctypes.data, 1, None) == -1
    own = (C.c_void_p * 2)(sp.data.data_ptr(), sp.data.data_ptr())
    assert lib.r1_push_rects(ctx.h, C.byref(p), own, 2, x.ctypes.data, 1, None) == -1
    ctx.close()
