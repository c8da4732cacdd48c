"""Tile sharding across GPUs and the one real exchange step of the path.

Tiles are encoded independently in the reference (separate TileStateMut per
tile, src/encoder.rs:3245-3257), so rank r owns tile r and evaluates that
tile's candidates with no data-path collective.  What every rank needs after a
frame is the whole reconstructed frame as the next reference (motion vectors
are clamped to the frame, not the tile: src/me.rs:339-362), i.e. one
all-gather of reconstructed rows per coded frame.  For the post
filters the tile borders travel point to point first (exchange_tile_halos), so
that the all-gather carries final pixels.  Everything here works on any
torch.distributed backend: RCCL ("nccl") on the GPUs, gloo on CPU for the tests.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import workload as W

HALO_XFER = np.dtype([("peer", "<i4"), ("dir", "<i4"), ("x0", "<i4"), ("y0", "<i4"), ("x1", "<i4"), ("y1", "<i4")])


class Comm:
    """The exchange behind the C ABI (csrc/comm.hip: RCCL linked into librav1e_hip.so): what a
    Rust host would bind.  torch.distributed is used ONLY to carry the 128-byte unique id from
    rank 0 to the others (any channel would do)."""

    def __init__(self, ctx, rank, world, group=None):
        from . import _lib
        self.lib = _lib.load()
        self.rank, self.world = rank, world
        # every rank goes through the same collectives whatever fails locally: a rank that
        # raised before the broadcast (or created alone) would leave the others waiting
        idb, err = bytearray(128), None
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            if self.lib.r1_comm_unique_id(buf) != 0:
                err = "r1_comm_unique_id: " + self.lib.r1_last_error().decode()
            idb = bytearray(buf)
        if world > 1:
            box = [None if err else bytes(idb)]
            dist.broadcast_object_list(box, src=0, group=group)
            if box[0] is None:
                raise RuntimeError(err or "rank 0 could not make an RCCL unique id")
            idb = bytearray(box[0])
        elif err:
            raise RuntimeError(err)
        h = C.c_void_p()
        arr = (C.c_uint8 * 128).from_buffer(idb)
        rc = self.lib.r1_comm_create(ctx.h, rank, world, arr, C.byref(h))
        if rc != 0:
            err = "r1_comm_create: " + self.lib.r1_last_error().decode()
        if world > 1:
            oks = [None] * world
            dist.all_gather_object(oks, err is None, group=group)
            if not all(oks):
                if rc == 0:
                    self.lib.r1_comm_destroy(h)
                raise RuntimeError(err or "r1_comm_create failed on rank(s) %s"
                                   % [i for i, o in enumerate(oks) if not o])
        elif err:
            raise RuntimeError(err)
        self.h = h

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.r1_last_error().decode()))

    def close(self):
        if self.h:
            self.lib.r1_comm_destroy(self.h)
            self.h = None

    def allgather(self, send, recv):
        """tensors: every rank's `send` (same byte size) lands in recv in rank order"""
        nbytes = send.numel() * send.element_size()
        assert recv.numel() * recv.element_size() == nbytes * self.world
        self._check(self.lib.r1_comm_allgather(self.h, send.data_ptr(), recv.data_ptr(), nbytes,
                                               torch.cuda.current_stream().cuda_stream), "r1_comm_allgather")
        return recv

    def allgather_tiles(self, plane, rects):
        """plane: rav1e_amd.api.Plane; rects[r] = (x0, y0, x1, y1) owned by rank r"""
        assert len(rects) == self.world
        r4 = np.ascontiguousarray(np.array(rects, np.int32).reshape(-1))
        p = plane.cstruct()
        self._check(self.lib.r1_comm_allgather_tiles(self.h, C.byref(p), r4.ctypes.data,
                                                     torch.cuda.current_stream().cuda_stream),
                    "r1_comm_allgather_tiles")

    def exchange_tile_halos(self, plane, rects, halo=None):
        """the tile-boundary rectangles of tile_halo_plan, in one grouped send / receive"""
        halo = POSTFILTER_HALO if halo is None else halo
        sends, recvs = tile_halo_plan(rects, self.rank, halo, plane.width, plane.height)
        x = np.zeros(len(sends) + len(recvs), HALO_XFER)
        for i, (peer, r) in enumerate(sends + recvs):
            x[i] = (peer, int(i >= len(sends)), r[0], r[1], r[2], r[3])
        if not len(x):
            return 0
        p = plane.cstruct()
        self._check(self.lib.r1_comm_exchange_halos(self.h, C.byref(p), x.ctypes.data, len(x),
                                                    torch.cuda.current_stream().cuda_stream),
                    "r1_comm_exchange_halos")
        return len(x)


IPC_MEM = np.dtype([("handle", "u1", (64,)), ("offset", "<u8"), ("bytes", "<u8")])
PUSH_RECT = np.dtype([("peer", "<i4"), ("x0", "<i4"), ("y0", "<i4"), ("x1", "<i4"), ("y1", "<i4")])
assert IPC_MEM.itemsize == 80 and PUSH_RECT.itemsize == 20


class PeerPlanes:
    """Every rank's copy of one plane (same geometry everywhere) mapped into this process, for the
    exchange as direct peer stores (csrc/comm.hip: r1_push_rects and friends): push_tile stores
    this rank's tile into every peer's plane, push_halos the border rectangles of tile_halo_plan,
    each followed by the hand-shake.

    comm given: the exports travel through the C ABI's communicator and the hand-shake is
    r1_comm_barrier (stream-ordered, RCCL).  comm None: the exports travel through
    torch.distributed (`group`; any backend) and the hand-shake is a stream synchronize + a
    torch.distributed barrier -- the form the tests use to drive the stores between two processes
    that share ONE GPU, where RCCL refuses to form a communicator."""

    @classmethod
    def open_pool(cls, ctx, planes, rank, world, comm=None, group=None):
        """The plane POOL of a host (live reference slots + 1 planes of one geometry) -> [PeerPlanes], one per plane.
        With a communicator the whole pool is mapped in ONE blocking collective (r1_comm_plane_pool_open: n_planes x 80
        bytes per rank in one all-gather) instead of one per plane; close() of the FIRST entry unmaps the pool.  Without
        one (the torch.distributed form of the tests) the planes are opened one by one."""
        if comm is None:
            return [cls(ctx, p, rank, world, None, group) for p in planes]
        from . import _lib
        lib = _lib.load()
        n = len(planes)
        arr = (_lib.R1Plane * n)(*[p.cstruct() for p in planes])
        ptrs = (C.c_void_p * (n * world))()
        rc = lib.r1_comm_plane_pool_open(comm.h, ctx.h, arr, n, ptrs)
        if rc != 0:
            raise RuntimeError("r1_comm_plane_pool_open failed (%d): %s" % (rc, lib.r1_last_error().decode()))
        out = []
        for i, p in enumerate(planes):
            pp = cls.__new__(cls)
            pp.lib, pp.ctx, pp.plane, pp.rank, pp.world, pp.comm, pp.group = lib, ctx, p, rank, world, comm, group
            pp.ptrs = (C.c_void_p * world).from_buffer(ptrs, i * world * C.sizeof(C.c_void_p))   # a view, not a copy
            pp._pool = (ptrs, n) if i == 0 else None
            pp._pooled = True
            out.append(pp)
        return out

    def __init__(self, ctx, plane, rank, world, comm=None, group=None):
        from . import _lib
        self.lib = _lib.load()
        self.ctx, self.plane, self.rank, self.world, self.comm, self.group = ctx, plane, rank, world, comm, group
        self.ptrs = (C.c_void_p * world)()
        self._pool, self._pooled = None, False
        p = plane.cstruct()
        if comm is not None:
            self._check(self.lib.r1_comm_open_peer_planes(comm.h, ctx.h, C.byref(p), self.ptrs),
                        "r1_comm_open_peer_planes")
        else:
            mine = np.zeros(1, IPC_MEM)
            nbytes = plane.data.numel() * plane.data.element_size()
            rc = self.lib.r1_ipc_export(ctx.h, plane.data.data_ptr(), nbytes, mine.ctypes.data)
            err = None if rc == 0 else "r1_ipc_export: " + self.lib.r1_last_error().decode()
            box = [None] * world
            if world > 1:
                dist.all_gather_object(box, None if err else mine.tobytes(), group=group)
            else:
                box[0] = None if err else mine.tobytes()
            if any(b is None for b in box):
                raise RuntimeError(err or "r1_ipc_export failed on rank(s) %s"
                                   % [i for i, b in enumerate(box) if b is None])
            self.ptrs[rank] = plane.data.data_ptr()
            for r in range(world):
                if r == rank:
                    continue
                m = np.frombuffer(box[r], IPC_MEM).copy()
                if int(m["bytes"][0]) != nbytes:
                    err = err or "rank %d exported %d bytes, this rank's plane has %d" % (r, int(m["bytes"][0]), nbytes)
                    continue
                out = C.c_void_p()
                if self.lib.r1_ipc_open(ctx.h, m.ctypes.data, C.byref(out)) != 0:
                    err = err or "r1_ipc_open(rank %d): %s" % (r, self.lib.r1_last_error().decode())
                    continue
                self.ptrs[r] = out.value
            if err:
                self.close()
                raise RuntimeError(err)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.r1_last_error().decode()))

    def close(self):
        if self.ptrs is None:
            return
        if self._pooled:
            if self._pool is not None:      # the first entry of an open_pool list owns the mapping
                self.lib.r1_comm_plane_pool_close(self.comm.h, self.ctx.h, self._pool[1], self._pool[0])
                self._pool = None
        elif self.comm is not None:
            self.lib.r1_comm_close_peer_planes(self.comm.h, self.ctx.h, self.ptrs)
        else:
            for r in range(self.world):
                if r != self.rank and self.ptrs[r]:
                    self.lib.r1_ipc_close(self.ctx.h, self.ptrs[r])
                self.ptrs[r] = None
        self.ptrs = None

    def _handshake(self):
        torch.cuda.current_stream().synchronize()
        if self.world > 1:
            dist.barrier(group=self.group)

    def push_tile(self, rects):
        """this rank's tile into every peer's plane; afterwards (in stream order with comm, on
        return without) every rank's plane holds every tile"""
        st = torch.cuda.current_stream().cuda_stream
        p = self.plane.cstruct()
        if self.comm is not None:
            r4 = np.ascontiguousarray(np.array(rects, np.int32).reshape(-1))
            self._check(self.lib.r1_comm_push_tile(self.comm.h, self.ctx.h, C.byref(p), self.ptrs,
                                                   r4.ctypes.data, st), "r1_comm_push_tile")
            return
        x = np.zeros(self.world - 1, PUSH_RECT)
        for i, r in enumerate(q for q in range(self.world) if q != self.rank):
            x[i] = (r,) + tuple(int(v) for v in rects[self.rank])
        self._check(self.lib.r1_push_rects(self.ctx.h, C.byref(p), self.ptrs, self.world,
                                           x.ctypes.data, len(x), st), "r1_push_rects")
        self._handshake()

    def push_halos(self, rects, halo=None):
        """the border rectangles of tile_halo_plan into the neighbours' planes"""
        halo = POSTFILTER_HALO if halo is None else halo
        st = torch.cuda.current_stream().cuda_stream
        p = self.plane.cstruct()
        sends, _ = tile_halo_plan(rects, self.rank, halo, self.plane.width, self.plane.height)
        if self.comm is not None:
            x = np.zeros(len(sends), HALO_XFER)
            for i, (peer, r) in enumerate(sends):
                x[i] = (peer, 0, r[0], r[1], r[2], r[3])
            self._check(self.lib.r1_comm_push_halos(self.comm.h, self.ctx.h, C.byref(p), self.ptrs,
                                                    x.ctypes.data, len(x), st), "r1_comm_push_halos")
            return len(x)
        x = np.zeros(len(sends), PUSH_RECT)
        for i, (peer, r) in enumerate(sends):
            x[i] = (peer, r[0], r[1], r[2], r[3])
        self._check(self.lib.r1_push_rects(self.ctx.h, C.byref(p), self.ptrs, self.world,
                                           x.ctypes.data, len(x), st), "r1_push_rects")
        self._handshake()
        return len(x)


def _push_frame(self, rects, halo=None):
    """both legs of a frame's exchange -- the border rectangles, then the tile into every peer -- behind ONE hand-shake
    (r1_comm_push_frame; without a communicator: two store launches, one stream synchronize + barrier)"""
    halo = POSTFILTER_HALO if halo is None else halo
    st = torch.cuda.current_stream().cuda_stream
    p = self.plane.cstruct()
    sends, _ = tile_halo_plan(rects, self.rank, halo, self.plane.width, self.plane.height)
    if self.comm is not None:
        x = np.zeros(len(sends), HALO_XFER)
        for i, (peer, r) in enumerate(sends):
            x[i] = (peer, 0, r[0], r[1], r[2], r[3])
        r4 = np.ascontiguousarray(np.array(rects, np.int32).reshape(-1))
        self._check(self.lib.r1_comm_push_frame(self.comm.h, self.ctx.h, C.byref(p), self.ptrs, x.ctypes.data, len(x),
                                                r4.ctypes.data, st), "r1_comm_push_frame")
        return
    x = np.zeros(len(sends) + self.world - 1, PUSH_RECT)
    for i, (peer, r) in enumerate(sends):
        x[i] = (peer, r[0], r[1], r[2], r[3])
    for i, r in enumerate(q for q in range(self.world) if q != self.rank):
        x[len(sends) + i] = (r,) + tuple(int(v) for v in rects[self.rank])
    self._check(self.lib.r1_push_rects(self.ctx.h, C.byref(p), self.ptrs, self.world, x.ctypes.data, len(x), st),
                "r1_push_rects")
    self._handshake()


PeerPlanes.push_frame = _push_frame


def _push_halos_async(self, rects, halo=None, stream=None):
    """the border rectangles of tile_halo_plan into the neighbours' planes on `stream` (a torch stream; None = the
    current one), NO hand-shake: r1_push_rects only.  The caller orders these stores before the step's hand-shake --
    TileRing.advance_overlapped makes the main stream wait for `stream` before it stores the tile and hand-shakes.
    This is the leg that can leave early: the neighbours' post filters (deblock up to 7 px, CDEF 2 px, restoration 4 px
    across a tile edge, src/encoder.rs:3263-3322) need only the blocks along the tile's edges, which a host codes
    first or knows to be final long before the tile's last superblock."""
    halo = POSTFILTER_HALO if halo is None else halo
    st = (stream or torch.cuda.current_stream()).cuda_stream
    p = self.plane.cstruct()
    sends, _ = tile_halo_plan(rects, self.rank, halo, self.plane.width, self.plane.height)
    x = np.zeros(len(sends), PUSH_RECT)
    for i, (peer, r) in enumerate(sends):
        x[i] = (peer, r[0], r[1], r[2], r[3])
    if len(x):
        self._check(self.lib.r1_push_rects(self.ctx.h, C.byref(p), self.ptrs, self.world, x.ctypes.data, len(x), st),
                    "r1_push_rects")
    return len(x)


PeerPlanes.push_halos_async = _push_halos_async


def verify_exchange(plane, rects, rank, world, do_halos, do_gather, halo=None, pre=None):
    """Self-check of the exchange before a multi-GPU run is timed: did the bytes land where the
    tile grid says?  Every rank paints its own tile of `plane` with its tag (rank + 1; the rest of
    the visible area with 0 = nobody), runs the halo exchange and checks that the `halo`-pixel
    ring of its tile holds the OWNER's tag wherever a neighbour's tile covers it; then runs the
    tile gather and checks every tile of the frame.  The plane's contents are restored.
    do_halos / do_gather: callables that run the exchange on the current stream (None: that
    leg is not part of the path in use, e.g. the torch.distributed fallback has no halo leg).
    pre: called after painting, before the exchange (a host barrier when the exchange is peer
    stores: a rank's paint must not run over what a faster peer already stored into its plane).
    -> dict(halo=bool | None, gather=bool | None) for THIS rank; the caller reduces over ranks."""
    halo = POSTFILTER_HALO if halo is None else halo
    saved = plane.data.clone()
    fw, fh = plane.width, plane.height

    def sync():
        if plane.data.is_cuda:
            torch.cuda.synchronize()
    vis = plane.data[plane.yorigin:plane.yorigin + fh, plane.xorigin:plane.xorigin + fw]

    def paint():
        vis.zero_()
        x0, y0, x1, y1 = rects[rank]
        vis[y0:y1, x0:x1] = rank + 1
        if pre is not None:
            # peer stores write into OTHER ranks' planes: nobody stores before everybody painted
            sync()
            pre()
    res = {"halo": None, "gather": None}
    try:
        if do_halos is not None:
            paint()
            do_halos()
            sync()
            _, recvs = tile_halo_plan(rects, rank, halo, fw, fh)
            ok = True
            for peer, (x0, y0, x1, y1) in recvs:
                ok = ok and bool((vis[y0:y1, x0:x1] == peer + 1).all().item())
            # and nothing beyond the ring was touched: my tile keeps my tag, the rest stays 0
            x0, y0, x1, y1 = rects[rank]
            ok = ok and bool((vis[y0:y1, x0:x1] == rank + 1).all().item())
            ex = expanded_rect(rects[rank], halo, fw, fh)
            outside = vis.clone()
            outside[ex[1]:ex[3], ex[0]:ex[2]] = 0
            ok = ok and not bool(outside.any().item())
            res["halo"] = ok
        if do_gather is not None:
            paint()
            do_gather()
            sync()
            ok = True
            for r, (x0, y0, x1, y1) in enumerate(rects):
                ok = ok and bool((vis[y0:y1, x0:x1] == r + 1).all().item())
            res["gather"] = ok
    finally:
        plane.data.copy_(saved)
        sync()
    return res


# ---- the stand-in reconstruction of the multi-GPU runs, and the check that survives the timed loop ----
# Step t turns a rank's own tile into ORIGINAL ^ ring_tag(t): four tags (a period longer than the
# two-plane ring), so a store that arrives a whole ring cycle late is as visible as one that is missing.
def ring_tag(t):
    return t & 3


def ring_delta(t):
    """what step t -> t + 1 xors into the tile: ring_tag(t) ^ ring_tag(t + 1)"""
    return (t & 3) ^ ((t + 1) & 3)


def visible(plane):
    return plane.data[plane.yorigin:plane.yorigin + plane.height, plane.xorigin:plane.xorigin + plane.width]


class TileRing:
    """The reconstruction as the tile-sharded runs model it (bench.py --gpus N, tests/test_distributed.py):
    a ring of two planes mapped on every rank (PeerPlanes); advance() writes this rank's tile of the
    NEXT plane (the stand-in reconstruction: the current tile ^ ring_delta(t)), stores its borders and
    then the tile into every peer's copy of that plane and hand-shakes; the next step's launches read
    that plane.  Nobody stores into a plane a peer may still be reading: the previous frame is the only
    reader of the other plane (the invariant a real encoder needs is "the destination is in no peer's
    live reference set": rav1e keeps up to 8 reference slots, so a pool of live slots + 1 planes, mapped
    once at start-up -- include/rav1e_amd.h, r1_comm_open_peer_planes).
    check(): every tile of the current plane, the peers' included, must be ORIGINAL ^ ring_tag(steps
    done) on this rank -- run it after the hand-shake of a step (bench.py runs it once more AFTER the
    timed loop: an ordering race that a pre-run exchange cannot see shows as a stale tag)."""

    def __init__(self, planes, peers, rects, rank, original_visible):
        assert len(planes) == 2 and len(peers) == 2
        self.planes, self.peers, self.rects, self.rank = planes, peers, rects, rank
        self.orig = original_visible           # device tensor: the visible area both planes started from
        x0, y0, x1, y1 = rects[rank]
        self.tiles = [visible(p)[y0:y1, x0:x1] for p in planes]
        self.cur, self.t = 0, 0

    def advance(self):
        nxt = self.cur ^ 1
        torch.bitwise_xor(self.tiles[self.cur], ring_delta(self.t), out=self.tiles[nxt])
        self.peers[nxt].push_frame(self.rects)      # halo stores, tile stores, ONE hand-shake
        self.cur, self.t = nxt, self.t + 1

    def advance_overlapped(self, side, halo=None, interior_work=None):
        """The same step with the exchange's first leg OVERLAPPED with the rest of the tile's work (VERDICT r5 item 8;
        the reference has no such step -- its tiles share memory -- this is the schedule a tile-per-GPU host wants):
          main stream:  the border of the tile (the stand-in reconstruction of the blocks along its edges) -> event E
          side stream:  waits for E, stores the border rectangles into the neighbours' planes (no hand-shake)
          main stream:  meanwhile the tile's interior (+ `interior_work()`: whatever else the step computes), then
                        waits for the side stream, stores the whole tile into every peer, ONE hand-shake.
        The halo stores and the interior's kernels run concurrently; what the hand-shake orders is unchanged, so
        check() holds after every step exactly as for advance().  begin_overlapped / finish_overlapped are the two
        halves for a caller whose other work sits between them (bench.py --overlap-exchange)."""
        self.begin_overlapped(side, halo)
        if interior_work is not None:
            interior_work()
        self.finish_overlapped()

    def begin_overlapped(self, side, halo=None):
        halo = POSTFILTER_HALO if halo is None else halo
        nxt = self.cur ^ 1
        cur_t, nxt_t = self.tiles[self.cur], self.tiles[nxt]
        h, w = cur_t.shape
        b = min(halo, h // 2, w // 2)
        d = ring_delta(self.t)
        main = torch.cuda.current_stream()
        # the border ring first: four strips
        for sl in ((slice(0, b), slice(None)), (slice(h - b, h), slice(None)), (slice(b, h - b), slice(0, b)),
                   (slice(b, h - b), slice(w - b, w))):
            torch.bitwise_xor(cur_t[sl], d, out=nxt_t[sl])
        e_border = torch.cuda.Event()
        e_border.record(main)
        side.wait_event(e_border)
        self.peers[nxt].push_halos_async(self.rects, halo, stream=side)
        self._e_halo = torch.cuda.Event()
        self._e_halo.record(side)
        self._b = b

    def finish_overlapped(self):
        nxt = self.cur ^ 1
        cur_t, nxt_t = self.tiles[self.cur], self.tiles[nxt]
        h, w = cur_t.shape
        b, d = self._b, ring_delta(self.t)
        # the interior beside the halo stores
        if h > 2 * b and w > 2 * b:
            torch.bitwise_xor(cur_t[b:h - b, b:w - b], d, out=nxt_t[b:h - b, b:w - b])
        torch.cuda.current_stream().wait_event(self._e_halo)
        self.peers[nxt].push_tile(self.rects)       # the tile into every peer + the step's ONE hand-shake
        self.cur, self.t = nxt, self.t + 1

    def check(self):
        return bool(torch.equal(visible(self.planes[self.cur]), torch.bitwise_xor(self.orig, ring_tag(self.t))))


def owned_rows(alloc_height, rank, world):
    """Contiguous slab of plane rows rank `rank` contributes: ceil split of the
    padded allocation (every rank sends the same count; the tail is zero-padded)."""
    rows = -(-alloc_height // world)
    lo = min(rank * rows, alloc_height)
    hi = min(lo + rows, alloc_height)
    return rows, lo, hi


def make_exchange_buffers(plane_data, rank, world):
    """(send, gathered) tensors for `exchange_rows`; plane_data is the 2-D
    (alloc_height, stride) tensor of a Plane."""
    rows, lo, hi = owned_rows(plane_data.shape[0], rank, world)
    send = torch.zeros((rows, plane_data.shape[1]), dtype=plane_data.dtype, device=plane_data.device)
    send[: hi - lo] = plane_data[lo:hi]
    gathered = torch.empty((world * rows, plane_data.shape[1]), dtype=plane_data.dtype,
                           device=plane_data.device)
    return send, gathered


def exchange_rows(send, gathered, group=None):
    """All-gather the per-rank row slabs: afterwards gathered[:alloc_height] is
    the whole plane on every rank."""
    dist.all_gather_into_tensor(gathered, send, group=group)
    return gathered


def shard_candidates(frame_w, frame_h, k, rank, world, **kw):
    """Rank `rank`'s candidate lists = the whole-frame lists restricted to its
    tile (uniform tiling, src/tiling/tiler.rs:56-150).  The union over ranks is
    exactly the unsharded list (tests/test_distributed.py)."""
    if world == 1:
        return W.speed6_ladder(frame_w, frame_h, k, **kw)
    rect = W.tile_rects(world, frame_w, frame_h)[rank]
    return W.speed6_ladder(frame_w, frame_h, k, rect=rect, **kw)


# ---- the widened rows across GPUs --------------------------------
def me_jobs_for_rank(n_tiles, n_refs, rank, world):
    """(tile index, reference index) pairs rank `rank` runs in its
    r1_estimate_tile_motion_batch call: jobs are independent (tiles are, and so
    are the reference frames inside estimate_tile_motion, src/me.rs:190-199), dealt
    round-robin.  No collective on the data path."""
    jobs = [(t, r) for r in range(n_refs) for t in range(n_tiles)]
    return jobs[rank::world]


def merge_me_stats(stats, group=None):
    """After every rank filled the MEStats of its own (tile, reference) jobs in a
    ZEROED frame array (int32 view, (refs, rows, cols, 2)): one all-reduce(SUM) makes
    the whole FrameMEStats set resident everywhere -- the tile rectangles are disjoint,
    so the sum is a merge.  4 MB per reference at 4K."""
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


POSTFILTER_HALO = 64   # rows.  The data dependence is 11 rows (a horizontal edge 4 rows outside a
                       # slab still changes its first two rows and reads 7 rows further out:
                       # 14-tap filter, src/deblock.rs:846); one SUPERBLOCK row keeps the slab's
                       # block array aligned with the partition tree (the edge tests take 4x4
                       # coordinates modulo the transform size, src/deblock.rs:1109,1205)


def postfilter_slab(plane_rows, rank, world, align=64):
    """Row slab [lo, hi) of a plane that rank `rank` deblocks / CDEFs after the
    reconstructed-frame all-gather, and the rows [lo - halo, hi + halo) it has to read:
    the filters are local (deblock: 7 rows, CDEF: 2 + 8-row direction blocks), so slabs
    with POSTFILTER_HALO rows of unfiltered context reproduce the whole-frame pass."""
    per = -(-(-(-plane_rows // world)) // align) * align
    lo = min(rank * per, plane_rows)
    hi = min(lo + per, plane_rows)
    return lo, hi, max(0, lo - POSTFILTER_HALO), min(plane_rows, hi + POSTFILTER_HALO)


# ---- tile-boundary exchange for the post filters ---------------------------
def _clip(r, fw, fh):
    return (max(r[0], 0), max(r[1], 0), min(r[2], fw), min(r[3], fh))


def _isect(a, b):
    r = (max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3]))
    return r if r[0] < r[2] and r[1] < r[3] else None


def expanded_rect(rect, halo, fw, fh):
    """the tile plus `halo` pixels of context on every side, clipped to the frame"""
    return _clip((rect[0] - halo, rect[1] - halo, rect[2] + halo, rect[3] + halo), fw, fh)


def tile_halo_plan(rects, rank, halo, fw, fh):
    """What rank `rank` (owner of rects[rank]) sends and receives so that every rank ends up with
    its tile plus a `halo`-pixel ring of its neighbours' reconstruction: (sends, recvs), lists of
    (peer, (x0, y0, x1, y1)).  A send is (my tile) ^ (peer's expanded tile), a receive is
    (peer's tile) ^ (my expanded tile) -- the peer computes the same rectangles from its side, so
    no sizes are negotiated.  With POSTFILTER_HALO = 64 on 4 x 2 4K tiles a rank moves the
    64-pixel borders it shares with up to 5 neighbours (~0.3 MB of 8-bit luma) instead of the
    9.4 MB frame."""
    mine = rects[rank]
    mine_ext = expanded_rect(mine, halo, fw, fh)
    sends, recvs = [], []
    for peer, r in enumerate(rects):
        if peer == rank:
            continue
        s = _isect(mine, expanded_rect(r, halo, fw, fh))
        v = _isect(r, mine_ext)
        if s is not None:
            sends.append((peer, s))
        if v is not None:
            recvs.append((peer, v))
    return sends, recvs


def exchange_tile_halos(visible, rects, rank, halo=POSTFILTER_HALO, group=None):
    """visible: 2-D tensor of the frame's visible area in which rects[rank] holds this rank's
    unfiltered reconstruction.  Point-to-point exchange (RCCL send / recv on the GPUs, gloo in the
    tests) of the tile borders; afterwards expanded_rect(rects[rank]) is valid.  The post filters
    (deblock: 7 pixels either side of an edge, CDEF: 8x8 direction blocks + 3 taps) then run on
    the expanded tile and their output is exact inside the tile."""
    fh, fw = visible.shape
    sends, recvs = tile_halo_plan(rects, rank, halo, fw, fh)
    ops, landing = [], []
    for peer, (x0, y0, x1, y1) in sends:
        ops.append(dist.P2POp(dist.isend, visible[y0:y1, x0:x1].contiguous(), peer, group))
    for peer, (x0, y0, x1, y1) in recvs:
        buf = torch.empty((y1 - y0, x1 - x0), dtype=visible.dtype, device=visible.device)
        ops.append(dist.P2POp(dist.irecv, buf, peer, group))
        landing.append(((x0, y0, x1, y1), buf))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for (x0, y0, x1, y1), buf in landing:
        visible[y0:y1, x0:x1] = buf
    return visible
