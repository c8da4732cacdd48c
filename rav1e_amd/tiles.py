"""Tile sharding across GPUs and the one real exchange step of the path.

Tiles are encoded independently in the reference (separate TileStateMut per
tile, src/encoder.rs:3245-3257), so rank r owns tile r and evaluates that
tile's candidates with no data-path collective.  What every rank needs after a
frame is the whole reconstructed frame as the next reference (motion vectors
are clamped to the frame, not the tile: src/me.rs:339-362), i.e. one
all-gather of reconstructed rows per coded frame.  The
exchange works on any torch.distributed backend: RCCL ("nccl") on the GPUs,
gloo on CPU for the tests.
"""
import torch
import torch.distributed as dist

from . import workload as W


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
