// comm.hip -- the one real exchange step of the path, behind the C ABI (RCCL over xGMI).
//
// Tiles are encoded independently (src/encoder.rs:3245-3257), one tile per GPU.  What travels:
//   r1_comm_exchange_halos   the tile-boundary reconstructed pixels the post filters of the
//                            neighbouring tiles read (deblock: 7 px either side of an edge,
//                            src/deblock.rs:878-934; CDEF: 2 px + 8x8 direction blocks,
//                            src/cdef.rs:161-194) -- point-to-point rectangles, grouped
//                            ncclSend / ncclRecv
//   r1_comm_allgather        every rank's rows of the filtered reconstruction, so that the next
//                            frame's motion search / compensation can read the whole reference
//                            (motion vectors are clamped to the frame, not the tile:
//                            src/me.rs:339-362) -- one ncclAllGather per plane
// A Rust host binds these like every other entry point; rendezvous = the 128-byte unique id of
// r1_comm_unique_id, carried by whatever channel the host has (rank 0 creates it).
#include <rccl/rccl.h>

#include <cstring>

#include <vector>

#include "common.hpp"

struct r1_comm {
  ncclComm_t nccl;
  int rank, world, device;
  uint8_t *pack;          // staging for strided rectangles (send side | receive side)
  size_t pack_bytes;
};

#define R1_NCCL_CHECK(expr)                                                        \
  do {                                                                             \
    ncclResult_t r_ = (expr);                                                      \
    if (r_ != ncclSuccess) {                                                       \
      r1_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r_)); \
      return R1_ECOMM;                                                             \
    }                                                                              \
  } while (0)

extern "C" int r1_comm_unique_id(uint8_t *id128) {
  R1_REQUIRE(id128);
  static_assert(sizeof(ncclUniqueId) == 128, "the ABI carries the id as 128 bytes");
  ncclUniqueId id;
  R1_NCCL_CHECK(ncclGetUniqueId(&id));
  memcpy(id128, &id, 128);
  return R1_OK;
}

extern "C" int r1_comm_create(r1_ctx *ctx, int rank, int world, const uint8_t *id128, r1_comm **out) {
  R1_REQUIRE(ctx && id128 && out && world >= 1 && rank >= 0 && rank < world);
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  int dev = 0;
  R1_HIP_CHECK(hipGetDevice(&dev));
  r1_comm *c = new r1_comm();
  c->rank = rank;
  c->world = world;
  c->device = dev;
  c->pack = nullptr;
  c->pack_bytes = 0;
  ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) {
    r1_set_error("ncclCommInitRank(rank %d of %d) -> %s", rank, world, ncclGetErrorString(r));
    delete c;
    return R1_ECOMM;
  }
  *out = c;
  return R1_OK;
}

extern "C" void r1_comm_destroy(r1_comm *c) {
  if (!c) return;
  (void)ncclCommDestroy(c->nccl);
  if (c->pack) (void)hipFree(c->pack);
  delete c;
}

extern "C" int r1_comm_rank(const r1_comm *c) { return c ? c->rank : -1; }
extern "C" int r1_comm_world(const r1_comm *c) { return c ? c->world : -1; }

// Every rank contributes `bytes_per_rank` bytes at `send`; afterwards recv holds the
// contributions in rank order (world * bytes_per_rank bytes).  For a plane: the rank's slab of
// whole rows, so recv is the plane allocation (rav1e_amd/tiles.py owned_rows).
extern "C" int r1_comm_allgather(r1_comm *c, const void *send, void *recv, size_t bytes_per_rank,
                                 void *stream) {
  R1_REQUIRE(c && send && recv);
  if (bytes_per_rank == 0) return R1_OK;
  R1_NCCL_CHECK(ncclAllGather(send, recv, bytes_per_rank, ncclUint8, c->nccl, (hipStream_t)stream));
  return R1_OK;
}

// Rectangles of a plane to and from peers: xfers[i] = {peer, dir (0 send / 1 receive), x0, y0,
// x1, y1} in plane pixels (visible-area coordinates).  Both sides compute matching rectangles
// from the tile grid (tile_halo_plan), so no sizes are negotiated.  Strided rectangles are
// packed into a staging buffer with 2-D copies on the same stream, all sends and receives go
// out as ONE group, received rectangles are unpacked afterwards.
extern "C" int r1_comm_exchange_halos(r1_comm *c, const R1Plane *plane, const R1HaloXfer *xfers, int n,
                                      void *stream) {
  R1_REQUIRE(c && plane && (n == 0 || xfers));
  if (n == 0) return R1_OK;
  hipStream_t st = (hipStream_t)stream;
  const int bpp = plane->bytes_per_px;
  std::vector<size_t> off(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const R1HaloXfer &x = xfers[i];
    R1_REQUIRE(x.peer >= 0 && x.peer < c->world && x.peer != c->rank);
    R1_REQUIRE(x.x0 < x.x1 && x.y0 < x.y1 && (x.dir == 0 || x.dir == 1));
    const size_t bytes = (size_t)(x.x1 - x.x0) * (x.y1 - x.y0) * bpp;
    off[i + 1] = off[i] + ((bytes + 255) & ~(size_t)255);
  }
  if (c->pack_bytes < off[n]) {
    if (c->pack) R1_HIP_CHECK(hipFree(c->pack));
    c->pack = nullptr;
    R1_HIP_CHECK(hipMalloc((void **)&c->pack, off[n] * 2));
    c->pack_bytes = off[n] * 2;
  }
  uint8_t *base = (uint8_t *)plane->data;
  const size_t pitch = (size_t)plane->stride * bpp;
  auto rect_ptr = [&](const R1HaloXfer &x) {
    return base + ((size_t)(plane->yorigin + x.y0) * plane->stride + (size_t)(plane->xorigin + x.x0)) * bpp;
  };
  for (int i = 0; i < n; i++) {
    const R1HaloXfer &x = xfers[i];
    if (x.dir != 0) continue;
    const size_t rb = (size_t)(x.x1 - x.x0) * bpp;
    R1_HIP_CHECK(hipMemcpy2DAsync(c->pack + off[i], rb, rect_ptr(x), pitch, rb, x.y1 - x.y0,
                                  hipMemcpyDeviceToDevice, st));
  }
  R1_NCCL_CHECK(ncclGroupStart());
  for (int i = 0; i < n; i++) {
    const R1HaloXfer &x = xfers[i];
    const size_t bytes = (size_t)(x.x1 - x.x0) * (x.y1 - x.y0) * bpp;
    if (x.dir == 0) R1_NCCL_CHECK(ncclSend(c->pack + off[i], bytes, ncclUint8, x.peer, c->nccl, st));
    else R1_NCCL_CHECK(ncclRecv(c->pack + off[i], bytes, ncclUint8, x.peer, c->nccl, st));
  }
  R1_NCCL_CHECK(ncclGroupEnd());
  for (int i = 0; i < n; i++) {
    const R1HaloXfer &x = xfers[i];
    if (x.dir != 1) continue;
    const size_t rb = (size_t)(x.x1 - x.x0) * bpp;
    R1_HIP_CHECK(hipMemcpy2DAsync(rect_ptr(x), pitch, c->pack + off[i], rb, rb, x.y1 - x.y0,
                                  hipMemcpyDeviceToDevice, st));
  }
  return R1_OK;
}

// The reference-frame all-gather on TILES: rank r owns rects[r] (x0, y0, x1, y1 in plane pixels)
// of `plane`; afterwards every rank's plane holds every tile.  Tiles are rectangles, so each
// rank packs its own into a contiguous slot (2-D copy), one ncclAllGather moves the slots
// (all the size of the largest tile), and the other ranks' tiles are unpacked into place.
extern "C" int r1_comm_allgather_tiles(r1_comm *c, const R1Plane *plane, const int32_t *rects4, void *stream) {
  R1_REQUIRE(c && plane && rects4);
  hipStream_t st = (hipStream_t)stream;
  const int bpp = plane->bytes_per_px;
  size_t slot = 0;
  for (int r = 0; r < c->world; r++) {
    const int32_t *q = rects4 + 4 * r;
    R1_REQUIRE(q[0] < q[2] && q[1] < q[3]);
    const size_t b = (size_t)(q[2] - q[0]) * (q[3] - q[1]) * bpp;
    if (b > slot) slot = b;
  }
  slot = (slot + 255) & ~(size_t)255;
  const size_t need = slot * (c->world + 1);
  if (c->pack_bytes < need) {
    if (c->pack) R1_HIP_CHECK(hipFree(c->pack));
    c->pack = nullptr;
    R1_HIP_CHECK(hipMalloc((void **)&c->pack, need));
    c->pack_bytes = need;
  }
  uint8_t *base = (uint8_t *)plane->data;
  const size_t pitch = (size_t)plane->stride * bpp;
  auto rect_ptr = [&](const int32_t *q) {
    return base + ((size_t)(plane->yorigin + q[1]) * plane->stride + (size_t)(plane->xorigin + q[0])) * bpp;
  };
  const int32_t *mine = rects4 + 4 * c->rank;
  uint8_t *send = c->pack, *recv = c->pack + slot;
  const size_t mrb = (size_t)(mine[2] - mine[0]) * bpp;
  R1_HIP_CHECK(hipMemcpy2DAsync(send, mrb, rect_ptr(mine), pitch, mrb, mine[3] - mine[1],
                                hipMemcpyDeviceToDevice, st));
  R1_NCCL_CHECK(ncclAllGather(send, recv, slot, ncclUint8, c->nccl, st));
  for (int r = 0; r < c->world; r++) {
    if (r == c->rank) continue;
    const int32_t *q = rects4 + 4 * r;
    const size_t rb = (size_t)(q[2] - q[0]) * bpp;
    R1_HIP_CHECK(hipMemcpy2DAsync(rect_ptr(q), pitch, recv + slot * r, rb, rb, q[3] - q[1],
                                  hipMemcpyDeviceToDevice, st));
  }
  return R1_OK;
}
