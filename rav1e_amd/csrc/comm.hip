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
//
// RCCL is NOT linked: it is dlopen-ed on the first r1_comm_* call (rccl_api()).  A single-GPU user
// needs no RCCL to load librav1e_hip.so, and a process that already holds an RCCL (PyTorch ships
// its own librccl.so) gets THAT one -- one RCCL per process, chosen deliberately, instead of
// whichever of two copies the import order happened to bind (include/rav1e_amd.h, r1_comm_library).
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>   // types and enums only; every call goes through the table below

#include <cstring>

#include <string>
#include <vector>

#include "common.hpp"

namespace {
struct RcclApi {
  void *handle = nullptr;
  std::string path;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false;
};

int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
    *(std::string *)out = info->dlpi_name;
    return 1;
  }
  return 0;
}

// why no library could be bound: the loader's message for the LAST candidate that failed, kept from
// the one attempt (dlerror() is per thread and is cleared by the read: a later call would see nothing)
std::string g_rccl_why;

RcclApi *rccl_api() {
  static std::mutex mu;
  static RcclApi api;
  static bool tried = false;
  std::lock_guard<std::mutex> lk(mu);
  if (tried) return api.ok ? &api : nullptr;
  tried = true;
  std::vector<std::string> cands;
  if (const char *e = getenv("R1_RCCL_LIBRARY")) {
    cands.push_back(e);
  } else {
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) cands.push_back(loaded);
    cands.push_back("librccl.so.1");
    cands.push_back("/opt/rocm/lib/librccl.so.1");
  }
  for (const std::string &c : cands) {
    void *h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      const char *why = dlerror();
      g_rccl_why = c + ": " + (why ? why : "dlopen failed");
      continue;
    }
    RcclApi a;
    a.handle = h;
#define R1_SYM(field, name) *(void **)(&a.field) = dlsym(h, name)
    R1_SYM(GetUniqueId, "ncclGetUniqueId");
    R1_SYM(CommInitRank, "ncclCommInitRank");
    R1_SYM(CommDestroy, "ncclCommDestroy");
    R1_SYM(GetErrorString, "ncclGetErrorString");
    R1_SYM(AllGather, "ncclAllGather");
    R1_SYM(Send, "ncclSend");
    R1_SYM(Recv, "ncclRecv");
    R1_SYM(GroupStart, "ncclGroupStart");
    R1_SYM(GroupEnd, "ncclGroupEnd");
#undef R1_SYM
    if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GetErrorString && a.AllGather && a.Send && a.Recv &&
        a.GroupStart && a.GroupEnd) {
      // the path the loader actually resolved (a bare soname says little)
      struct link_map *lm = nullptr;
      a.path = (dlinfo(h, RTLD_DI_LINKMAP, &lm) == 0 && lm && lm->l_name && lm->l_name[0]) ? lm->l_name : c;
      a.ok = true;
      api = a;
      return &api;
    }
    g_rccl_why = c + ": an ncclXxx symbol is missing";
    dlclose(h);
  }
  return nullptr;
}
}  // namespace

struct r1_comm {
  ncclComm_t nccl;
  RcclApi *api;
  int rank, world, device;
  // staging for strided rectangles.  ONE buffer per communicator: the calls of a communicator
  // are collectives and have to be issued in the same order on every rank anyway; what the buffer
  // adds is that a call on a DIFFERENT stream than the previous one first waits (stream-side) for
  // that call's staging traffic (pack_done)
  uint8_t *pack;
  size_t pack_bytes;
  hipEvent_t pack_done;
  hipStream_t pack_stream;
  bool pack_used;
};

#define R1_NCCL_CHECK(api, expr)                                                   \
  do {                                                                             \
    ncclResult_t r_ = (expr);                                                      \
    if (r_ != ncclSuccess) {                                                       \
      r1_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (api)->GetErrorString(r_)); \
      return R1_ECOMM;                                                             \
    }                                                                              \
  } while (0)

#define R1_NEED_RCCL(api)                                                          \
  RcclApi *api = rccl_api();                                                       \
  if (!api) {                                                                      \
    r1_set_error("no RCCL library could be loaded (R1_RCCL_LIBRARY, a loaded librccl.so, librccl.so.1, " \
                 "/opt/rocm/lib/librccl.so.1): %s", g_rccl_why.c_str());                                   \
    return R1_ECOMM;                                                               \
  }

extern "C" const char *r1_comm_library(void) {
  RcclApi *a = rccl_api();
  return a ? a->path.c_str() : nullptr;
}

extern "C" int r1_comm_unique_id(uint8_t *id128) {
  R1_REQUIRE(id128);
  static_assert(sizeof(ncclUniqueId) == 128, "the ABI carries the id as 128 bytes");
  R1_NEED_RCCL(api);
  ncclUniqueId id;
  R1_NCCL_CHECK(api, api->GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return R1_OK;
}

extern "C" int r1_comm_create(r1_ctx *ctx, int rank, int world, const uint8_t *id128, r1_comm **out) {
  R1_REQUIRE(ctx && id128 && out && world >= 1 && rank >= 0 && rank < world);
  R1_NEED_RCCL(api);
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  // the communicator belongs to the context's device, whatever the calling thread's current one is
  R1DeviceGuard dev_guard(ctx);
  r1_comm *c = new r1_comm();
  c->api = api;
  c->rank = rank;
  c->world = world;
  c->device = ctx->device;
  c->pack = nullptr;
  c->pack_bytes = 0;
  c->pack_done = nullptr;
  c->pack_stream = nullptr;
  c->pack_used = false;
  if (hipEventCreateWithFlags(&c->pack_done, hipEventDisableTiming) != hipSuccess) {
    r1_set_error("r1_comm_create: hipEventCreate failed");
    delete c;
    return R1_EHIP;
  }
  ncclResult_t r = api->CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) {
    r1_set_error("ncclCommInitRank(rank %d of %d) -> %s", rank, world, api->GetErrorString(r));
    (void)hipEventDestroy(c->pack_done);
    delete c;
    return R1_ECOMM;
  }
  *out = c;
  return R1_OK;
}

extern "C" void r1_comm_destroy(r1_comm *c) {
  if (!c) return;
  int prev = -1;
  if (hipGetDevice(&prev) == hipSuccess && prev != c->device) (void)hipSetDevice(c->device);
  (void)c->api->CommDestroy(c->nccl);
  if (c->pack) (void)hipFree(c->pack);
  if (c->pack_done) (void)hipEventDestroy(c->pack_done);
  if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
  delete c;
}

extern "C" int r1_comm_rank(const r1_comm *c) { return c ? c->rank : -1; }
extern "C" int r1_comm_world(const r1_comm *c) { return c ? c->world : -1; }

namespace {
struct CommDeviceGuard {
  int prev = -1, want;
  explicit CommDeviceGuard(const r1_comm *c) : want(c->device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want) (void)hipSetDevice(want);
  }
  ~CommDeviceGuard() {
    if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
  }
};

// the staging buffer, at least `need` bytes, safe to touch from `st`: a previous call that staged
// on another stream is waited for on the stream side; growing it waits for that call on the host
int comm_staging(r1_comm *c, size_t need, hipStream_t st) {
  if (c->pack_used && c->pack_stream != st) R1_HIP_CHECK(hipStreamWaitEvent(st, c->pack_done, 0));
  if (c->pack_bytes < need) {
    if (c->pack_used) R1_HIP_CHECK(hipEventSynchronize(c->pack_done));
    if (c->pack) R1_HIP_CHECK(hipFree(c->pack));
    c->pack = nullptr;
    c->pack_bytes = 0;
    R1_HIP_CHECK(hipMalloc((void **)&c->pack, need));
    c->pack_bytes = need;
  }
  return R1_OK;
}
int comm_staging_done(r1_comm *c, hipStream_t st) {
  R1_HIP_CHECK(hipEventRecord(c->pack_done, st));
  c->pack_stream = st;
  c->pack_used = true;
  return R1_OK;
}
// From the first copy into the staging buffer to the end of the call: whatever way the call is
// left (an R1_HIP_CHECK return included), pack_done is recorded on the stream, so that a later call
// on another stream waits for the staging traffic already enqueued.
struct StagingScope {
  r1_comm *c;
  hipStream_t st;
  StagingScope(r1_comm *c_, hipStream_t st_) : c(c_), st(st_) {}
  ~StagingScope() { (void)comm_staging_done(c, st); }
};
// a rectangle in visible-area coordinates must lie inside the plane's visible area and the rows it
// touches inside the allocation
bool rect_in_plane(const R1Plane *p, int x0, int y0, int x1, int y1) {
  return x0 >= 0 && y0 >= 0 && x0 < x1 && y0 < y1 && x1 <= p->width && y1 <= p->height &&
         p->xorigin + x1 <= p->stride && p->yorigin + y1 <= p->alloc_height;
}
}  // namespace

// Every rank contributes `bytes_per_rank` bytes at `send`; afterwards recv holds the
// contributions in rank order (world * bytes_per_rank bytes).  For a plane: the rank's slab of
// whole rows, so recv is the plane allocation (rav1e_amd/tiles.py owned_rows).
extern "C" int r1_comm_allgather(r1_comm *c, const void *send, void *recv, size_t bytes_per_rank,
                                 void *stream) {
  R1_REQUIRE(c && send && recv);
  if (bytes_per_rank == 0) return R1_OK;
  CommDeviceGuard guard(c);
  R1_NCCL_CHECK(c->api, c->api->AllGather(send, recv, bytes_per_rank, ncclUint8, c->nccl, (hipStream_t)stream));
  return R1_OK;
}

// Rectangles of a plane to and from peers: xfers[i] = {peer, dir (0 send / 1 receive), x0, y0,
// x1, y1} in plane pixels (visible-area coordinates).  Both sides compute matching rectangles
// from the tile grid (tile_halo_plan), so no sizes are negotiated.  Strided rectangles are
// packed into a staging buffer with 2-D copies on the same stream, all sends and receives go
// out as ONE group, received rectangles are unpacked afterwards.
extern "C" int r1_comm_exchange_halos(r1_comm *c, const R1Plane *plane, const R1HaloXfer *xfers, int n,
                                      void *stream) {
  R1_REQUIRE(c && plane && plane->data && (n == 0 || xfers));
  if (n == 0) return R1_OK;
  hipStream_t st = (hipStream_t)stream;
  const int bpp = plane->bytes_per_px;
  R1_REQUIRE(bpp == 1 || bpp == 2);
  std::vector<size_t> off(n + 1, 0);
  for (int i = 0; i < n; i++) {
    const R1HaloXfer &x = xfers[i];
    R1_REQUIRE(x.peer >= 0 && x.peer < c->world && x.peer != c->rank);
    R1_REQUIRE(x.dir == 0 || x.dir == 1);
    R1_REQUIRE(rect_in_plane(plane, x.x0, x.y0, x.x1, x.y1));
    const size_t bytes = (size_t)(x.x1 - x.x0) * (x.y1 - x.y0) * bpp;
    off[i + 1] = off[i] + ((bytes + 255) & ~(size_t)255);
  }
  CommDeviceGuard guard(c);
  { const int rc = comm_staging(c, off[n], st); if (rc != R1_OK) return rc; }
  StagingScope staged(c, st);
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
  // the group is always closed, also when a call inside it fails: an open group would swallow
  // every later RCCL call of the process
  R1_NCCL_CHECK(c->api, c->api->GroupStart());
  ncclResult_t first = ncclSuccess;
  for (int i = 0; i < n && first == ncclSuccess; i++) {
    const R1HaloXfer &x = xfers[i];
    const size_t bytes = (size_t)(x.x1 - x.x0) * (x.y1 - x.y0) * bpp;
    first = x.dir == 0 ? c->api->Send(c->pack + off[i], bytes, ncclUint8, x.peer, c->nccl, st)
                       : c->api->Recv(c->pack + off[i], bytes, ncclUint8, x.peer, c->nccl, st);
  }
  const ncclResult_t end = c->api->GroupEnd();
  if (first != ncclSuccess || end != ncclSuccess) {
    r1_set_error("r1_comm_exchange_halos: %s", c->api->GetErrorString(first != ncclSuccess ? first : end));
    return R1_ECOMM;
  }
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
// rank packs its own into a contiguous slot (2-D copy) and the other ranks' tiles are unpacked
// into place.  What moves the slots:
//   p2p  (default)  one group of ncclSend (my tile to every peer) + ncclRecv (every peer's tile):
//        xGMI is point to point -- every GPU has its own link to each of the 7 others -- so the
//        N - 1 transfers of a rank run on N - 1 links at once, each carrying one tile's exact bytes;
//   ring ($R1_COMM_GATHER=ring)  one ncclAllGather of equal slots (the size of the largest tile):
//        RCCL's ring puts the N - 1 hops behind each other on one link per direction.
// Both orders of operations are the same on every rank (a collective).
extern "C" int r1_comm_allgather_tiles(r1_comm *c, const R1Plane *plane, const int32_t *rects4, void *stream) {
  R1_REQUIRE(c && plane && plane->data && rects4);
  hipStream_t st = (hipStream_t)stream;
  const int bpp = plane->bytes_per_px;
  R1_REQUIRE(bpp == 1 || bpp == 2);
  size_t slot = 0;
  for (int r = 0; r < c->world; r++) {
    const int32_t *q = rects4 + 4 * r;
    R1_REQUIRE(rect_in_plane(plane, q[0], q[1], q[2], q[3]));
    const size_t b = (size_t)(q[2] - q[0]) * (q[3] - q[1]) * bpp;
    if (b > slot) slot = b;
  }
  slot = (slot + 255) & ~(size_t)255;
  static const bool ring = [] {
    const char *e = getenv("R1_COMM_GATHER");
    return e && !strcmp(e, "ring");
  }();
  CommDeviceGuard guard(c);
  { const int rc = comm_staging(c, slot * (c->world + 1), st); if (rc != R1_OK) return rc; }
  StagingScope staged(c, st);
  uint8_t *base = (uint8_t *)plane->data;
  const size_t pitch = (size_t)plane->stride * bpp;
  auto rect_ptr = [&](const int32_t *q) {
    return base + ((size_t)(plane->yorigin + q[1]) * plane->stride + (size_t)(plane->xorigin + q[0])) * bpp;
  };
  auto rect_bytes = [&](const int32_t *q) { return (size_t)(q[2] - q[0]) * (q[3] - q[1]) * bpp; };
  const int32_t *mine = rects4 + 4 * c->rank;
  uint8_t *send = c->pack, *recv = c->pack + slot;
  const size_t mrb = (size_t)(mine[2] - mine[0]) * bpp;
  R1_HIP_CHECK(hipMemcpy2DAsync(send, mrb, rect_ptr(mine), pitch, mrb, mine[3] - mine[1],
                                hipMemcpyDeviceToDevice, st));
  if (ring) {
    const ncclResult_t r = c->api->AllGather(send, recv, slot, ncclUint8, c->nccl, st);
    if (r != ncclSuccess) {
      r1_set_error("r1_comm_allgather_tiles: %s", c->api->GetErrorString(r));
      return R1_ECOMM;
    }
  } else if (c->world > 1) {
    // the group is always closed (see r1_comm_exchange_halos)
    R1_NCCL_CHECK(c->api, c->api->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int rk = 0; rk < c->world && first == ncclSuccess; rk++) {
      if (rk == c->rank) continue;
      first = c->api->Send(send, rect_bytes(mine), ncclUint8, rk, c->nccl, st);
      if (first == ncclSuccess)
        first = c->api->Recv(recv + slot * rk, rect_bytes(rects4 + 4 * rk), ncclUint8, rk, c->nccl, st);
    }
    const ncclResult_t end = c->api->GroupEnd();
    if (first != ncclSuccess || end != ncclSuccess) {
      r1_set_error("r1_comm_allgather_tiles: %s", c->api->GetErrorString(first != ncclSuccess ? first : end));
      return R1_ECOMM;
    }
  }
  for (int rk = 0; rk < c->world; rk++) {
    if (rk == c->rank) continue;
    const int32_t *q = rects4 + 4 * rk;
    const size_t rb = (size_t)(q[2] - q[0]) * bpp;
    R1_HIP_CHECK(hipMemcpy2DAsync(rect_ptr(q), pitch, recv + slot * rk, rb, rb, q[3] - q[1],
                                  hipMemcpyDeviceToDevice, st));
  }
  return R1_OK;
}
