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
  // optional (r1_comm_barrier): a library without it still serves the exchange entry points
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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

// One attempt per process (positive and negative results are both cached); `why` receives the reason of a
// failure under the same lock that guards the attempt.
RcclApi *rccl_api_locked();
RcclApi *rccl_api(std::string *why = nullptr) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  RcclApi *a = rccl_api_locked();
  if (!a && why) *why = g_rccl_why;
  return a;
}
RcclApi *rccl_api_locked() {
  static RcclApi api;
  static bool tried = false;
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
    R1_SYM(AllReduce, "ncclAllReduce");
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
  int32_t *flag;   // r1_comm_barrier's two words
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
  std::string why_;                                                                \
  RcclApi *api = rccl_api(&why_);                                                  \
  if (!api) {                                                                      \
    r1_set_error("no RCCL library could be loaded (R1_RCCL_LIBRARY, a loaded librccl.so, librccl.so.1, " \
                 "/opt/rocm/lib/librccl.so.1): %s", why_.c_str());                                         \
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
  c->flag = nullptr;
  if (hipEventCreateWithFlags(&c->pack_done, hipEventDisableTiming) != hipSuccess) {
    r1_set_error("r1_comm_create: hipEventCreate failed");
    delete c;
    return R1_EHIP;
  }
  // r1_comm_barrier's two words: here, not on the barrier's first call (a hipMalloc + synchronous hipMemset in
  // the per-frame path, and a rank failing between them would have left its peers inside the all-reduce)
  if (hipMalloc((void **)&c->flag, 8) != hipSuccess || hipMemset(c->flag, 0, 8) != hipSuccess) {
    r1_set_error("r1_comm_create: the barrier's flag words could not be allocated");
    if (c->flag) (void)hipFree(c->flag);
    (void)hipEventDestroy(c->pack_done);
    delete c;
    return R1_EHIP;
  }
  ncclResult_t r = api->CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) {
    r1_set_error("ncclCommInitRank(rank %d of %d) -> %s", rank, world, api->GetErrorString(r));
    (void)hipEventDestroy(c->pack_done);
    (void)hipFree(c->flag);
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
  if (c->flag) (void)hipFree(c->flag);
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
// finish(): the same on the success path, with the record's own error handed to the caller (the destructor
// cannot return it).
struct StagingScope {
  r1_comm *c;
  hipStream_t st;
  bool done = false;
  StagingScope(r1_comm *c_, hipStream_t st_) : c(c_), st(st_) {}
  int finish() {
    done = true;
    return comm_staging_done(c, st);
  }
  ~StagingScope() {
    if (!done) (void)comm_staging_done(c, st);
  }
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
  return staged.finish();   // pack_done recorded: its own failure is the call's
}

// The reference-frame all-gather on TILES: rank r owns rects[r] (x0, y0, x1, y1 in plane pixels)
// of `plane`; afterwards every rank's plane holds every tile.  Tiles are rectangles, so each
// rank packs its own into a contiguous slot (2-D copy) and the other ranks' tiles are unpacked
// into place.  What moves the slots:
//   p2p  (default)  one group of ncclSend (my tile to every peer) + ncclRecv (every peer's tile):
//        xGMI is point to point -- every GPU has its own link to each of the 7 others -- so the
//        N - 1 transfers of a rank run on N - 1 links at once, each carrying one tile's exact bytes;
//   ring (-DR1_COMM_GATHER_RING=1)  one ncclAllGather of equal slots (the size of the largest tile):
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
#ifndef R1_COMM_GATHER_RING
#define R1_COMM_GATHER_RING 0   // A/B builds: -DR1_COMM_GATHER_RING=1
#endif
  constexpr bool ring = R1_COMM_GATHER_RING != 0;
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
  return staged.finish();
}

// ---- direct peer stores: a rank writes its rectangles straight into the peers' planes --------
// xGMI is a load / store fabric: a kernel on GPU a can store into GPU b's HBM once b's allocation
// is mapped here (hipIpcOpenMemHandle).  With every rank's plane mapped on every other rank, the
// tile gather is ONE kernel per rank that stores the finished tile into the N - 1 peers' planes
// (N - 1 links at once, no staging copy on either side, nothing to unpack), and the halo exchange
// is the same kernel on the border rectangles.  What RCCL keeps is the hand-shake: a 4-byte
// all-reduce (r1_comm_barrier) behind the stores -- it completes on a rank only after every rank's
// stream reached it, i.e. after every rank's store kernel finished, and the kernels a rank enqueues
// behind it start with the plane complete (kernel boundaries are where a GPU's L2s are written
// back and invalidated).  The DESTINATION must not be a plane a peer may still be reading: the
// reconstruction of frame k is a new buffer in the reference (src/encoder.rs:3322 hands it to the
// reference slots afterwards), so the host rotates two planes and needs no barrier before the stores
// (rav1e_amd/tiles.py PeerPlanes; bench.py).
namespace {
constexpr int R1_PUSH_MAX = 16;      // rectangles per launch (the kernel argument carries them)
constexpr uint32_t R1_PUSH_ROWS = 4; // rows of a rectangle per workgroup
struct PushRect {
  uint8_t *dst;
  const uint8_t *src;
  uint32_t row_bytes, rows, head, first_wg;
};
struct PushArgs {
  PushRect r[R1_PUSH_MAX];
  int n;
  uint32_t pitch;
};

__global__ __launch_bounds__(256) void k_push_rects(const PushArgs a) {
  int ri = 0;
  while (ri + 1 < a.n && blockIdx.x >= a.r[ri + 1].first_wg) ri++;
  const PushRect r = a.r[ri];
  const uint32_t row0 = (blockIdx.x - r.first_wg) * R1_PUSH_ROWS;
  const uint32_t nrows = min(R1_PUSH_ROWS, r.rows - row0);
  const uint8_t *src = r.src + (size_t)row0 * a.pitch;
  uint8_t *dst = r.dst + (size_t)row0 * a.pitch;
  // a row = `head` bytes up to the first 16-byte boundary (source and destination have the same
  // phase, the host checked), 16-byte words, the remaining bytes
  const uint32_t body = (r.row_bytes - r.head) >> 4;
  const uint32_t edge = r.row_bytes - (body << 4);   // head + tail bytes of a row
  for (uint32_t i = threadIdx.x; i < nrows * body; i += 256) {
    const uint32_t y = i / body, v = i - y * body;
    const size_t o = (size_t)y * a.pitch + r.head + ((size_t)v << 4);
    *(uint4 *)(dst + o) = *(const uint4 *)(src + o);
  }
  for (uint32_t i = threadIdx.x; i < nrows * edge; i += 256) {
    const uint32_t y = i / edge, e = i - y * edge;
    const size_t o = (size_t)y * a.pitch + (e < r.head ? e : (body << 4) + e);
    dst[o] = src[o];
  }
}

// peer mappings of this process.  An allocation is mapped ONCE however many exported ranges lie in
// it (two planes of a peer often share one allocator block, and a second hipIpcOpenMemHandle of the
// same handle is not something to rely on): `maps` holds handle -> base with a use count, `ptrs`
// the pointers handed out -> their map entry.
struct IpcMap {
  uint8_t handle[64];
  void *base;
  int device, uses;
};
std::mutex g_ipc_mu;
std::vector<IpcMap> g_ipc_maps;
struct IpcPtr {
  void *ptr, *base;   // pointer handed out, base of its mapping
  int device;         // the device of the context it was opened on
};
std::vector<IpcPtr> g_ipc_ptrs;
}  // namespace

static_assert(sizeof(hipIpcMemHandle_t) == 64, "R1IpcMem carries the handle as 64 bytes");

// `ptr` .. ptr + bytes (device memory of this process, any offset inside a hipMalloc allocation, a
// PyTorch caching-allocator block for instance) as 80 bytes another process can map.
extern "C" int r1_ipc_export(r1_ctx *ctx, const void *ptr, size_t bytes, R1IpcMem *out) {
  R1_REQUIRE(ctx && ptr && bytes && out);
  R1DeviceGuard guard(ctx);
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  R1_HIP_CHECK(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr));
  const size_t off = (const uint8_t *)ptr - (const uint8_t *)base;
  R1_REQUIRE(off + bytes <= size);
  hipIpcMemHandle_t h;
  R1_HIP_CHECK(hipIpcGetMemHandle(&h, base));
  memcpy(out->handle, &h, 64);
  out->offset = off;
  out->bytes = bytes;
  return R1_OK;
}

// Can kernels of the context's device address the memory of every other GPU this process sees?
// 1 yes, 0 no (hipDeviceCanAccessPeer said no for some device: do not map its memory), -1 unknown
// (this process sees one GPU only -- the peers' ordinals are not visible from here).
extern "C" int r1_ipc_peer_access(r1_ctx *ctx) {
  if (!ctx) return -1;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 2) return -1;
  for (int d = 0; d < n; d++) {
    if (d == ctx->device) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, ctx->device, d) != hipSuccess) return -1;
    if (!can) return 0;
  }
  return 1;
}

// Maps a peer process's exported memory on the context's device; *ptr addresses the exported
// range.  (Not for memory of the calling process: HIP refuses to open its own handles.)
extern "C" int r1_ipc_open(r1_ctx *ctx, const R1IpcMem *mem, void **ptr) {
  R1_REQUIRE(ctx && mem && ptr);
  R1DeviceGuard guard(ctx);
  hipIpcMemHandle_t h;
  memcpy(&h, mem->handle, 64);
  std::lock_guard<std::mutex> lk(g_ipc_mu);
  IpcMap *m = nullptr;
  for (IpcMap &e : g_ipc_maps)
    if (e.device == ctx->device && !memcmp(e.handle, mem->handle, 64)) m = &e;
  if (!m) {
    void *base = nullptr;
    R1_HIP_CHECK(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
    IpcMap e;
    memcpy(e.handle, mem->handle, 64);
    e.base = base;
    e.device = ctx->device;
    e.uses = 0;
    g_ipc_maps.push_back(e);
    m = &g_ipc_maps.back();
  }
  m->uses++;
  *ptr = (uint8_t *)m->base + mem->offset;
  g_ipc_ptrs.push_back(IpcPtr{*ptr, m->base, ctx->device});
  return R1_OK;
}

extern "C" int r1_ipc_close(r1_ctx *ctx, void *ptr) {
  R1_REQUIRE(ctx && ptr);
  void *base = nullptr;
  bool unmap = false;
  {
    std::lock_guard<std::mutex> lk(g_ipc_mu);
    for (size_t i = 0; i < g_ipc_ptrs.size(); i++)
      if (g_ipc_ptrs[i].ptr == ptr && g_ipc_ptrs[i].device == ctx->device) {
        base = g_ipc_ptrs[i].base;
        g_ipc_ptrs.erase(g_ipc_ptrs.begin() + i);
        break;
      }
    for (size_t i = 0; base && i < g_ipc_maps.size(); i++)
      if (g_ipc_maps[i].base == base && g_ipc_maps[i].device == ctx->device) {
        unmap = --g_ipc_maps[i].uses == 0;
        if (unmap) g_ipc_maps.erase(g_ipc_maps.begin() + i);
        break;
      }
  }
  if (!base) {
    r1_set_error("r1_ipc_close: %p was not returned by r1_ipc_open on device %d", ptr, ctx->device);
    return R1_EINVAL;
  }
  if (!unmap) return R1_OK;
  R1DeviceGuard guard(ctx);
  R1_HIP_CHECK(hipIpcCloseMemHandle(base));
  return R1_OK;
}

// rects[i]: the rectangle (visible-area pixels) of `plane` to store into peer_data[rects[i].peer]
// -- the data pointer of a plane with the SAME geometry (stride, origins, pixel size) on a peer,
// mapped with r1_ipc_open (or a second plane of this process).  One launch per 16 rectangles.
extern "C" int r1_push_rects(r1_ctx *ctx, const R1Plane *plane, void *const *peer_data, int n_peers,
                             const R1PushRect *rects, int n, void *stream) {
  R1_REQUIRE(ctx && plane && plane->data && (n == 0 || (rects && peer_data)) && n >= 0);
  const int bpp = plane->bytes_per_px;
  R1_REQUIRE(bpp == 1 || bpp == 2);
  const size_t pitch = (size_t)plane->stride * bpp;
  R1_REQUIRE(pitch < (1ull << 32));
  for (int i = 0; i < n; i++) {
    const R1PushRect &q = rects[i];
    R1_REQUIRE(q.peer >= 0 && q.peer < n_peers && peer_data[q.peer]);
    R1_REQUIRE(peer_data[q.peer] != plane->data);
    R1_REQUIRE(rect_in_plane(plane, q.x0, q.y0, q.x1, q.y1));
  }
  R1DeviceGuard guard(ctx);
  for (int i0 = 0; i0 < n; i0 += R1_PUSH_MAX) {
    PushArgs a;
    a.n = n - i0 < R1_PUSH_MAX ? n - i0 : R1_PUSH_MAX;
    a.pitch = (uint32_t)pitch;
    uint32_t wgs = 0;
    for (int j = 0; j < a.n; j++) {
      const R1PushRect &q = rects[i0 + j];
      const size_t o = ((size_t)(plane->yorigin + q.y0) * plane->stride + (size_t)(plane->xorigin + q.x0)) * bpp;
      PushRect &r = a.r[j];
      r.src = (const uint8_t *)plane->data + o;
      r.dst = (uint8_t *)peer_data[q.peer] + o;
      r.row_bytes = (uint32_t)(q.x1 - q.x0) * bpp;
      r.rows = (uint32_t)(q.y1 - q.y0);
      const bool same_phase = (((uintptr_t)r.src ^ (uintptr_t)r.dst) & 15) == 0 && (pitch & 15) == 0;
      const uint32_t to16 = (uint32_t)(-(intptr_t)(uintptr_t)r.src & 15);
      r.head = (!same_phase || to16 > r.row_bytes) ? r.row_bytes : to16;
      r.first_wg = wgs;
      wgs += (r.rows + R1_PUSH_ROWS - 1) / R1_PUSH_ROWS;
    }
    hipLaunchKernelGGL(k_push_rects, dim3(wgs), dim3(256), 0, (hipStream_t)stream, a);
    R1_HIP_CHECK(hipGetLastError());
  }
  return R1_OK;
}

// Every rank's `plane` (same geometry everywhere) mapped on every rank: peer_data[r] = rank r's
// plane as this rank addresses it (peer_data[rank] = plane->data itself).  The 80-byte exports
// travel in one all-gather.  Blocking (done once per plane, not per frame).
extern "C" int r1_comm_plane_pool_close(r1_comm *c, r1_ctx *ctx, int n_planes, void **peer_data);

// A POOL of planes (live reference slots + 1, see the header) mapped on every rank in ONE blocking collective:
// peer_data[p * world + r] = plane p of rank r as this rank addresses it ([.. + rank] = planes[p].data itself).
// The n_planes exports of a rank travel together in one all-gather of n_planes * 80 bytes per rank.
extern "C" int r1_comm_plane_pool_open(r1_comm *c, r1_ctx *ctx, const R1Plane *planes, int n_planes, void **peer_data) {
  R1_REQUIRE(c && ctx && planes && peer_data && n_planes >= 1 && n_planes <= 64 && ctx->device == c->device);
  for (int p = 0; p < n_planes; p++) R1_REQUIRE(planes[p].data);
  for (int i = 0; i < n_planes * c->world; i++) peer_data[i] = nullptr;
  for (int p = 0; p < n_planes; p++) peer_data[p * c->world + c->rank] = planes[p].data;
  if (c->world == 1) return R1_OK;
  // every rank takes part in the all-gather whatever failed locally (a rank that returned early
  // would leave the others inside the collective); a failed export travels as bytes == 0
  std::vector<R1IpcMem> mine(n_planes);
  std::vector<size_t> bytes(n_planes);
  int rc = R1_OK;
  for (int p = 0; p < n_planes; p++) {
    bytes[p] = (size_t)planes[p].stride * planes[p].alloc_height * planes[p].bytes_per_px;
    memset(&mine[p], 0, sizeof(R1IpcMem));
    const int e = r1_ipc_export(ctx, planes[p].data, bytes[p], &mine[p]);
    if (e != R1_OK) {
      memset(&mine[p], 0, sizeof(R1IpcMem));
      rc = e;
    }
  }
  CommDeviceGuard guard(c);
  const size_t slot = sizeof(R1IpcMem) * (size_t)n_planes;
  std::vector<R1IpcMem> all((size_t)c->world * n_planes);
  // the exports travel through the communicator's staging buffer (grown here if need be).  NOTHING returns
  // before the all-gather: if the buffer cannot be had or the upload fails, this rank still enters the
  // collective -- with whatever the buffer holds when there is one, its slot zeroed where possible (bytes == 0 =
  // "no export") -- and reports its error afterwards; only a rank with no device buffer at all cannot enter,
  // and that is an out-of-memory the peers see as an RCCL error / timeout rather than a silent hang here.
  const size_t need = slot * (size_t)(c->world + 1);
  hipError_t he = hipSuccess;
  if (c->pack_bytes < need) {
    if (c->pack_used) (void)hipEventSynchronize(c->pack_done);
    if (c->pack) (void)hipFree(c->pack);
    c->pack = nullptr;
    c->pack_bytes = 0;
    he = hipMalloc((void **)&c->pack, need);
    if (he == hipSuccess) c->pack_bytes = need;
  } else if (c->pack_used) {
    (void)hipEventSynchronize(c->pack_done);
  }
  ncclResult_t nr = ncclSuccess;
  if (c->pack) {
    uint8_t *dev = (uint8_t *)c->pack;
    uint8_t *send = dev + slot * (size_t)c->world;
    hipError_t up = hipMemcpy(send, mine.data(), slot, hipMemcpyHostToDevice);
    if (up != hipSuccess) {
      (void)hipMemset(send, 0, slot);   // best effort: travel as "no export"
      he = up;
    }
    nr = c->api->AllGather(send, dev, slot, ncclUint8, c->nccl, (hipStream_t) nullptr);
    hipError_t sy = nr == ncclSuccess ? hipStreamSynchronize(nullptr) : hipSuccess;
    if (nr == ncclSuccess && sy == hipSuccess)
      sy = hipMemcpy(all.data(), dev, slot * (size_t)c->world, hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = sy;
  }
  if (nr != ncclSuccess) {
    r1_set_error("r1_comm_plane_pool_open: %s", c->api->GetErrorString(nr));
    return R1_ECOMM;
  }
  if (he != hipSuccess) {
    r1_set_error("r1_comm_plane_pool_open: %s", hipGetErrorString(he));
    return R1_EHIP;
  }
  for (int r = 0; r < c->world && rc == R1_OK; r++) {
    if (r == c->rank) continue;
    for (int p = 0; p < n_planes && rc == R1_OK; p++) {
      const R1IpcMem &m = all[(size_t)r * n_planes + p];
      if (m.bytes != bytes[p]) {
        r1_set_error("r1_comm_plane_pool_open: rank %d exported %llu bytes for plane %d, this rank's plane has %llu", r,
                     (unsigned long long)m.bytes, p, (unsigned long long)bytes[p]);
        rc = R1_ECOMM;
      } else {
        rc = r1_ipc_open(ctx, &m, &peer_data[p * c->world + r]);
      }
    }
  }
  if (rc != R1_OK) (void)r1_comm_plane_pool_close(c, ctx, n_planes, peer_data);
  return rc;
}

extern "C" int r1_comm_plane_pool_close(r1_comm *c, r1_ctx *ctx, int n_planes, void **peer_data) {
  R1_REQUIRE(c && ctx && peer_data && n_planes >= 1);
  int rc = R1_OK;
  for (int p = 0; p < n_planes; p++) {
    const int e = r1_comm_close_peer_planes(c, ctx, peer_data + (size_t)p * c->world);
    if (e != R1_OK) rc = e;
  }
  return rc;
}

// one plane: the pool of one
extern "C" int r1_comm_open_peer_planes(r1_comm *c, r1_ctx *ctx, const R1Plane *plane, void **peer_data) {
  R1_REQUIRE(c && ctx && plane && plane->data && peer_data && ctx->device == c->device);
  return r1_comm_plane_pool_open(c, ctx, plane, 1, peer_data);
}

extern "C" int r1_comm_close_peer_planes(r1_comm *c, r1_ctx *ctx, void **peer_data) {
  R1_REQUIRE(c && ctx && peer_data);
  int rc = R1_OK;
  for (int r = 0; r < c->world; r++) {
    if (r != c->rank && peer_data[r]) {
      const int e = r1_ipc_close(ctx, peer_data[r]);
      if (e != R1_OK) rc = e;
    }
    peer_data[r] = nullptr;
  }
  return rc;
}

// Stream-ordered hand-shake: returns at once; the work enqueued on `stream` behind it starts only
// after every rank's stream reached its r1_comm_barrier (a 4-byte all-reduce).
extern "C" int r1_comm_barrier(r1_comm *c, void *stream) {
  R1_REQUIRE(c);
  if (c->world == 1) return R1_OK;
  if (!c->api->AllReduce) {
    r1_set_error("r1_comm_barrier: %s has no ncclAllReduce", c->api->path.c_str());
    return R1_ECOMM;
  }
  CommDeviceGuard guard(c);
  R1_NCCL_CHECK(c->api, c->api->AllReduce(c->flag, c->flag + 1, 1, ncclInt32, ncclSum, c->nccl, (hipStream_t)stream));
  return R1_OK;
}

// The tile gather by peer stores: this rank's tile (rects4[4 rank ..]) into every peer's plane,
// then the hand-shake.  peer_data from r1_comm_open_peer_planes for THIS plane.
extern "C" int r1_comm_push_tile(r1_comm *c, r1_ctx *ctx, const R1Plane *plane, void *const *peer_data,
                                 const int32_t *rects4, void *stream) {
  R1_REQUIRE(c && ctx && plane && peer_data && rects4);
  const int32_t *q = rects4 + 4 * c->rank;
  std::vector<R1PushRect> rects;
  for (int r = 0; r < c->world; r++)
    if (r != c->rank) rects.push_back(R1PushRect{r, q[0], q[1], q[2], q[3]});
  const int rc = r1_push_rects(ctx, plane, peer_data, c->world, rects.data(), (int)rects.size(), stream);
  return rc != R1_OK ? rc : r1_comm_barrier(c, stream);
}

// Both legs of a frame's exchange behind ONE hand-shake: the border rectangles first (the neighbours' post filters
// wait for nothing else), then the tile into every peer, then one r1_comm_barrier -- the all-reduce is the expensive
// part of a leg (a launch and a round trip over the fabric for 4 bytes), and nothing between the two legs needs it.
extern "C" int r1_comm_push_frame(r1_comm *c, r1_ctx *ctx, const R1Plane *plane, void *const *peer_data,
                                  const R1HaloXfer *xfers, int n, const int32_t *rects4, void *stream) {
  R1_REQUIRE(c && ctx && plane && peer_data && rects4 && (n == 0 || xfers));
  std::vector<R1PushRect> rects;
  for (int i = 0; i < n; i++) {
    R1_REQUIRE(xfers[i].dir == 0 || xfers[i].dir == 1);
    R1_REQUIRE(xfers[i].peer >= 0 && xfers[i].peer < c->world && xfers[i].peer != c->rank);
    if (xfers[i].dir == 0) rects.push_back(R1PushRect{xfers[i].peer, xfers[i].x0, xfers[i].y0, xfers[i].x1, xfers[i].y1});
  }
  // one list, the border rectangles in front: up to 8 neighbours + 7 peers fit ONE store launch (R1_PUSH_MAX = 16)
  const int32_t *q = rects4 + 4 * c->rank;
  for (int r = 0; r < c->world; r++)
    if (r != c->rank) rects.push_back(R1PushRect{r, q[0], q[1], q[2], q[3]});
  const int rc = r1_push_rects(ctx, plane, peer_data, c->world, rects.data(), (int)rects.size(), stream);
  return rc != R1_OK ? rc : r1_comm_barrier(c, stream);
}

// The halo exchange by peer stores: the dir == 0 (send) entries of the same list
// r1_comm_exchange_halos takes are stored into the peers' planes (the receives are the peers'
// sends), then the hand-shake.
extern "C" int r1_comm_push_halos(r1_comm *c, r1_ctx *ctx, const R1Plane *plane, void *const *peer_data,
                                  const R1HaloXfer *xfers, int n, void *stream) {
  R1_REQUIRE(c && ctx && plane && peer_data && (n == 0 || xfers));
  std::vector<R1PushRect> rects;
  for (int i = 0; i < n; i++) {
    R1_REQUIRE(xfers[i].dir == 0 || xfers[i].dir == 1);
    R1_REQUIRE(xfers[i].peer >= 0 && xfers[i].peer < c->world && xfers[i].peer != c->rank);
    if (xfers[i].dir == 0) rects.push_back(R1PushRect{xfers[i].peer, xfers[i].x0, xfers[i].y0, xfers[i].x1, xfers[i].y1});
  }
  const int rc = r1_push_rects(ctx, plane, peer_data, c->world, rects.data(), (int)rects.size(), stream);
  return rc != R1_OK ? rc : r1_comm_barrier(c, stream);
}
