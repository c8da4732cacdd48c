// importance.hip -- update_block_importances on device (SURVEY.md 8f "N1";
// reference src/api/internal.rs:911-1068).
//
// After its SATD map (the map of r1_estimate_inter_costs) the reference hands,
// for every 8x8 importance block of the current frame,
//   amount = (intra_cost + future_importance) * (1 - inter_cost / intra_cost) / len
// to the up to four importance blocks of the REFERENCE frame that the block's
// motion-compensated position overlaps, weighted by overlap area -- a
// scatter-add in f32 whose additions happen in the raster order of the source
// blocks (top-left, top-right, bottom-left, bottom-right inside a block).
// f32 addition does not commute with reordering, so atomic adds would not
// reproduce the reference.  A counting sort by destination instead:
//   1  one thread per source block: the four (destination, amount * fraction)
//      pairs, every operation a single IEEE f32 operation (__f*_rn: the compiler
//      must not contract a * b + c, Rust does not); integer atomics COUNT the
//      pairs of every destination (order-independent);
//   2  exclusive prefix sum of the counts (three small kernels);
//   3  one thread per pair: claim a slot in its destination's segment (integer
//      atomic, any order) and store (pair index, value);
//   4  one thread per destination: order its segment by pair index -- that IS the
//      reference's order (source raster order, corner order) -- and accumulate
//      sequentially.  Segments hold about four pairs (insertion sort); a motion
//      field that sends thousands of blocks to one destination gets a heap sort.
// The result is bit-identical to the sequential loop for any motion field.
#include "common.hpp"

namespace {

constexpr long long U = 64;   // IMP_BLOCK_SIZE_IN_MV_UNITS: 8 pixels * 8 units per pixel
constexpr int SCAN_TILE = 1024;

struct ImpPair { uint32_t order; float val; };

__global__ __launch_bounds__(256) void k_imp_pairs(const uint32_t *__restrict__ intra_costs,
                                                   const float *__restrict__ future,
                                                   const uint32_t *__restrict__ inter_costs,
                                                   const int16_t *__restrict__ mvs, int w, int h,
                                                   float flen, uint32_t *__restrict__ dest,
                                                   float *__restrict__ vals,
                                                   uint32_t *__restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  const int x = i % w, y = i / w;
  const long long rx = (long long)x * U + mvs[2 * i + 1], ry = (long long)y * U + mvs[2 * i];
  const float inter = (float)inter_costs[i], intra = (float)intra_costs[i];
  float frac = 0.f;
  if (!(intra <= inter)) frac = __fsub_rn(1.f, __fdiv_rn(inter, intra));
  const float amount = __fdiv_rn(__fmul_rn(__fadd_rn(intra, future[i]), frac), flen);
  // floor to the block grid (the reference's `- (U - 1) if negative` before a truncating division)
  const long long tlx = (rx - (rx < 0 ? U - 1 : 0)) / U * U, tly = (ry - (ry < 0 ? U - 1 : 0)) / U * U;
  const long long ax0 = tlx + U - rx, ax1 = rx - tlx, ay0 = tly + U - ry, ay1 = ry - tly;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const long long dx = tlx / U + (c & 1), dy = tly / U + (c >> 1);
    const long long area = ((c & 1) ? ax1 : ax0) * ((c >> 1) ? ay1 : ay0);
    const bool in = dx >= 0 && dy >= 0 && dx < w && dy < h;
    const uint32_t d = in ? (uint32_t)(dy * w + dx) : 0xFFFFFFFFu;   // off-frame: dropped
    dest[4 * i + c] = d;
    vals[4 * i + c] = __fmul_rn(amount, __fdiv_rn((float)area, 4096.f));
    if (in) atomicAdd(&count[d], 1u);
  }
}

// exclusive prefix sum of count[n] -> offset[n]: tile sums, then tile scans (each adds the sums before it)
__global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t *__restrict__ count, int n,
                                                    uint32_t *__restrict__ tile_sum) {
  __shared__ uint32_t red[256];
  const int base = blockIdx.x * SCAN_TILE;
  uint32_t s = 0;
  for (int k = threadIdx.x; k < SCAN_TILE; k += 256)
    if (base + k < n) s += count[base + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *__restrict__ count, int n,
                                                    const uint32_t *__restrict__ tile_sum,
                                                    uint32_t *__restrict__ offset) {
  // thread t owns 4 consecutive entries of the tile: local serial scan + scan across threads.
  // The tile's own offset = the sum of the tile sums before it, formed here by the block itself
  // (a few hundred values): a separate one-thread scan kernel between the two launches cost 14 us.
  __shared__ uint32_t part[256];
  __shared__ uint32_t pre[256];
  {
    uint32_t ps = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) ps += tile_sum[t];
    pre[threadIdx.x] = ps;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
      if ((int)threadIdx.x < m) pre[threadIdx.x] += pre[threadIdx.x + m];
      __syncthreads();
    }
  }
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v[k] = base + k < n ? count[base + k] : 0;
    s += v[k];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int m = 1; m < 256; m <<= 1) {          // Hillis-Steele inclusive scan
    const uint32_t add = threadIdx.x >= m ? part[threadIdx.x - m] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = pre[0] + part[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < n) offset[base + k] = run;
    run += v[k];
  }
}

__global__ __launch_bounds__(256) void k_imp_scatter(const uint32_t *__restrict__ dest,
                                                     const float *__restrict__ vals, int n_pairs,
                                                     const uint32_t *__restrict__ offset,
                                                     uint32_t *__restrict__ fill,
                                                     ImpPair *__restrict__ seg) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t d = dest[p];
  if (d == 0xFFFFFFFFu) return;
  const uint32_t slot = offset[d] + atomicAdd(&fill[d], 1u);
  seg[slot].order = (uint32_t)p;
  seg[slot].val = vals[p];
}

__device__ void sift_down(ImpPair *a, int start, int end) {
  int root = start;
  while (2 * root + 1 <= end) {
    int child = 2 * root + 1;
    if (child + 1 <= end && a[child].order < a[child + 1].order) child++;
    if (a[root].order >= a[child].order) return;
    const ImpPair t = a[root];
    a[root] = a[child];
    a[child] = t;
    root = child;
  }
}

__global__ __launch_bounds__(256) void k_imp_accumulate(ImpPair *__restrict__ seg,
                                                        const uint32_t *__restrict__ offset,
                                                        const uint32_t *__restrict__ count,
                                                        int n_blocks, float *__restrict__ ref_imp) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= n_blocks) return;
  const int n = (int)count[d];
  if (n == 0) return;
  ImpPair *a = seg + offset[d];
  if (n <= 32) {
    for (int i = 1; i < n; i++) {
      const ImpPair t = a[i];
      int j = i - 1;
      while (j >= 0 && a[j].order > t.order) {
        a[j + 1] = a[j];
        j--;
      }
      a[j + 1] = t;
    }
  } else {   // heap sort, in place
    for (int s = (n - 2) / 2; s >= 0; s--) sift_down(a, s, n - 1);
    for (int e = n - 1; e > 0; e--) {
      const ImpPair t = a[0];
      a[0] = a[e];
      a[e] = t;
      sift_down(a, 0, e - 1);
    }
  }
  float acc = ref_imp[d];
  for (int i = 0; i < n; i++) acc = __fadd_rn(acc, a[i].val);
  ref_imp[d] = acc;
}

struct ImpScratch { size_t dest, vals, seg, count, fill, offset, tiles, total; int n_tiles; };

void layout(int n_blocks, ImpScratch &s) {
  const size_t np = (size_t)n_blocks * 4;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  s.n_tiles = (n_blocks + SCAN_TILE - 1) / SCAN_TILE;
  s.dest = 0;
  s.vals = s.dest + up(np * 4);
  s.seg = s.vals + up(np * 4);
  s.count = s.seg + up(np * sizeof(ImpPair));
  s.fill = s.count + up((size_t)n_blocks * 4);      // count and fill are zeroed together
  s.offset = s.fill + up((size_t)n_blocks * 4);
  s.tiles = s.offset + up((size_t)n_blocks * 4);
  s.total = s.tiles + up((size_t)s.n_tiles * 4);
}

}  // namespace

extern "C" long long r1_update_block_importances_scratch_bytes(int w_in_imp_b, int h_in_imp_b) {
  if (w_in_imp_b <= 0 || h_in_imp_b <= 0 || (long long)w_in_imp_b * h_in_imp_b > (1 << 28)) return -1;
  ImpScratch s;
  layout(w_in_imp_b * h_in_imp_b, s);
  return (long long)s.total;
}

extern "C" int r1_update_block_importances(r1_ctx *ctx, const uint32_t *intra_costs,
                                           const float *future_importances,
                                           const uint32_t *inter_costs, const int16_t *mvs,
                                           int w_in_imp_b, int h_in_imp_b, int len,
                                           float *ref_importances, void *scratch,
                                           long long scratch_bytes, void *stream) {
  R1_REQUIRE(ctx);
  R1_REQUIRE(w_in_imp_b >= 0 && h_in_imp_b >= 0 && len >= 1);
  R1_REQUIRE((long long)w_in_imp_b * h_in_imp_b <= (1 << 28));
  const int nb = w_in_imp_b * h_in_imp_b;
  if (nb == 0) return R1_OK;
  R1_REQUIRE(intra_costs && future_importances && inter_costs && mvs && ref_importances && scratch);
  ImpScratch s;
  layout(nb, s);
  R1_REQUIRE(scratch_bytes >= (long long)s.total);
  R1_REQUIRE(((uintptr_t)scratch & 255) == 0);
  hipStream_t st = (hipStream_t)stream;
  uint8_t *b = (uint8_t *)scratch;
  uint32_t *dest = (uint32_t *)(b + s.dest), *count = (uint32_t *)(b + s.count);
  uint32_t *fill = (uint32_t *)(b + s.fill), *offset = (uint32_t *)(b + s.offset);
  uint32_t *tiles = (uint32_t *)(b + s.tiles);
  float *vals = (float *)(b + s.vals);
  ImpPair *seg = (ImpPair *)(b + s.seg);
  R1_HIP_CHECK(hipMemsetAsync(count, 0, s.offset - s.count, st));   // count and fill
  const unsigned gb = (unsigned)((nb + 255) / 256), gp = (unsigned)((4 * (size_t)nb + 255) / 256);
  hipLaunchKernelGGL(k_imp_pairs, dim3(gb), dim3(256), 0, st, intra_costs, future_importances,
                     inter_costs, mvs, w_in_imp_b, h_in_imp_b, (float)len, dest, vals, count);
  hipLaunchKernelGGL(k_scan_tiles, dim3(s.n_tiles), dim3(256), 0, st, count, nb, tiles);
  hipLaunchKernelGGL(k_scan_apply, dim3(s.n_tiles), dim3(256), 0, st, count, nb, tiles, offset);
  hipLaunchKernelGGL(k_imp_scatter, dim3(gp), dim3(256), 0, st, dest, vals, 4 * nb, offset, fill, seg);
  hipLaunchKernelGGL(k_imp_accumulate, dim3(gb), dim3(256), 0, st, seg, offset, count, nb,
                     ref_importances);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
