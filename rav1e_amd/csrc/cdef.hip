// cdef.hip -- CDEF direction search and filter
// (reference: cdef_find_dir src/cdef.rs:84-143, constrain 146-159,
// cdef_filter_block 198-298, adjust_strength 313-321,
// cdef_analyze_superblock 340-373, cdef_filter_superblock / cdef_filter_tile
// 405-625; x86 dispatch src/asm/x86/cdef.rs).
//
// Mapping: one wave per 8x8 luma block position, lane = pixel.
//  direction  every lane adds its pixel (>> coeff_shift, - 128) into the 8x15
//             partial-sum table in LDS (ds_add: integer adds, order-free), the
//             eight direction costs are formed by lanes 0..7 and the first
//             maximum (cdef.rs:64-73) is taken across them.
//  filter     lane = pixel of the (8>>xdec)x(8>>ydec) block of the plane; the
//             twelve taps are read straight from the deblocked input plane
//             (L1/L2 resident: neighbouring blocks share their halos); taps in
//             a halo the `edges` flags do not grant read as CDEF_VERY_LARGE,
//             exactly like the reference's padded u16 tile (cdef.rs:161-196).
// The frame kernel fuses cdef_analyze_superblock + cdef_filter_superblock for
// one plane: skip test, direction, strength adjustment, edge flags, filter or
// copy.  Integer arithmetic throughout.
#include "common.hpp"

namespace {

constexpr int VERY_LARGE = 0x8000;
enum { HAVE_LEFT = 1, HAVE_RIGHT = 2, HAVE_TOP = 4, HAVE_BOTTOM = 8 };

template <int BPP>
__device__ __forceinline__ int32_t ldpx(const uint8_t *p) {
  if constexpr (BPP == 1) return *p;
  else return *(const uint16_t *)p;
}

// lane = pixel (i = lane >> 3, j = lane & 7); returns dir (all lanes), var via ref
template <int BPP>
__device__ __forceinline__ int find_dir_wave(const uint8_t *blk, size_t stride_bytes,
                                             int coeff_shift, int32_t *part /* [8*16] LDS */,
                                             uint32_t &var) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  for (int k = lane; k < 128; k += 64) part[k] = 0;
  __builtin_amdgcn_wave_barrier();
  const int32_t x = (ldpx<BPP>(blk + i * stride_bytes + j * BPP) >> coeff_shift) - 128;
  atomicAdd(&part[0 * 16 + i + j], x);
  atomicAdd(&part[1 * 16 + i + j / 2], x);
  atomicAdd(&part[2 * 16 + i], x);
  atomicAdd(&part[3 * 16 + 3 + i - j / 2], x);
  atomicAdd(&part[4 * 16 + 7 + i - j], x);
  atomicAdd(&part[5 * 16 + 3 - i / 2 + j], x);
  atomicAdd(&part[6 * 16 + j], x);
  atomicAdd(&part[7 * 16 + i / 2 + j], x);
  __builtin_amdgcn_wave_barrier();
  // lane d < 8 forms cost[d] (cdef.rs:110-133); DIV = 840 / n
  int32_t cost = 0;
  if (lane < 8) {
    const int32_t *p = part + lane * 16;
    auto sq = [&](int k) -> int32_t { return p[k] * p[k]; };
    if (lane == 2 || lane == 6) {
      for (int k = 0; k < 8; k++) cost += sq(k);
      cost *= 105;
    } else if (lane == 0 || lane == 4) {
      constexpr int32_t DIV[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
      for (int k = 0; k < 7; k++) cost += (sq(k) + sq(14 - k)) * DIV[k + 1];
      cost += sq(7) * 105;
    } else {
      constexpr int32_t DIV[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
      for (int k = 0; k < 5; k++) cost += sq(3 + k);
      cost *= 105;
      for (int k = 0; k < 3; k++) cost += (sq(k) + sq(10 - k)) * DIV[2 * k + 2];
    }
  }
  int best = 0;
  int32_t best_cost = __shfl(cost, 0, 64);
  int32_t costs[8];
#pragma unroll
  for (int d = 0; d < 8; d++) costs[d] = __shfl(cost, d, 64);
#pragma unroll
  for (int d = 1; d < 8; d++)
    if (costs[d] > best_cost) { best_cost = costs[d]; best = d; }
  int32_t orth = costs[0];
#pragma unroll
  for (int d = 1; d < 8; d++)
    if (d == ((best + 4) & 7)) orth = costs[d];
  var = (uint32_t)((best_cost - orth) >> 10);
  return best;
}

__device__ __forceinline__ int32_t constrain(int32_t diff, int32_t threshold, int32_t damping) {
  if (!threshold) return 0;
  int shift = damping - (31 - __clz(threshold));
  shift = shift < 0 ? 0 : shift;
  const int32_t ad = diff < 0 ? -diff : diff;
  int32_t mag = threshold - (ad >> shift);
  mag = mag < 0 ? 0 : (mag > ad ? ad : mag);
  return diff < 0 ? -mag : mag;
}

// one pixel (i, j) of the block whose top-left input pixel is `in0`
template <int BPP>
__device__ __forceinline__ int32_t filter_pixel(const uint8_t *in0, size_t istride_bytes, int i,
                                                int j, int xs, int ys, int pri, int sec, int dir,
                                                int damping, int coeff_shift, int edges) {
  constexpr int8_t D[8][2][2] = {{{-1, 1}, {-2, 2}}, {{0, 1}, {-1, 2}}, {{0, 1}, {0, 2}},
                                 {{0, 1}, {1, 2}},   {{1, 1}, {2, 2}},  {{1, 0}, {2, 1}},
                                 {{1, 0}, {2, 0}},   {{1, 0}, {2, -1}}};
  auto rd = [&](int yy, int xx) -> int32_t {
    const bool ok = (yy >= 0 || (edges & HAVE_TOP)) && (yy < ys || (edges & HAVE_BOTTOM)) &&
                    (xx >= 0 || (edges & HAVE_LEFT)) && (xx < xs || (edges & HAVE_RIGHT));
    return ok ? ldpx<BPP>(in0 + (ptrdiff_t)yy * (ptrdiff_t)istride_bytes + (ptrdiff_t)xx * BPP)
              : VERY_LARGE;
  };
  const int32_t x = rd(i, j);
  int32_t sum = 0, mx = x, mn = x;
  const int odd = (pri >> coeff_shift) & 1;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int pri_tap = odd ? 3 : (k == 0 ? 4 : 2);
    const int sec_tap = k == 0 ? 2 : 1;
    const int d0y = D[dir][k][0], d0x = D[dir][k][1];
    const int d1y = D[(dir + 2) & 7][k][0], d1x = D[(dir + 2) & 7][k][1];
    const int d2y = D[(dir + 6) & 7][k][0], d2x = D[(dir + 6) & 7][k][1];
    const int32_t p[2] = {rd(i + d0y, j + d0x), rd(i - d0y, j - d0x)};
#pragma unroll
    for (int t = 0; t < 2; t++) {
      sum += pri_tap * constrain(p[t] - x, pri, damping);
      if (p[t] != VERY_LARGE && p[t] > mx) mx = p[t];
      if (p[t] < mn) mn = p[t];
    }
    const int32_t s[4] = {rd(i + d1y, j + d1x), rd(i - d1y, j - d1x), rd(i + d2y, j + d2x),
                          rd(i - d2y, j - d2x)};
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (s[t] != VERY_LARGE && s[t] > mx) mx = s[t];
      if (s[t] < mn) mn = s[t];
      sum += sec_tap * constrain(s[t] - x, sec, damping);
    }
  }
  const int32_t v = x + ((8 + sum - (sum < 0)) >> 4);
  return v < mn ? mn : (v > mx ? mx : v);
}

template <int BPP>
__device__ __forceinline__ void stpx(uint8_t *p, int32_t v) {
  if constexpr (BPP == 1) *p = (uint8_t)v;
  else *(uint16_t *)p = (uint16_t)v;
}

__device__ __forceinline__ int adjust_strength(int strength, int var) {
  const int v6 = var >> 6;
  int i = 0;
  if (v6 != 0) {
    i = 31 - __clz(v6);
    i = i < 12 ? i : 12;
  }
  return var != 0 ? (strength * (4 + i) + 8) >> 4 : 0;
}

struct CdefFrameArgs {
  R1Plane luma, in, out;
  int p, xdec, ydec, tile_w, tile_h;
  const uint8_t *skip_mi;
  int mi_stride, mi_cols, mi_rows;
  const uint8_t *cdef_index_sb;
  int sb_stride;
  R1CdefParams prm;
  int nbx, nby;
};

template <int BPP>
__global__ __launch_bounds__(256) void k_cdef_frame(CdefFrameArgs a) {
  __shared__ int32_t part[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // 2x2 blocks per workgroup
  const int gbx = blockIdx.x * 2 + (wave & 1), gby = blockIdx.y * 2 + (wave >> 1);
  if (gbx >= a.nbx || gby >= a.nby) return;
  const int fbx = gbx >> 3, fby = gby >> 3, bx = gbx & 7, by = gby & 7;
  const int mx = gbx * 2, my = gby * 2;
  if (!(mx < a.mi_cols && my < a.mi_rows)) return;
  const int bd = a.prm.bit_depth, coeff_shift = bd - 8;
  // edge flags exactly as the by/bx loops of cdef_filter_superblock leave them
  const int in_xoff = fbx * 64, in_yoff = fby * 64;
  const int xavail = a.luma.width - in_xoff, yavail = a.luma.height - in_yoff;
  int edges = 0;
  if (fby > 0 || by > 0) edges |= HAVE_TOP;
  if (fbx > 0 || bx > 0) edges |= HAVE_LEFT;
  // BOTTOM survives row `by` iff no row r <= by had r + 1 >= yavail >> 3
  if (!(by + 1 >= (yavail >> 3))) edges |= HAVE_BOTTOM;
  if (!(bx + 1 >= (xavail >> 3))) edges |= HAVE_RIGHT;
  const uint8_t *sk = a.skip_mi + (size_t)my * a.mi_stride + mx;
  const int skip = sk[0] & sk[1] & sk[a.mi_stride] & sk[a.mi_stride + 1];
  const int xs = 8 >> a.xdec, ys = 8 >> a.ydec;
  const int px = (in_xoff >> a.xdec) + bx * xs, py = (in_yoff >> a.ydec) + by * ys;
  const uint8_t *src = px_addr<BPP>(a.in, px, py);
  uint8_t *dst = (uint8_t *)px_addr<BPP>(a.out, px, py);
  const size_t sst = (size_t)a.in.stride * BPP, dstb = (size_t)a.out.stride * BPP;
  const int i = lane / xs, j = lane % xs;
  const bool act = lane < xs * ys;
  if (skip) {   // wave-uniform
    if (act) stpx<BPP>(dst + i * dstb + j * BPP, ldpx<BPP>(src + i * sst + j * BPP));
    return;
  }
  uint32_t var = 0;
  const int dir = find_dir_wave<BPP>(px_addr<BPP>(a.luma, in_xoff + 8 * bx, in_yoff + 8 * by),
                                     (size_t)a.luma.stride * BPP, coeff_shift, part[wave], var);
  const int ci = a.cdef_index_sb[fby * a.sb_stride + fbx];
  const int ysr = a.prm.y_strengths[ci], uvs = a.prm.uv_strengths[ci];
  int lpri, lsec, ldir, ldamp = a.prm.damping + coeff_shift;
  if (a.p == 0) {
    const int pri_y = ysr / 4;
    int sec_y = ysr % 4;
    sec_y += sec_y == 3;
    lpri = adjust_strength(pri_y << coeff_shift, (int)var);
    lsec = sec_y << coeff_shift;
    ldir = pri_y != 0 ? dir : 0;
  } else {
    constexpr uint8_t UVDIR[8] = {7, 0, 2, 4, 5, 6, 6, 6};
    const int pri_uv = uvs / 4;
    int sec_uv = uvs % 4;
    sec_uv += sec_uv == 3;
    lpri = pri_uv << coeff_shift;
    lsec = sec_uv << coeff_shift;
    ldamp -= 1;
    ldir = pri_uv != 0 ? (a.xdec != a.ydec ? UVDIR[dir] : dir) : 0;
  }
  if (act)
    stpx<BPP>(dst + i * dstb + j * BPP,
              filter_pixel<BPP>(src, sst, i, j, xs, ys, lpri, lsec, ldir, ldamp, coeff_shift, edges));
}

template <int BPP>
__global__ __launch_bounds__(64) void k_cdef_find_dir(R1Plane luma, const R1CdefDirCand *cands,
                                                      int n, uint8_t *dir_out, int32_t *var_out) {
  __shared__ int32_t part[128];
  const int c = blockIdx.x;
  if (c >= n) return;
  uint32_t var;
  const int d = find_dir_wave<BPP>(px_addr<BPP>(luma, cands[c].x, cands[c].y),
                                   (size_t)luma.stride * BPP, luma.bit_depth - 8, part, var);
  if (threadIdx.x == 0) {
    dir_out[c] = (uint8_t)d;
    var_out[c] = (int32_t)var;
  }
}

template <int BPP>
__global__ __launch_bounds__(64) void k_cdef_filter(R1Plane in, R1Plane out, int xdec, int ydec,
                                                    const R1CdefBlockCand *cands, int n) {
  const int c = blockIdx.x;
  if (c >= n) return;
  const R1CdefBlockCand cd = cands[c];
  const int xs = 8 >> xdec, ys = 8 >> ydec;
  const int lane = threadIdx.x, i = lane / xs, j = lane % xs;
  if (lane >= xs * ys) return;
  const uint8_t *src = px_addr<BPP>(in, cd.x, cd.y);
  uint8_t *dst = (uint8_t *)px_addr<BPP>(out, cd.x, cd.y);
  const int32_t v = filter_pixel<BPP>(src, (size_t)in.stride * BPP, i, j, xs, ys, cd.pri_strength,
                                      cd.sec_strength, cd.dir, cd.damping, in.bit_depth - 8, cd.edges);
  stpx<BPP>(dst + (size_t)i * out.stride * BPP + j * BPP, v);
}

}  // namespace

extern "C" int r1_cdef_find_dir_batch(r1_ctx *ctx, const R1Plane *luma, const R1CdefDirCand *cands,
                                      int n, uint8_t *dir_out, int32_t *var_out, void *stream) {
  R1_REQUIRE(ctx && luma);
  R1_REQUIRE(luma->bytes_per_px == 1 || luma->bytes_per_px == 2);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && dir_out && var_out);
  hipStream_t st = (hipStream_t)stream;
  if (luma->bytes_per_px == 1)
    hipLaunchKernelGGL((k_cdef_find_dir<1>), dim3(n), dim3(64), 0, st, *luma, cands, n, dir_out, var_out);
  else
    hipLaunchKernelGGL((k_cdef_find_dir<2>), dim3(n), dim3(64), 0, st, *luma, cands, n, dir_out, var_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_cdef_filter_block_batch(r1_ctx *ctx, const R1Plane *in, const R1Plane *out,
                                          int xdec, int ydec, const R1CdefBlockCand *cands, int n,
                                          void *stream) {
  R1_REQUIRE(ctx && in && out);
  R1_REQUIRE(in->bytes_per_px == out->bytes_per_px);
  R1_REQUIRE(in->bytes_per_px == 1 || in->bytes_per_px == 2);
  R1_REQUIRE(in->data != out->data);   // the filter reads neighbours of other blocks
  R1_REQUIRE(xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands);
  hipStream_t st = (hipStream_t)stream;
  if (in->bytes_per_px == 1)
    hipLaunchKernelGGL((k_cdef_filter<1>), dim3(n), dim3(64), 0, st, *in, *out, xdec, ydec, cands, n);
  else
    hipLaunchKernelGGL((k_cdef_filter<2>), dim3(n), dim3(64), 0, st, *in, *out, xdec, ydec, cands, n);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_cdef_filter_frame_plane(r1_ctx *ctx, const R1Plane *luma, const R1Plane *in,
                                          const R1Plane *out, int p, int xdec, int ydec,
                                          int tile_w, int tile_h, const uint8_t *skip_mi,
                                          int mi_stride, int mi_cols, int mi_rows,
                                          const uint8_t *cdef_index_sb, int sb_stride,
                                          const R1CdefParams *params, void *stream) {
  R1_REQUIRE(ctx && luma && in && out && params && skip_mi && cdef_index_sb);
  R1_REQUIRE(in->bytes_per_px == out->bytes_per_px && in->bytes_per_px == luma->bytes_per_px);
  R1_REQUIRE(in->bytes_per_px == 1 || in->bytes_per_px == 2);
  R1_REQUIRE(in->data != out->data);
  R1_REQUIRE(p >= 0 && p <= 2 && xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  R1_REQUIRE(p != 0 || (xdec == 0 && ydec == 0));
  R1_REQUIRE(tile_w > 0 && tile_h > 0 && mi_stride >= mi_cols);
  R1_REQUIRE(params->bit_depth == 8 || params->bit_depth == 10 || params->bit_depth == 12);
  CdefFrameArgs a;
  a.luma = *luma; a.in = *in; a.out = *out;
  a.p = p; a.xdec = xdec; a.ydec = ydec; a.tile_w = tile_w; a.tile_h = tile_h;
  a.skip_mi = skip_mi; a.mi_stride = mi_stride; a.mi_cols = mi_cols; a.mi_rows = mi_rows;
  a.cdef_index_sb = cdef_index_sb; a.sb_stride = sb_stride;
  a.prm = *params;
  // fb loops run over ceil(tile / 64) superblocks x 8x8 block positions
  a.nbx = ((tile_w + 63) / 64) * 8;
  a.nby = ((tile_h + 63) / 64) * 8;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((a.nbx + 1) / 2, (a.nby + 1) / 2);
  if (in->bytes_per_px == 1) hipLaunchKernelGGL((k_cdef_frame<1>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_cdef_frame<2>), grid, dim3(256), 0, st, a);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
