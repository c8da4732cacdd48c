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
#include "cdef_common.hpp"

namespace {
using namespace r1cdef;

__device__ __forceinline__ int32_t constrain(int32_t diff, int32_t threshold, int32_t damping) {
  if (!threshold) return 0;
  int shift = damping - (31 - __clz(threshold));
  shift = shift < 0 ? 0 : shift;
  const int32_t ad = diff < 0 ? -diff : diff;
  int32_t mag = threshold - (ad >> shift);
  mag = mag < 0 ? 0 : (mag > ad ? ad : mag);
  return diff < 0 ? -mag : mag;
}

// one pixel (i, j) of a block; rd(yy, xx) returns the tap at block-relative
// (yy, xx) or CDEF_VERY_LARGE where the halo does not exist
template <typename RD>
__device__ __forceinline__ int32_t filter_pixel_rd(RD rd, int i, int j, int pri, int sec, int dir,
                                                   int damping, int coeff_shift) {
  // cdef_directions (cdef.rs:225-234) packed as nibbles (value + 2) so that the
  // lookup is two shifts instead of a dependent table load
  constexpr uint32_t DY0 = 0x33332221u, DX0 = 0x22233333u, DY1 = 0x44443210u, DX1 = 0x12344444u;
  auto dyx = [&](int d, int k, int &dy, int &dx) {
    const int sh = 4 * d;
    dy = (int)(((k == 0 ? DY0 : DY1) >> sh) & 0xf) - 2;
    dx = (int)(((k == 0 ? DX0 : DX1) >> sh) & 0xf) - 2;
  };
  const int32_t x = rd(i, j);
  int32_t sum = 0, mx = x, mn = x;
  const int odd = (pri >> coeff_shift) & 1;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int pri_tap = odd ? 3 : (k == 0 ? 4 : 2);
    const int sec_tap = k == 0 ? 2 : 1;
    int d0y, d0x, d1y, d1x, d2y, d2x;
    dyx(dir, k, d0y, d0x);
    dyx((dir + 2) & 7, k, d1y, d1x);
    dyx((dir + 6) & 7, k, d2y, d2x);
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

// taps straight from the input plane, halo availability from the edge flags
template <int BPP>
__device__ __forceinline__ int32_t filter_pixel(const uint8_t *in0, size_t istride_bytes, int i,
                                                int j, int xs, int ys, int pri, int sec, int dir,
                                                int damping, int coeff_shift, int edges) {
  auto rd = [&](int yy, int xx) -> int32_t {
    const bool ok = (yy >= 0 || (edges & HAVE_TOP)) && (yy < ys || (edges & HAVE_BOTTOM)) &&
                    (xx >= 0 || (edges & HAVE_LEFT)) && (xx < xs || (edges & HAVE_RIGHT));
    return ok ? ldpx<BPP>(in0 + (ptrdiff_t)yy * (ptrdiff_t)istride_bytes + (ptrdiff_t)xx * BPP)
              : VERY_LARGE;
  };
  return filter_pixel_rd(rd, i, j, pri, sec, dir, damping, coeff_shift);
}

template <int BPP>
__device__ __forceinline__ void stpx(uint8_t *p, int32_t v) {
  if constexpr (BPP == 1) *p = (uint8_t)v;
  else *(uint16_t *)p = (uint16_t)v;
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

// The 2 x 2 blocks of a workgroup share one LDS tile of the input plane (their
// region plus the 2-pixel halo), staged once with CDEF_VERY_LARGE where the
// frame ends.  A halo pixel is missing exactly when it lies outside
// [0, 8*floor(W/8)) x [0, 8*floor(H/8)) in luma units: that is what the
// reference's edge flags (`bx + 1 >= xavail >> 3`, first row / column of the
// frame; cdef.rs:441-459) say for every block at once.
template <int BPP, int XD, int YD>
__global__ __launch_bounds__(256) void k_cdef_frame(CdefFrameArgs a) {
  __shared__ int32_t part[4][128];
  __shared__ uint16_t tile[20 * 20];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // the plane's decimation is a template parameter: tile geometry, the index
  // arithmetic of the staging and the lane -> pixel map become constants
  constexpr int xs = 8 >> XD, ys = 8 >> YD;
  constexpr int TW = 2 * xs + 4, TH = 2 * ys + 4;
  const int gbx = blockIdx.x * 2 + (wave & 1), gby = blockIdx.y * 2 + (wave >> 1);
  const int fbx = gbx >> 3, fby = gby >> 3, bx = gbx & 7, by = gby & 7;
  const int mx = gbx * 2, my = gby * 2;
  const bool in_grid = gbx < a.nbx && gby < a.nby && mx < a.mi_cols && my < a.mi_rows;
  const int bd = a.prm.bit_depth, coeff_shift = bd - 8;
  const int in_xoff = fbx * 64, in_yoff = fby * 64;
  // ---- every global read of this block is issued here, before the barrier,
  // so that a wave pays one memory round trip, not five dependent ones ----
  int skip = 1, ci = 0;
  int32_t lum = 0;
  if (in_grid) {
    const uint8_t *sk = a.skip_mi + (size_t)my * a.mi_stride + mx;
    skip = sk[0] & sk[1] & sk[a.mi_stride] & sk[a.mi_stride + 1];
    ci = a.cdef_index_sb[fby * a.sb_stride + fbx];
    lum = ldpx<BPP>(px_addr<BPP>(a.luma, in_xoff + 8 * bx + (lane & 7), in_yoff + 8 * by + (lane >> 3)));
  }
  // plane position of the workgroup's region (no tile offset: whole frame)
  const int rx0 = (blockIdx.x * 2) * xs, ry0 = (blockIdx.y * 2) * ys;
  const int lim_x = ((a.luma.width >> 3) << 3) >> XD, lim_y = ((a.luma.height >> 3) << 3) >> YD;
  {
    // all loads of the tile first (a luma thread has two), then the LDS stores: one
    // memory round trip instead of one per loop iteration
    const uint8_t *p0 = (const uint8_t *)a.in.data;
    constexpr int NLD = (TW * TH + 255) / 256;
    int32_t tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const int t = threadIdx.x + 256 * k;
      const int ty = t / TW, tx = t - ty * TW;
      const int py = ry0 + ty - 2, px = rx0 + tx - 2;
      tv[k] = VERY_LARGE;
      if (t < TW * TH && py >= 0 && py < lim_y && px >= 0 && px < lim_x)
        tv[k] = ldpx<BPP>(p0 + ((size_t)(a.in.yorigin + py) * a.in.stride + a.in.xorigin + px) * BPP);
    }
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const int t = threadIdx.x + 256 * k;
      if (t < TW * TH) tile[t] = (uint16_t)tv[k];
    }
  }
  __syncthreads();
  if (!in_grid) return;
  const int px = (in_xoff >> XD) + bx * xs, py = (in_yoff >> YD) + by * ys;
  uint8_t *dst = (uint8_t *)px_addr<BPP>(a.out, px, py);
  const size_t dstb = (size_t)a.out.stride * BPP;
  const int i = lane / xs, j = lane % xs;
  const bool act = lane < xs * ys;
  // this block's top-left inside the tile
  const uint16_t *t0 = tile + ((wave >> 1) * ys + 2) * TW + (wave & 1) * xs + 2;
  if (skip) {   // wave-uniform
    if (act) stpx<BPP>(dst + i * dstb + j * BPP, t0[i * TW + j]);
    return;
  }
  // everything below the direction search is the same in all 64 lanes of the block: keep it in
  // scalar registers (strengths, direction, damping, the tap offsets they select)
  uint32_t var_v = 0;
  const int dir = __builtin_amdgcn_readfirstlane(find_dir_wave(lum, coeff_shift, part[wave], var_v));
  const uint32_t var = (uint32_t)__builtin_amdgcn_readfirstlane((int)var_v);
  ci = __builtin_amdgcn_readfirstlane(ci);
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
    // Cdef_Uv_Dir for 4:2:2: {7, 0, 2, 4, 5, 6, 6, 6} packed as nibbles
    const int uvdir = (int)((0x66654207u >> (4 * dir)) & 0xf);
    const int pri_uv = uvs / 4;
    int sec_uv = uvs % 4;
    sec_uv += sec_uv == 3;
    lpri = pri_uv << coeff_shift;
    lsec = sec_uv << coeff_shift;
    ldamp -= 1;
    ldir = pri_uv != 0 ? (XD != YD ? uvdir : dir) : 0;
  }
  if (act) {
    auto rd = [&](int yy, int xx) -> int32_t { return t0[yy * TW + xx]; };
    stpx<BPP>(dst + i * dstb + j * BPP,
              filter_pixel_rd(rd, i, j, lpri, lsec, ldir, ldamp, coeff_shift));
  }
}

template <int BPP>
__global__ __launch_bounds__(64) void k_cdef_find_dir(R1Plane luma, const R1CdefDirCand *cands,
                                                      int n, uint8_t *dir_out, int32_t *var_out) {
  __shared__ int32_t part[128];
  const int c = blockIdx.x;
  if (c >= n) return;
  uint32_t var;
  const int lane = threadIdx.x;
  const int32_t pix = ldpx<BPP>(px_addr<BPP>(luma, cands[c].x + (lane & 7), cands[c].y + (lane >> 3)));
  const int d = find_dir_wave(pix, luma.bit_depth - 8, part, var);
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
#define R1_CDEF_LAUNCH(B, X, Y) hipLaunchKernelGGL((k_cdef_frame<B, X, Y>), grid, dim3(256), 0, st, a)
#define R1_CDEF_DEC(B)                                   \
  do {                                                   \
    if (xdec == 0 && ydec == 0) R1_CDEF_LAUNCH(B, 0, 0); \
    else if (xdec == 1 && ydec == 1) R1_CDEF_LAUNCH(B, 1, 1); \
    else if (xdec == 1) R1_CDEF_LAUNCH(B, 1, 0);         \
    else R1_CDEF_LAUNCH(B, 0, 1);                        \
  } while (0)
  if (in->bytes_per_px == 1) R1_CDEF_DEC(1);
  else R1_CDEF_DEC(2);
#undef R1_CDEF_DEC
#undef R1_CDEF_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
