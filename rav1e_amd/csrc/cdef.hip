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
  const uint8_t *dirs;    // [nby][nbx], k_cdef_analyze
  const int32_t *vars;
};

// ---- cdef_filter_superblock for one plane of the whole frame.
// Workgroup = 4 waves = a 32 x 16 pixel region of the plane, wave = 16 x 8 of it, LANE = TWO
// horizontally adjacent pixels held as one packed i16 pair: all of constrain(), the tap sum and
// the min / max tracking run on v_pk_* instructions, 12 (+ 1 to pair the two reads) per tap for
// two pixels.
//  * the region plus its halo (2 rows above / below, 4 columns left / right so that every global
//    load is one aligned 4-pixel group) is staged once in LDS as u16 with CDEF_VERY_LARGE where
//    the picture ends -- a halo pixel is missing exactly when it lies outside
//    [0, 8*floor(W/8)) x [0, 8*floor(H/8)) in luma units, which is what the reference's edge flags
//    say block by block (`bx + 1 >= xavail >> 3`, first row / column; cdef.rs:441-459);
//  * what depends on the 8x8 block only (skip, strengths after adjust_strength, damping shifts,
//    the six tap offsets of its direction) is worked out once per block by the first lanes of
//    wave 0 and left in LDS as a 16-dword record: the pixel lanes read it back with four
//    ds_read_b128 and unpack nothing;
//  * the LDS row stride is 24 dwords: the four rows a half-wave reads fall into disjoint bank
//    octets for any tap offset;
//  * 0x8000 is the smallest i16: the signed maximum ignores it, the unsigned minimum sees it as
//    large, and constrain() of it is 0 because (0x8000 >> shift) >= threshold for every legal
//    strength / damping -- the three things the reference does with CDEF_VERY_LARGE.
constexpr int CT_STRIDE = 48;           // u16 per tile row (40 used)
constexpr int CT_ROWS = 20, CT_X0 = 4, CT_Y0 = 2;
constexpr int CT_REC = 16;              // dwords per block record

template <int BPP, int XD, int YD>
__global__ __launch_bounds__(256) void k_cdef_frame(CdefFrameArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[CT_ROWS * CT_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t rec[32 * CT_REC];
  constexpr int xs = 8 >> XD, ys = 8 >> YD;
  constexpr int NBX = 32 / xs, NBY = 16 / ys, NB = NBX * NBY;     // blocks of the workgroup
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int cs = a.prm.bit_depth - 8;
  const int rx0 = blockIdx.x * 32, ry0 = blockIdx.y * 16;         // plane position of the region
  // ---- per-block records (wave 0; the loads overlap the tile's)
  if (tid < NB) {
    const int gbx = blockIdx.x * NBX + tid % NBX, gby = blockIdx.y * NBY + tid / NBX;
    const int mx = gbx * 2, my = gby * 2;
    const bool in_grid = gbx < a.nbx && gby < a.nby && mx < a.mi_cols && my < a.mi_rows;
    // every load of the record is issued here, together (one round trip for wave 0)
    int skip = 1, ci = 0, dir = 0, var = 0;
    if (in_grid) {
      const uint8_t *sk = a.skip_mi + (size_t)my * a.mi_stride + mx;
      const uint32_t s0 = *(const uint16_t *)sk, s1 = *(const uint16_t *)(sk + a.mi_stride);   // mx is even
      ci = a.cdef_index_sb[(gby >> 3) * a.sb_stride + (gbx >> 3)];
      dir = a.dirs[(size_t)gby * a.nbx + gbx];
      var = a.vars[(size_t)gby * a.nbx + gbx];
      skip = r1cdef::skip4(s0, s1);
    }
    // strengths of the superblock's cdef_index out of the argument registers (no dependent load)
    uint32_t st8[2];
    __builtin_memcpy(st8, a.p == 0 ? a.prm.y_strengths : a.prm.uv_strengths, 8);
    const int strength = (int)(((ci & 4 ? st8[1] : st8[0]) >> (8 * (ci & 3))) & 0xff);
    const int pri_s = strength >> 2;
    int sec_s = strength & 3;
    sec_s += sec_s == 3;
    int lpri = 0, lsec = 0, ldir = 0, ldamp = a.prm.damping + cs;
    if (a.p != 0) ldamp -= 1;
    if (!skip) {
      lsec = sec_s << cs;
      if (a.p == 0) {
        lpri = adjust_strength(pri_s << cs, var);
        ldir = pri_s != 0 ? dir : 0;
      } else {
        // Cdef_Uv_Dir for 4:2:2: {7, 0, 2, 4, 5, 6, 6, 6} packed as nibbles
        const int uvdir = (int)((0x66654207u >> (4 * dir)) & 0xf);
        lpri = pri_s << cs;
        ldir = pri_s != 0 ? (XD != YD ? uvdir : dir) : 0;
      }
    }
    auto shift_of = [&](int thr) {
      const int sft = thr ? ldamp - (31 - __clz(thr)) : 0;
      return sft < 0 ? 0 : sft;
    };
    const uint32_t psh = shift_of(lpri), ssh = shift_of(lsec);
    const int odd = (lpri >> cs) & 1;
    // cdef_directions (cdef.rs:225-234) packed as nibbles (value + 2)
    constexpr uint32_t DY0 = 0x33332221u, DX0 = 0x22233333u, DY1 = 0x44443210u, DX1 = 0x12344444u;
    auto off = [&](int d, int k) -> uint32_t {
      const int sh4 = 4 * (d & 7);
      const int dy = (int)(((k == 0 ? DY0 : DY1) >> sh4) & 0xf) - 2;
      const int dx = (int)(((k == 0 ? DX0 : DX1) >> sh4) & 0xf) - 2;
      return (uint32_t)((dy * CT_STRIDE + dx) * 2);
    };
    uint32_t *r = rec + tid * CT_REC;
    r[0] = (uint32_t)lpri * 0x10001u;
    r[1] = (uint32_t)lsec * 0x10001u;
    r[2] = psh * 0x10001u;
    r[3] = ssh * 0x10001u;
    r[4] = (odd ? 3u : 4u) * 0x10001u;
    r[5] = (odd ? 3u : 2u) * 0x10001u;
    r[6] = (in_grid ? 1u : 0u) | (skip ? 0u : 2u);
    r[7] = 0;
    r[8] = off(ldir, 0);
    r[9] = off(ldir, 1);
    r[10] = off(ldir + 2, 0);
    r[11] = off(ldir + 2, 1);
    r[12] = off(ldir + 6, 0);
    r[13] = off(ldir + 6, 1);
    r[14] = 0;
    r[15] = 0;
  }
  // ---- the tile: 20 rows x 10 groups of 4 pixels, one per thread
  {
    const int lim_x = ((a.luma.width >> 3) << 3) >> XD, lim_y = ((a.luma.height >> 3) << 3) >> YD;
    if (tid < CT_ROWS * 10) {
      const int ty = tid / 10, tq = tid - ty * 10;
      const int py = ry0 - CT_Y0 + ty, px = rx0 - CT_X0 + 4 * tq;
      uint32_t lo = 0x80008000u, hi = 0x80008000u;
      if (py >= 0 && py < lim_y && px >= 0 && px < lim_x) {
        const uint8_t *g = (const uint8_t *)a.in.data +
                           ((size_t)(a.in.yorigin + py) * a.in.stride + a.in.xorigin + px) * BPP;
        if constexpr (BPP == 1) {
          const uint32_t q = ld_u32(g);
          lo = __builtin_amdgcn_perm(0, q, 0x0c010c00u);
          hi = __builtin_amdgcn_perm(0, q, 0x0c030c02u);
        } else {
          const U32x2 q = ld_u32x2(g);
          lo = q.a;
          hi = q.b;
        }
      }
      *(uint2 *)(tile + ty * CT_STRIDE + 4 * tq) = make_uint2(lo, hi);
    }
  }
  __syncthreads();
  // ---- lane -> pixel pair
  const int r = lane >> 3, pc = lane & 7;
  const int lx = (wave & 1) * 16 + 2 * pc, ly = (wave >> 1) * 8 + r;     // inside the region
  const int blk = (ly / ys) * NBX + lx / xs;
  const uint4 q0 = *(const uint4 *)(rec + blk * CT_REC);
  const uint32_t flags = rec[blk * CT_REC + 6];
  const uint32_t base = (uint32_t)(((ly + CT_Y0) * CT_STRIDE + lx + CT_X0) * 2);
  Pk x;
  x.u = *(const uint32_t *)((const uint8_t *)tile + base);
  // lds_pair takes LDS byte addresses, not offsets into the tile
  typedef __attribute__((address_space(3))) uint16_t LdsU16;
  const uint32_t tbase = (uint32_t)(uintptr_t)(LdsU16 *)tile + base;
  if (!(flags & 1)) return;
  uint8_t *dst = (uint8_t *)px_addr<BPP>(a.out, rx0 + lx, ry0 + ly);
  auto store = [&](Pk v) {
    if constexpr (BPP == 1) *(uint16_t *)dst = (uint16_t)__builtin_amdgcn_perm(0, v.u, 0x0c0c0200u);
    else *(uint32_t *)dst = v.u;
  };
  if (!__any((int)(flags & 2))) {   // every block of the wave is skipped: copy
    store(x);
    return;
  }
  const uint4 q1 = *(const uint4 *)(rec + blk * CT_REC + 4);
  const uint4 q2 = *(const uint4 *)(rec + blk * CT_REC + 8);
  const uint2 q3 = *(const uint2 *)(rec + blk * CT_REC + 12);
  Pk pri, sec, psh, ssh, pt0, pt1;
  pri.u = q0.x; sec.u = q0.y; psh.u = q0.z; ssh.u = q0.w;
  pt0.u = q1.x; pt1.u = q1.y;
  const uint32_t offs[6] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
  // all 24 reads are issued before the first is used
  uint32_t tlo[12], thi[12], tp[12];
#pragma unroll
  for (int t = 0; t < 6; t++) {
    lds_two(tbase + offs[t], tlo[2 * t], thi[2 * t]);
    lds_two(tbase - offs[t], tlo[2 * t + 1], thi[2 * t + 1]);
  }
  lds_wait(tlo, thi);
#pragma unroll
  for (int t = 0; t < 12; t++) tp[t] = tlo[t] | (thi[t] << 16);
  i16x2 sum = (i16x2)0, mx = x.s;
  u16x2 mn = x.v;
#pragma unroll
  for (int t = 0; t < 12; t++) {
    Pk p;
    p.u = tp[t];
    const int dirsel = t >> 1;             // 0,1: primary k = 0,1; 2,3: dir + 2; 4,5: dir + 6
    const int k = dirsel & 1;
    mx = __builtin_elementwise_max(mx, p.s);
    mn = __builtin_elementwise_min(mn, p.v);
    if (dirsel < 2) {
      const i16x2 c = constrain2(p.s - x.s, pri.s, psh.v);
      sum += c * (k == 0 ? pt0.s : pt1.s);
    } else {
      const i16x2 c = constrain2(p.s - x.s, sec.s, ssh.v);
      sum += k == 0 ? c * (i16x2)2 : c;
    }
  }
  // x + ((8 + sum - (sum < 0)) >> 4), clamped to the taps' range
  Pk v;
  v.s = x.s + ((sum + (sum >> (i16x2)15) + (i16x2)8) >> (i16x2)4);
  v.s = __builtin_elementwise_min(__builtin_elementwise_max(v.s, (i16x2)mn), mx);
  store(v);
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

namespace {
int cdef_grid(int tile_w, int tile_h, int *nbx, int *nby) {
  // fb loops run over ceil(tile / 64) superblocks x 8x8 block positions
  *nbx = ((tile_w + 63) / 64) * 8;
  *nby = ((tile_h + 63) / 64) * 8;
  return *nbx * *nby;
}

int cdef_filter_launch(const CdefFrameArgs &a, int bpp, hipStream_t st) {
  // plane pixels the grid covers, in 32 x 16 regions
  const int pw = (a.nbx * 8) >> a.xdec, ph = (a.nby * 8) >> a.ydec;
  const dim3 grid((pw + 31) / 32, (ph + 15) / 16);
#define R1_CDEF_LAUNCH(B, X, Y) hipLaunchKernelGGL((k_cdef_frame<B, X, Y>), grid, dim3(256), 0, st, a)
#define R1_CDEF_DEC(B)                                   \
  do {                                                   \
    if (a.xdec == 0 && a.ydec == 0) R1_CDEF_LAUNCH(B, 0, 0); \
    else if (a.xdec == 1 && a.ydec == 1) R1_CDEF_LAUNCH(B, 1, 1); \
    else if (a.xdec == 1) R1_CDEF_LAUNCH(B, 1, 0);       \
    else R1_CDEF_LAUNCH(B, 0, 1);                        \
  } while (0)
  if (bpp == 1) R1_CDEF_DEC(1);
  else R1_CDEF_DEC(2);
#undef R1_CDEF_DEC
#undef R1_CDEF_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
}  // namespace

void r1_cdef_scratch_free(r1_ctx *c) {
  for (int k = 0; k < r1_ctx::kCdefSlots; k++) {
    if (c->cdef_done[k]) {
      (void)hipEventSynchronize(c->cdef_done[k]);
      (void)hipEventDestroy(c->cdef_done[k]);
    }
    if (c->cdef_scratch[k]) (void)hipFree(c->cdef_scratch[k]);
  }
}

extern "C" long long r1_cdef_analyze_blocks(int tile_w, int tile_h) {
  int nbx, nby;
  return tile_w > 0 && tile_h > 0 ? cdef_grid(tile_w, tile_h, &nbx, &nby) : 0;
}

extern "C" int r1_cdef_analyze_frame(r1_ctx *ctx, const R1Plane *luma, int tile_w, int tile_h,
                                     int mi_cols, int mi_rows, uint8_t *dir_out, int32_t *var_out,
                                     void *stream) {
  R1_REQUIRE(ctx && luma && dir_out && var_out);
  R1_REQUIRE(luma->bytes_per_px == 1 || luma->bytes_per_px == 2);
  R1_REQUIRE(luma->bit_depth == 8 || luma->bit_depth == 10 || luma->bit_depth == 12);
  R1_REQUIRE(tile_w > 0 && tile_h > 0);
  R1DeviceGuard guard(ctx);
  int nbx, nby;
  cdef_grid(tile_w, tile_h, &nbx, &nby);
  return cdef_analyze_launch(luma, nbx, nby, mi_cols, mi_rows, dir_out, var_out, (hipStream_t)stream);
}

namespace {
int cdef_frame_plane(r1_ctx *ctx, const R1Plane *luma, const uint8_t *dirs, const int32_t *vars,
                     const R1Plane *in, const R1Plane *out, int p, int xdec, int ydec, int tile_w,
                     int tile_h, const uint8_t *skip_mi, int mi_stride, int mi_cols, int mi_rows,
                     const uint8_t *cdef_index_sb, int sb_stride, const R1CdefParams *params,
                     void *stream) {
  R1_REQUIRE(ctx && in && out && params && skip_mi && cdef_index_sb && (luma || (dirs && vars)));
  R1_REQUIRE(in->bytes_per_px == out->bytes_per_px);
  R1_REQUIRE(!luma || in->bytes_per_px == luma->bytes_per_px);
  R1_REQUIRE(in->bytes_per_px == 1 || in->bytes_per_px == 2);
  R1_REQUIRE(in->data != out->data);
  R1_REQUIRE(p >= 0 && p <= 2 && xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  R1_REQUIRE(p != 0 || (xdec == 0 && ydec == 0));
  R1_REQUIRE(tile_w > 0 && tile_h > 0 && mi_stride >= mi_cols);
  R1_REQUIRE(params->bit_depth == 8 || params->bit_depth == 10 || params->bit_depth == 12);
  // the analysis takes its coefficient shift from the luma plane, the filter from params: one value
  R1_REQUIRE(!luma || luma->bit_depth == params->bit_depth);
  // without a luma plane the picture limits come from tile_w / tile_h: the plane size as v_frame
  // pads it (a multiple of 8), not fi.width -- the last 8x8 block column would fall outside
  R1_REQUIRE(luma || (tile_w % 8 == 0 && tile_h % 8 == 0));
  R1DeviceGuard guard(ctx);
  hipStream_t st = (hipStream_t)stream;
  CdefFrameArgs a;
  a.in = *in; a.out = *out;
  // the picture's limits are the luma plane's; without one, the tile's (the whole frame)
  a.luma = luma ? *luma : *in;
  if (!luma) {
    a.luma.width = tile_w;
    a.luma.height = tile_h;
  }
  a.p = p; a.xdec = xdec; a.ydec = ydec; a.tile_w = tile_w; a.tile_h = tile_h;
  a.skip_mi = skip_mi; a.mi_stride = mi_stride; a.mi_cols = mi_cols; a.mi_rows = mi_rows;
  a.cdef_index_sb = cdef_index_sb; a.sb_stride = sb_stride;
  a.prm = *params;
  const int nb = cdef_grid(tile_w, tile_h, &a.nbx, &a.nby);
  if (dirs) {
    a.dirs = dirs;
    a.vars = vars;
    return cdef_filter_launch(a, in->bytes_per_px, st);
  }
  // directions of this call only, in a slot of the context's ring (not hipMallocAsync: the default
  // pool hands its memory back at every synchronisation and the next call pays a driver allocation)
  std::lock_guard<std::mutex> lock(ctx->cdef_mu);
  const int slot = ctx->cdef_next;
  ctx->cdef_next = (slot + 1) % r1_ctx::kCdefSlots;
  if (!ctx->cdef_done[slot]) R1_HIP_CHECK(hipEventCreateWithFlags(&ctx->cdef_done[slot], hipEventDisableTiming));
  else R1_HIP_CHECK(hipEventSynchronize(ctx->cdef_done[slot]));
  const size_t need = (size_t)nb * 5;
  if (ctx->cdef_scratch_bytes[slot] < need) {
    if (ctx->cdef_scratch[slot]) R1_HIP_CHECK(hipFree(ctx->cdef_scratch[slot]));
    ctx->cdef_scratch[slot] = nullptr;
    ctx->cdef_scratch_bytes[slot] = 0;
    R1_HIP_CHECK(hipMalloc(&ctx->cdef_scratch[slot], need));
    ctx->cdef_scratch_bytes[slot] = need;
  }
  int32_t *v = (int32_t *)ctx->cdef_scratch[slot];
  uint8_t *d = (uint8_t *)ctx->cdef_scratch[slot] + (size_t)nb * 4;
  int rc = cdef_analyze_launch(luma, a.nbx, a.nby, mi_cols, mi_rows, d, v, st);
  if (rc == R1_OK) {
    a.dirs = d;
    a.vars = v;
    rc = cdef_filter_launch(a, in->bytes_per_px, st);
  }
  R1_HIP_CHECK(hipEventRecord(ctx->cdef_done[slot], st));
  return rc;
}
}  // namespace

extern "C" int r1_cdef_filter_frame_plane(r1_ctx *ctx, const R1Plane *luma, const R1Plane *in,
                                          const R1Plane *out, int p, int xdec, int ydec,
                                          int tile_w, int tile_h, const uint8_t *skip_mi,
                                          int mi_stride, int mi_cols, int mi_rows,
                                          const uint8_t *cdef_index_sb, int sb_stride,
                                          const R1CdefParams *params, void *stream) {
  R1_REQUIRE(luma);
  return cdef_frame_plane(ctx, luma, nullptr, nullptr, in, out, p, xdec, ydec, tile_w, tile_h, skip_mi,
                          mi_stride, mi_cols, mi_rows, cdef_index_sb, sb_stride, params, stream);
}

extern "C" int r1_cdef_filter_frame_plane_dirs(r1_ctx *ctx, const uint8_t *dirs, const int32_t *vars,
                                               const R1Plane *in, const R1Plane *out, int p, int xdec,
                                               int ydec, int tile_w, int tile_h, const uint8_t *skip_mi,
                                               int mi_stride, int mi_cols, int mi_rows,
                                               const uint8_t *cdef_index_sb, int sb_stride,
                                               const R1CdefParams *params, void *stream) {
  R1_REQUIRE(dirs && vars);
  return cdef_frame_plane(ctx, nullptr, dirs, vars, in, out, p, xdec, ydec, tile_w, tile_h, skip_mi,
                          mi_stride, mi_cols, mi_rows, cdef_index_sb, sb_stride, params, stream);
}
