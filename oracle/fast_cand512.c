/*
 * oracle/fast_cand512.c -- TEST INFRASTRUCTURE (see r1_oracle.h): the AVX-512 leg of bench.py's CPU baseline.
 *
 * The candidate of fast_cand.c (put_8tap -> SAD -> SATD -> diff -> forward DCT_DCT) on 16 x i32
 * lanes: compiled with -march=x86-64-v4 (AVX-512 F / BW / DQ / VL / CD), called only when the host
 * CPU reports AVX-512 (r1o_fast512_available).  Same structure as the 8-lane file, twice as wide:
 *   * the 8-tap passes produce 16 outputs per step, the vertical pass of the 2-D case on packed i16
 *     pairs (vpmaddwd: two taps per multiply);
 *   * SATD: two 8x8 Hadamard tiles side by side in one vector;
 *   * the transform: 16 columns, then 16 rows per call of the SAME generated networks
 *     (fwd_tx_1d.inc, T = 16 lanes); 16 x 16 transposes in registers.
 * 8x8 blocks have only 8 columns: two candidates share a vector for SATD and the transform; their
 * predictions come from the 8-lane put_8tap of fast_cand.c, compiled here a second time under the
 * wider ISA (32 registers, VL encodings).
 * Like fast_cand.c this is a compiler-vectorised PORT of the reference's Rust loops, NOT rav1e's
 * hand-written avx512icl assembly (needs nasm, absent here); bench.py labels it so.
 * tests/test_oracle_fast.py holds it equal to the scalar restatement, value for value.
 *
 * Reference functions followed (through the scalar oracle): put_8tap src/mc.rs:250-353, get_sad
 * src/dist.rs:31-52, get_satd src/dist.rs:156-221, forward_transform src/transform/forward.rs:71-161.
 */
#ifndef WBODY
#include <immintrin.h>

/* the 8-lane file once more, under this unit's ISA, with its entry point renamed */
#define R1_FAST_ENTRY r1o_fast8v4_rdo_cand_batch
#include "fast_cand.c"

typedef int32_t v16si __attribute__((vector_size(64)));
typedef uint32_t v16su __attribute__((vector_size(64)));
typedef int16_t v16hi __attribute__((vector_size(32)));
typedef int32_t v16si_u __attribute__((vector_size(64), aligned(4)));
typedef int16_t v16hi_u __attribute__((vector_size(32), aligned(2)));
typedef uint8_t v16qu_u __attribute__((vector_size(16), aligned(1)));
typedef uint16_t v16hu_u __attribute__((vector_size(32), aligned(2)));

/* the generated 1-D networks on 16 lanes (function names get a suffix: the 8-lane instantiation
 * above already owns r1_fdct4 ...) */
#undef TX1D_FN
#undef TX_ADD
#undef TX_SUB
#undef TX_MUL
#undef TX_RSHIFT1
#undef TX_ADD_AVG
#undef TX_SUB_AVG
#define T T16
typedef v16si T16;
#define TX1D_FN static __attribute__((noinline, unused))
#define TX_ADD(a, b) ((T)((v16su)(a) + (v16su)(b)))
#define TX_SUB(a, b) ((T)((v16su)(a) - (v16su)(b)))
#define TX_MUL(a, m, s) (((T)((v16su)(a) * (uint32_t)(m) + (uint32_t)((1 << (s)) >> 1))) >> (s))
#define TX_RSHIFT1(a) (TX_SUB((a), (T)((a) < 0)) >> 1)
#define TX_ADD_AVG(a, b) (TX_ADD(a, b) >> 1)
#define TX_SUB_AVG(a, b) (TX_SUB(a, b) >> 1)
#define r1_fdct4 r1w_fdct4
#define r1_fdct8 r1w_fdct8
#define r1_fdct16 r1w_fdct16
#define r1_fdct32 r1w_fdct32
#define r1_fdct64 r1w_fdct64
#define r1_fdst_vii_4 r1w_fdst_vii_4
#define r1_fdst8 r1w_fdst8
#define r1_fdst16 r1w_fdst16
#define r1_fwht4 r1w_fwht4
#define r1_fidentity4 r1w_fidentity4
#define r1_fidentity8 r1w_fidentity8
#define r1_fidentity16 r1w_fidentity16
#define r1_fidentity32 r1w_fidentity32
#include "fwd_tx_1d.inc"
#undef T

typedef void (*wtx_fn)(T16 *);
static wtx_fn wdct_of(int n) { return n == 16 ? r1w_fdct16 : n == 32 ? r1w_fdct32 : r1w_fdct64; }

/* two 8x8 tiles side by side (lanes 0..7 and 8..15): transpose each in place */
static inline void transpose8x2(T16 *r) {
  T16 t[8], u[8];
  for (int i = 0; i < 4; i++) {
    t[2 * i] = __builtin_shuffle(r[2 * i], r[2 * i + 1],
                                 (v16si){0, 16, 1, 17, 4, 20, 5, 21, 8, 24, 9, 25, 12, 28, 13, 29});
    t[2 * i + 1] = __builtin_shuffle(r[2 * i], r[2 * i + 1],
                                     (v16si){2, 18, 3, 19, 6, 22, 7, 23, 10, 26, 11, 27, 14, 30, 15, 31});
  }
  for (int i = 0; i < 2; i++) {
    u[4 * i] = __builtin_shuffle(t[4 * i], t[4 * i + 2],
                                 (v16si){0, 1, 16, 17, 4, 5, 20, 21, 8, 9, 24, 25, 12, 13, 28, 29});
    u[4 * i + 1] = __builtin_shuffle(t[4 * i], t[4 * i + 2],
                                     (v16si){2, 3, 18, 19, 6, 7, 22, 23, 10, 11, 26, 27, 14, 15, 30, 31});
    u[4 * i + 2] = __builtin_shuffle(t[4 * i + 1], t[4 * i + 3],
                                     (v16si){0, 1, 16, 17, 4, 5, 20, 21, 8, 9, 24, 25, 12, 13, 28, 29});
    u[4 * i + 3] = __builtin_shuffle(t[4 * i + 1], t[4 * i + 3],
                                     (v16si){2, 3, 18, 19, 6, 7, 22, 23, 10, 11, 26, 27, 14, 15, 30, 31});
  }
  for (int i = 0; i < 4; i++) {
    r[i] = __builtin_shuffle(u[i], u[i + 4], (v16si){0, 1, 2, 3, 16, 17, 18, 19, 8, 9, 10, 11, 24, 25, 26, 27});
    r[i + 4] = __builtin_shuffle(u[i], u[i + 4], (v16si){4, 5, 6, 7, 20, 21, 22, 23, 12, 13, 14, 15, 28, 29, 30, 31});
  }
}

/* 16 x 16 i32 transpose out of four 8x8 ones: rows 0..7 hold [A | B], rows 8..15 [C | D]; every tile
 * is transposed in place (transpose8x2), then row i = [A^T_i | C^T_i] and row i + 8 = [B^T_i | D^T_i] */
static inline void transpose16(T16 *r) {
  transpose8x2(r);
  transpose8x2(r + 8);
  for (int i = 0; i < 8; i++) {
    const T16 top = r[i], bot = r[i + 8];
    r[i] = __builtin_shuffle(top, bot, (v16si){0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 22, 23});
    r[i + 8] = __builtin_shuffle(top, bot, (v16si){8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 28, 29, 30, 31});
  }
}

static inline void whadamard8(T16 *s) {   /* dist.rs:84-117's butterfly order on 8 vectors */
  T16 a[8], b[8];
  for (int k = 0; k < 4; k++) {
    a[2 * k] = s[2 * k] + s[2 * k + 1];
    a[2 * k + 1] = s[2 * k] - s[2 * k + 1];
  }
  b[0] = a[0] + a[2]; b[2] = a[0] - a[2];
  b[1] = a[1] + a[3]; b[3] = a[1] - a[3];
  b[4] = a[4] + a[6]; b[6] = a[4] - a[6];
  b[5] = a[5] + a[7]; b[7] = a[5] - a[7];
  for (int k = 0; k < 4; k++) {
    s[k] = b[k] + b[k + 4];
    s[k + 4] = b[k] - b[k + 4];
  }
}

/* SATD of an n x n residual (n a multiple of 16): pairs of 8x8 tiles */
static uint32_t wsatd_resid(const int16_t *d, int n) {
  v16su acc = {0};
  for (int cy = 0; cy < n; cy += 8)
    for (int cx = 0; cx < n; cx += 16) {
      T16 s[8];
      for (int y = 0; y < 8; y++)
        s[y] = __builtin_convertvector(*(const v16hi_u *)(d + (cy + y) * n + cx), v16si);
      whadamard8(s);     /* vertical */
      transpose8x2(s);
      whadamard8(s);     /* horizontal */
      for (int y = 0; y < 8; y++) {
        const T16 m = s[y] >> 31;
        acc += (v16su)((s[y] ^ m) - m);
      }
    }
  uint64_t sum = 0;
  for (int i = 0; i < 16; i++) sum += acc[i];
  return (uint32_t)((sum + 4) >> 3);
}

static inline T16 wshift(T16 a, int sh) {   /* av1_round_shift_array with bit = -sh */
  if (sh == 0) return a;
  if (sh > 0) return (T16)((v16su)a << (uint32_t)sh);
  return (a + ((1 << -sh) >> 1)) >> -sh;
}

/* n x n DCT_DCT (n = 16, 32, 64) of resid into the reference's coefficient order */
static void wfwd_dct2d(const int16_t *resid, void *out, int n, int bd, int coeff32) {
  const int cls = n == 16 ? 1 : n == 32 ? 2 : 3;
  const int8_t *shift = FAST_SHIFT[cls][(bd - 8) / 2];
  wtx_fn fn = wdct_of(n);
  static __thread int32_t lbt[64 * 64] __attribute__((aligned(64)));
  T16 c[64];
  for (int c0 = 0; c0 < n; c0 += 16) {
    for (int r = 0; r < n; r++)
      c[r] = wshift(__builtin_convertvector(*(const v16hi_u *)(resid + r * n + c0), v16si), shift[0]);
    fn(c);
    for (int r0 = 0; r0 < n; r0 += 16) {
      T16 blk[16];
      for (int j = 0; j < 16; j++) blk[j] = wshift(c[r0 + j], shift[1]);
      transpose16(blk);   /* blk[j] = column c0 + j, rows r0 .. r0 + 15 */
      for (int j = 0; j < 16; j++) *(T16 *)(lbt + (c0 + j) * n + r0) = blk[j];
    }
  }
  const int ostride = n < 32 ? n : 32, wc = ostride;
  for (int r0 = 0; r0 < n; r0 += 16) {
    for (int k = 0; k < n; k++) c[k] = *(const T16 *)(lbt + k * n + r0);
    fn(c);
    const size_t base = (size_t)(r0 >= 32) * ostride * wc;
    for (int k = 0; k < n; k++) {
      const T16 v = wshift(c[k], shift[2]);
      const size_t o = base + (size_t)n * (k & ~31) + (size_t)(k & 31) * ostride + (r0 & 31);
      if (coeff32) *(v16si_u *)((int32_t *)out + o) = v;
      else *(v16hi_u *)((int16_t *)out + o) = __builtin_convertvector(v, v16hi);
    }
  }
}

static inline T16 wclamp(T16 a, int32_t maxv) {
  const T16 hi = (T16){0} + maxv;
  a &= ~(a >> 31);
  const T16 over = a > hi;
  return (a & ~over) | (hi & over);
}

#define WBODY
#define PIX uint8_t
#define WPIXV v16qu_u
#define WFN(name) name##_w8
#define W8FN(name) name##_u8
#include "fast_cand512.c"
#undef PIX
#undef WPIXV
#undef WFN
#undef W8FN
#define PIX uint16_t
#define WPIXV v16hu_u
#define WFN(name) name##_w16
#define W8FN(name) name##_u16
#include "fast_cand512.c"
#undef PIX
#undef WPIXV
#undef WFN
#undef W8FN

int r1o_fast512_available(void) {
  return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
         __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
}

/* as r1o_fast_rdo_cand_batch; -2 when the host has no AVX-512 */
int r1o_fast512_rdo_cand_batch(const r1o_plane *org, const r1o_plane *ref, int n_px, int tx_size,
                               const r1o_rdo_cand *c, int n, int threads, uint32_t *sad_out,
                               uint32_t *satd_out, void *coeffs) {
  if (!r1o_fast512_available()) return -2;
  if (n_px != 8 && n_px != 16 && n_px != 32 && n_px != 64) return -1;
  if (r1o_tx_width(tx_size) != n_px || r1o_tx_height(tx_size) != n_px) return -1;
  if (org->bytes_per_px != ref->bytes_per_px) return -1;
  for (int i = 0; i < n; i++)
    if (c[i].tx_type != 0) return -1;
  if (threads < 1) threads = 1;
  if (n_px == 8) {
    const int np = (n + 1) / 2;
    if (org->bytes_per_px == 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
      for (int p = 0; p < np; p++) pair8_w8(org, ref, c, 2 * p, n - 2 * p >= 2 ? 2 : 1, sad_out, satd_out, coeffs);
    } else {
#pragma omp parallel for schedule(static) num_threads(threads)
      for (int p = 0; p < np; p++) pair8_w16(org, ref, c, 2 * p, n - 2 * p >= 2 ? 2 : 1, sad_out, satd_out, coeffs);
    }
    return 0;
  }
  if (org->bytes_per_px == 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < n; i++) cand_w8(org, ref, n_px, &c[i], i, sad_out, satd_out, coeffs);
  } else {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < n; i++) cand_w16(org, ref, n_px, &c[i], i, sad_out, satd_out, coeffs);
  }
  return 0;
}

#else /* WBODY: compiled once per pixel type */

/* 16 outputs of an 8-tap pass: taps[i] * p[i * step + 0..15], i32 lanes */
static inline v16si WFN(tap16)(const PIX *p, ptrdiff_t step, const int16_t *f) {
  v16si a = {0};
  for (int i = 0; i < 8; i++)
    a += __builtin_convertvector(*(const WPIXV *)(p + i * step), v16si) * (int32_t)f[i];
  return a;
}

/* put_8tap into a dense n x n block (n a multiple of 16), the four cases of mc.rs:268-351 */
static void WFN(put16)(PIX *dst, const PIX *src, ptrdiff_t ss, int n, int col_frac, int row_frac,
                       int mode_x, int mode_y, int bd) {
  const int16_t *yf = r1o_get_filter(mode_y, row_frac, n);
  const int16_t *xf = r1o_get_filter(mode_x, col_frac, n);
  const int32_t maxv = (1 << bd) - 1;
  const int ib = 4 - (bd == 12 ? 2 : 0);
  if (col_frac == 0 && row_frac == 0) {
    for (int r = 0; r < n; r++) memcpy(dst + r * n, src + r * ss, n * sizeof(PIX));
  } else if (col_frac == 0) {
    for (int r = 0; r < n; r++)
      for (int x = 0; x < n; x += 16) {
        const v16si a = (WFN(tap16)(src + (r - 3) * ss + x, ss, yf) + 64) >> 7;
        *(WPIXV *)(dst + r * n + x) = __builtin_convertvector(wclamp(a, maxv), WPIXV);
      }
  } else if (row_frac == 0) {
    const int s1 = 7 - ib;
    for (int r = 0; r < n; r++)
      for (int x = 0; x < n; x += 16) {
        v16si a = (WFN(tap16)(src + r * ss + x - 3, 1, xf) + ((1 << s1) >> 1)) >> s1;
        a = (a + ((1 << ib) >> 1)) >> ib;
        *(WPIXV *)(dst + r * n + x) = __builtin_convertvector(wclamp(a, maxv), WPIXV);
      }
  } else {
    int16_t mid[(64 + 7) * 64] __attribute__((aligned(64)));
    const int s1 = 7 - ib, s2 = 7 + ib;
    for (int r = 0; r < n + 7; r++)
      for (int x = 0; x < n; x += 16) {
        const v16si a = (WFN(tap16)(src + (r - 3) * ss + x - 3, 1, xf) + ((1 << s1) >> 1)) >> s1;
        *(v16hi_u *)(mid + r * n + x) = __builtin_convertvector(a, v16hi);
      }
    /* vertical pass on i16 pairs: rows r + 2j and r + 2j + 1 interleaved, vpmaddwd with (u[2j], u[2j+1]).
     * The in-lane interleaves (vpunpck*wd on 256 bits) split 16 columns as {0-3, 8-11} / {4-7, 12-15};
     * the two accumulators are put back in column order by one permute each way at the end. */
    __m256i tp[4];
    for (int j = 0; j < 4; j++)
      tp[j] = _mm256_set1_epi32((int32_t)(((uint32_t)(uint16_t)yf[2 * j + 1] << 16) | (uint16_t)yf[2 * j]));
    const __m256i rnd = _mm256_set1_epi32((1 << s2) >> 1);
    for (int r = 0; r < n; r++)
      for (int x = 0; x < n; x += 16) {
        __m256i lo = rnd, hi = rnd;
        for (int j = 0; j < 4; j++) {
          const __m256i m0 = _mm256_loadu_si256((const __m256i *)(mid + (r + 2 * j) * n + x));
          const __m256i m1 = _mm256_loadu_si256((const __m256i *)(mid + (r + 2 * j + 1) * n + x));
          lo = _mm256_add_epi32(lo, _mm256_madd_epi16(_mm256_unpacklo_epi16(m0, m1), tp[j]));
          hi = _mm256_add_epi32(hi, _mm256_madd_epi16(_mm256_unpackhi_epi16(m0, m1), tp[j]));
        }
        /* lo = columns {0-3, 8-11}, hi = {4-7, 12-15} as i32 */
        const __m256i c0 = _mm256_permute2x128_si256(lo, hi, 0x20);   /* 0-3, 4-7 */
        const __m256i c1 = _mm256_permute2x128_si256(lo, hi, 0x31);   /* 8-11, 12-15 */
        v16si a = (v16si)_mm512_inserti64x4(_mm512_castsi256_si512(c0), c1, 1);
        a = a >> s2;
        *(WPIXV *)(dst + r * n + x) = __builtin_convertvector(wclamp(a, maxv), WPIXV);
      }
  }
}

static void WFN(cand)(const r1o_plane *org, const r1o_plane *ref, int n, const r1o_rdo_cand *c, int i,
                      uint32_t *sad_out, uint32_t *satd_out, void *coeffs) {
  PIX pred[64 * 64] __attribute__((aligned(64)));
  int16_t resid[64 * 64] __attribute__((aligned(64)));
  const PIX *o = (const PIX *)org->data + (size_t)(org->yorigin + c->oy) * org->stride + org->xorigin + c->ox;
  const PIX *r = (const PIX *)ref->data + (size_t)(ref->yorigin + c->ry) * ref->stride + ref->xorigin + c->rx;
  WFN(put16)(pred, r, ref->stride, n, c->col_frac, c->row_frac, c->mode_x, c->mode_y, ref->bit_depth);
  v16su sadv = {0};
  for (int y = 0; y < n; y++) {
    const PIX *or_ = o + (size_t)y * org->stride;
    for (int x = 0; x < n; x += 16) {
      const v16si d = __builtin_convertvector(*(const WPIXV *)(or_ + x), v16si) -
                      __builtin_convertvector(*(const WPIXV *)(pred + y * n + x), v16si);
      *(v16hi_u *)(resid + y * n + x) = __builtin_convertvector(d, v16hi);
      const v16si m = d >> 31;
      sadv += (v16su)((d ^ m) - m);
    }
  }
  uint32_t sad = 0;
  for (int k = 0; k < 16; k++) sad += sadv[k];
  if (sad_out) sad_out[i] = sad;
  if (satd_out) satd_out[i] = wsatd_resid(resid, n);
  if (coeffs) {
    const int hbd = sizeof(PIX) == 2;
    wfwd_dct2d(resid, (uint8_t *)coeffs + (size_t)i * n * n * (hbd ? 4 : 2), n, org->bit_depth, hbd);
  }
}

/* TWO 8x8 candidates side by side (lanes 0..7 | 8..15): the prediction of each comes from the 8-lane
 * put_8tap of fast_cand.c (an 8x8 block has 8 columns), SATD and the transform run on both at once */
static void WFN(pair8)(const r1o_plane *org, const r1o_plane *ref, const r1o_rdo_cand *c, int i0, int npair,
                       uint32_t *sad_out, uint32_t *satd_out, void *coeffs) {
  PIX pred[2][64] __attribute__((aligned(64)));
  int16_t resid[8 * 16] __attribute__((aligned(64)));
  const int hbd = sizeof(PIX) == 2;
  for (int k = 0; k < 2; k++) {
    const r1o_rdo_cand *ck = &c[i0 + (k < npair ? k : 0)];     /* an odd tail computes its candidate twice */
    const PIX *o = (const PIX *)org->data + (size_t)(org->yorigin + ck->oy) * org->stride + org->xorigin + ck->ox;
    const PIX *r = (const PIX *)ref->data + (size_t)(ref->yorigin + ck->ry) * ref->stride + ref->xorigin + ck->rx;
    W8FN(put8)(pred[k], r, ref->stride, 8, ck->col_frac, ck->row_frac, ck->mode_x, ck->mode_y, ref->bit_depth);
    uint32_t sad = 0;
    for (int y = 0; y < 8; y++)
      for (int x = 0; x < 8; x++) {
        const int d = (int)o[(size_t)y * org->stride + x] - (int)pred[k][y * 8 + x];
        resid[y * 16 + k * 8 + x] = (int16_t)d;
        sad += (uint32_t)(d < 0 ? -d : d);
      }
    if (sad_out && k < npair) sad_out[i0 + k] = sad;
  }
  T16 s[8];
  if (satd_out) {
    for (int y = 0; y < 8; y++) s[y] = __builtin_convertvector(*(const v16hi *)(resid + y * 16), v16si);
    whadamard8(s);
    transpose8x2(s);
    whadamard8(s);
    v16su acc = {0};
    for (int y = 0; y < 8; y++) {
      const T16 m = s[y] >> 31;
      acc += (v16su)((s[y] ^ m) - m);
    }
    for (int k = 0; k < npair; k++) {
      uint64_t sum = 0;
      for (int j = 0; j < 8; j++) sum += acc[k * 8 + j];
      satd_out[i0 + k] = (uint32_t)((sum + 4) >> 3);
    }
  }
  if (coeffs) {
    const int8_t *shift = FAST_SHIFT[1][(org->bit_depth - 8) / 2];
    for (int y = 0; y < 8; y++)
      s[y] = wshift(__builtin_convertvector(*(const v16hi *)(resid + y * 16), v16si), shift[0]);
    r1w_fdct8(s);
    for (int y = 0; y < 8; y++) s[y] = wshift(s[y], shift[1]);
    transpose8x2(s);            /* s[k]: lane r (of each block) = column-pass output r of column k */
    r1w_fdct8(s);
    for (int k = 0; k < 8; k++) {
      const T16 v = wshift(s[k], shift[2]);
      for (int b = 0; b < npair; b++) {
        uint8_t *dst = (uint8_t *)coeffs + (size_t)(i0 + b) * 64 * (hbd ? 4 : 2);
        if (hbd) for (int r = 0; r < 8; r++) ((int32_t *)dst)[k * 8 + r] = v[b * 8 + r];
        else for (int r = 0; r < 8; r++) ((int16_t *)dst)[k * 8 + r] = (int16_t)v[b * 8 + r];
      }
    }
  }
}
#endif
