/*
 * oracle/fast_cand.c -- TEST INFRASTRUCTURE (see r1_oracle.h): the CPU-baseline leg of bench.py.
 *
 * The same candidate as r1o_rdo_cand_batch (batch.c) -- put_8tap -> SAD -> SATD -> diff ->
 * forward DCT_DCT -- written so that gcc emits SIMD for it: pixel types are compile-time
 * (this file includes itself once per pixel width), the 8-tap passes are unit-stride loops,
 * the Hadamard and the transform networks run on GCC vector-extension values of 8 x i32
 * (the 1-D networks are the SAME generated fwd_tx_1d.inc the scalar oracle uses, instantiated
 * with T = 8 lanes: 8 columns, then 8 rows, per call).  It is NOT the reference's hand-written
 * assembly (the .asm files under src/x86 need nasm, absent here): it is what a compiler gives the reference's
 * Rust loops with AVX2, and bench.py labels it so.  tests/test_oracle_fast.py holds it equal
 * to the scalar restatement, value for value.
 *
 * Reference functions followed (through the scalar oracle): put_8tap src/mc.rs:250-353,
 * get_sad src/dist.rs:31-52, get_satd src/dist.rs:156-221, forward_transform
 * src/transform/forward.rs:71-161.
 */
#ifndef FAST_BODY
#include <omp.h>
#include <string.h>

#include "r1_oracle.h"

typedef int32_t v8si __attribute__((vector_size(32)));
typedef uint32_t v8su __attribute__((vector_size(32)));
typedef int16_t v8hi __attribute__((vector_size(16)));
typedef int32_t v8si_u __attribute__((vector_size(32), aligned(4)));
typedef int16_t v8hi_u __attribute__((vector_size(16), aligned(2)));

typedef v8si T;
#define TX1D_FN static __attribute__((noinline, unused))
#define TX_ADD(a, b) ((T)((v8su)(a) + (v8su)(b)))
#define TX_SUB(a, b) ((T)((v8su)(a) - (v8su)(b)))
#define TX_MUL(a, m, s) (((T)((v8su)(a) * (uint32_t)(m) + (uint32_t)((1 << (s)) >> 1))) >> (s))
/* (a < 0) is -1 in a true lane: a + (a < 0 ? 1 : 0) == a - (a < 0) */
#define TX_RSHIFT1(a) (TX_SUB((a), (T)((a) < 0)) >> 1)
#define TX_ADD_AVG(a, b) (TX_ADD(a, b) >> 1)
#define TX_SUB_AVG(a, b) (TX_SUB(a, b) >> 1)
#include "fwd_tx_1d.inc"

typedef void (*vtx_fn)(T *);
static vtx_fn dct_of(int n) {
  return n == 4 ? r1_fdct4 : n == 8 ? r1_fdct8 : n == 16 ? r1_fdct16 : n == 32 ? r1_fdct32 : r1_fdct64;
}

static inline void transpose8(T *r) {
  /* 8x8 i32 transpose: three rounds of pairwise interleaves */
  T t[8], u[8];
  for (int i = 0; i < 4; i++) {
    t[2 * i] = __builtin_shuffle(r[2 * i], r[2 * i + 1], (v8si){0, 8, 1, 9, 4, 12, 5, 13});
    t[2 * i + 1] = __builtin_shuffle(r[2 * i], r[2 * i + 1], (v8si){2, 10, 3, 11, 6, 14, 7, 15});
  }
  for (int i = 0; i < 2; i++) {
    u[4 * i] = __builtin_shuffle(t[4 * i], t[4 * i + 2], (v8si){0, 1, 8, 9, 4, 5, 12, 13});
    u[4 * i + 1] = __builtin_shuffle(t[4 * i], t[4 * i + 2], (v8si){2, 3, 10, 11, 6, 7, 14, 15});
    u[4 * i + 2] = __builtin_shuffle(t[4 * i + 1], t[4 * i + 3], (v8si){0, 1, 8, 9, 4, 5, 12, 13});
    u[4 * i + 3] = __builtin_shuffle(t[4 * i + 1], t[4 * i + 3], (v8si){2, 3, 10, 11, 6, 7, 14, 15});
  }
  for (int i = 0; i < 4; i++) {
    r[i] = __builtin_shuffle(u[i], u[i + 4], (v8si){0, 1, 2, 3, 8, 9, 10, 11});
    r[i + 4] = __builtin_shuffle(u[i], u[i + 4], (v8si){4, 5, 6, 7, 12, 13, 14, 15});
  }
}

static inline void hadamard8_lanes(T *s) {
  /* dist.rs:84-117's butterfly order, on 8 vectors at once */
  T a[8], b[8];
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

/* SATD of an n x n residual (n a multiple of 8): 8x8 Hadamards, msb(8) = 3 */
static uint32_t satd_resid(const int16_t *d, int n) {
  v8su acc = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int cy = 0; cy < n; cy += 8)
    for (int cx = 0; cx < n; cx += 8) {
      T s[8];
      for (int y = 0; y < 8; y++)
        s[y] = __builtin_convertvector(*(const v8hi_u *)(d + (cy + y) * n + cx), v8si);
      hadamard8_lanes(s);   /* vertical */
      transpose8(s);
      hadamard8_lanes(s);   /* horizontal */
      for (int y = 0; y < 8; y++) {
        const T m = s[y] >> 31;
        acc += (v8su)((s[y] ^ m) - m);
      }
    }
  uint64_t sum = 0;
  for (int i = 0; i < 8; i++) sum += acc[i];
  return (uint32_t)((sum + 4) >> 3);
}

static const int8_t FAST_SHIFT[4][3][3] = {   /* FWD_TXFM_SHIFT_LS, forward_shared.rs:22-64 */
    {{3, 0, 0}, {2, 0, 1}, {0, 0, 3}},
    {{4, -1, 0}, {2, 0, 1}, {0, 0, 3}},
    {{4, -2, 0}, {2, 0, 0}, {0, 0, 2}},
    {{4, -1, -2}, {2, 0, -1}, {0, 0, 1}}};

static inline T vshift(T a, int sh) {   /* av1_round_shift_array with bit = -sh */
  if (sh == 0) return a;
  if (sh > 0) return (T)((v8su)a << (uint32_t)sh);
  return (a + ((1 << -sh) >> 1)) >> -sh;
}

/* n x n DCT_DCT of resid (row stride n) into the reference's coefficient order */
static void fwd_dct2d(const int16_t *resid, void *out, int n, int bd, int coeff32) {
  const int cls = n == 8 ? 1 : n == 16 ? 1 : n == 32 ? 2 : 3;
  const int8_t *shift = FAST_SHIFT[cls][(bd - 8) / 2];
  vtx_fn fn = dct_of(n);
  /* column pass on 8 columns at a time; lbt is the transpose: lbt[c][r] */
  static __thread int32_t lbt[64 * 64] __attribute__((aligned(32)));
  T c[64];
  for (int c0 = 0; c0 < n; c0 += 8) {
    for (int r = 0; r < n; r++)
      c[r] = vshift(__builtin_convertvector(*(const v8hi_u *)(resid + r * n + c0), v8si), shift[0]);
    fn(c);
    for (int r0 = 0; r0 < n; r0 += 8) {
      T blk[8];
      for (int j = 0; j < 8; j++) blk[j] = vshift(c[r0 + j], shift[1]);
      transpose8(blk);   /* blk[j] = column c0 + j, rows r0 .. r0 + 7 */
      for (int j = 0; j < 8; j++) *(T *)(lbt + (c0 + j) * n + r0) = blk[j];
    }
  }
  /* row pass on 8 rows at a time; the output is stored transposed in <= 32x32 chunks
   * (forward.rs:135-159): coefficient (r, c) lands at chunk + c' * ostride + (r & 31) */
  const int ostride = n < 32 ? n : 32, wc = ostride;
  for (int r0 = 0; r0 < n; r0 += 8) {
    for (int k = 0; k < n; k++) c[k] = *(const T *)(lbt + k * n + r0);
    fn(c);
    const size_t base = (size_t)(r0 >= 32) * ostride * wc;
    for (int k = 0; k < n; k++) {
      const T v = vshift(c[k], shift[2]);
      const size_t o = base + (size_t)n * (k & ~31) + (size_t)(k & 31) * ostride + (r0 & 31);
      if (coeff32) *(v8si_u *)((int32_t *)out + o) = v;
      else *(v8hi_u *)((int16_t *)out + o) = __builtin_convertvector(v, v8hi);
    }
  }
}

static inline int32_t rshift(int32_t v, int b) { return (v + ((1 << b) >> 1)) >> b; }
static inline int32_t clampi(int32_t v, int32_t hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

#define FAST_BODY
typedef uint8_t pixv_u8 __attribute__((vector_size(8), aligned(1)));
typedef uint16_t pixv_u16 __attribute__((vector_size(16), aligned(2)));
#define PIX uint8_t
#define FN(name) name##_u8
#include "fast_cand.c"
#undef PIX
#undef FN
#define PIX uint16_t
#define FN(name) name##_u16
#include "fast_cand.c"
#undef PIX
#undef FN

/* Mirrors r1o_rdo_cand_batch for square blocks 8..64 with tx_type DCT_DCT; `threads` OpenMP
 * threads over candidates (1 = the single-thread figure). */
#ifndef R1_FAST_ENTRY   /* fast_cand512.c compiles this file once more under its own ISA and name */
#define R1_FAST_ENTRY r1o_fast_rdo_cand_batch
#endif
int R1_FAST_ENTRY(const r1o_plane *org, const r1o_plane *ref, int n_px, int tx_size,
                            const r1o_rdo_cand *c, int n, int threads, uint32_t *sad_out,
                            uint32_t *satd_out, void *coeffs) {
  if (n_px != 8 && n_px != 16 && n_px != 32 && n_px != 64) return -1;
  if (r1o_tx_width(tx_size) != n_px || r1o_tx_height(tx_size) != n_px) return -1;
  if (org->bytes_per_px != ref->bytes_per_px) return -1;
  for (int i = 0; i < n; i++)
    if (c[i].tx_type != 0) return -1;
  if (threads < 1) threads = 1;
  if (org->bytes_per_px == 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < n; i++) cand_u8(org, ref, n_px, &c[i], i, sad_out, satd_out, coeffs);
  } else {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < n; i++) cand_u16(org, ref, n_px, &c[i], i, sad_out, satd_out, coeffs);
  }
  return 0;
}

#else /* FAST_BODY: everything below is compiled once per pixel type */

/* 8 outputs of an 8-tap pass: taps[i] * p[i * step + 0..7], i32 lanes */
static inline v8si FN(tap8)(const PIX *p, ptrdiff_t step, const int16_t *f) {
  v8si a = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 8; i++)
    a += __builtin_convertvector(*(const FN(pixv) *)(p + i * step), v8si) * (int32_t)f[i];
  return a;
}
static inline v8si FN(clampv)(v8si a, int32_t maxv) {
  const v8si hi = (v8si){0, 0, 0, 0, 0, 0, 0, 0} + maxv;
  a &= ~(a >> 31);              /* max(a, 0) */
  const v8si over = a > hi;     /* -1 where a > hi */
  return (a & ~over) | (hi & over);
}

/* put_8tap into a dense n x n block (n a multiple of 8), the four cases of mc.rs:268-351 */
static void FN(put8)(PIX *dst, const PIX *src, ptrdiff_t ss, int n, int col_frac, int row_frac,
                     int mode_x, int mode_y, int bd) {
  const int16_t *yf = r1o_get_filter(mode_y, row_frac, n);
  const int16_t *xf = r1o_get_filter(mode_x, col_frac, n);
  const int32_t maxv = (1 << bd) - 1;
  const int ib = 4 - (bd == 12 ? 2 : 0);
  if (col_frac == 0 && row_frac == 0) {
    for (int r = 0; r < n; r++) memcpy(dst + r * n, src + r * ss, n * sizeof(PIX));
  } else if (col_frac == 0) {
    for (int r = 0; r < n; r++)
      for (int x = 0; x < n; x += 8) {
        const v8si a = (FN(tap8)(src + (r - 3) * ss + x, ss, yf) + 64) >> 7;
        *(FN(pixv) *)(dst + r * n + x) = __builtin_convertvector(FN(clampv)(a, maxv), FN(pixv));
      }
  } else if (row_frac == 0) {
    const int s1 = 7 - ib;
    for (int r = 0; r < n; r++)
      for (int x = 0; x < n; x += 8) {
        v8si a = (FN(tap8)(src + r * ss + x - 3, 1, xf) + ((1 << s1) >> 1)) >> s1;
        a = (a + ((1 << ib) >> 1)) >> ib;
        *(FN(pixv) *)(dst + r * n + x) = __builtin_convertvector(FN(clampv)(a, maxv), FN(pixv));
      }
  } else {
    int16_t mid[(64 + 7) * 64] __attribute__((aligned(32)));
    const int s1 = 7 - ib, s2 = 7 + ib;
    for (int r = 0; r < n + 7; r++)
      for (int x = 0; x < n; x += 8) {
        const v8si a = (FN(tap8)(src + (r - 3) * ss + x - 3, 1, xf) + ((1 << s1) >> 1)) >> s1;
        *(v8hi_u *)(mid + r * n + x) = __builtin_convertvector(a, v8hi);
      }
    for (int r = 0; r < n; r++)
      for (int x = 0; x < n; x += 8) {
        v8si a = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8; i++)
          a += __builtin_convertvector(*(const v8hi_u *)(mid + (r + i) * n + x), v8si) * (int32_t)yf[i];
        a = (a + ((1 << s2) >> 1)) >> s2;
        *(FN(pixv) *)(dst + r * n + x) = __builtin_convertvector(FN(clampv)(a, maxv), FN(pixv));
      }
  }
}

static void FN(cand)(const r1o_plane *org, const r1o_plane *ref, int n, const r1o_rdo_cand *c, int i,
                     uint32_t *sad_out, uint32_t *satd_out, void *coeffs) {
  PIX pred[64 * 64] __attribute__((aligned(32)));
  int16_t resid[64 * 64] __attribute__((aligned(32)));
  const PIX *o = (const PIX *)org->data + (size_t)(org->yorigin + c->oy) * org->stride + org->xorigin + c->ox;
  const PIX *r = (const PIX *)ref->data + (size_t)(ref->yorigin + c->ry) * ref->stride + ref->xorigin + c->rx;
  FN(put8)(pred, r, ref->stride, n, c->col_frac, c->row_frac, c->mode_x, c->mode_y, ref->bit_depth);
  v8su sadv = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int y = 0; y < n; y++) {
    const PIX *or_ = o + (size_t)y * org->stride;
    for (int x = 0; x < n; x += 8) {
      const v8si d = __builtin_convertvector(*(const FN(pixv) *)(or_ + x), v8si) -
                     __builtin_convertvector(*(const FN(pixv) *)(pred + y * n + x), v8si);
      *(v8hi_u *)(resid + y * n + x) = __builtin_convertvector(d, v8hi);
      const v8si m = d >> 31;
      sadv += (v8su)((d ^ m) - m);
    }
  }
  uint32_t sad = 0;
  for (int k = 0; k < 8; k++) sad += sadv[k];
  if (sad_out) sad_out[i] = sad;
  if (satd_out) satd_out[i] = satd_resid(resid, n);
  if (coeffs) {
    const int hbd = sizeof(PIX) == 2;
    fwd_dct2d(resid, (uint8_t *)coeffs + (size_t)i * n * n * (hbd ? 4 : 2), n, org->bit_depth, hbd);
  }
}
#endif
