/* tests/c/test_dispatch.c -- every per-table-entry dispatch symbol of librav1e_hip.so
 * (include/rav1e_amd_dispatch.h) is looked up with dlsym, called through a function-pointer
 * typedef that restates the reference's fn type, and checked against the CPU oracle
 * (oracle/libr1oracle.so: test infrastructure).  Needs a GPU to run; built and run by
 * tests/test_dispatch_c.py.
 *
 * fn types: src/asm/x86/dist/mod.rs:21-43, dist/sse.rs:18-34, dist/cdef_dist.rs:18-24,
 * mc.rs:17-78, cdef.rs:16-37,184-191, quantize.rs:22-31, src/asm/shared/transform/inverse.rs:15-19,
 * predict.rs:21-236 (angular / z2 / cfl_ac / cfl prediction entry points)
 *
 *   test_dispatch <librav1e_hip.so> [list]     list: only dlsym every symbol (no GPU needed)
 */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/r1_oracle.h"

typedef uint32_t (*SadFn)(const uint8_t *, ptrdiff_t, const uint8_t *, ptrdiff_t);
typedef uint32_t (*SadHBDFn)(const uint16_t *, ptrdiff_t, const uint16_t *, ptrdiff_t);
typedef uint32_t (*SatdHBDFn)(const uint16_t *, ptrdiff_t, const uint16_t *, ptrdiff_t, uint32_t);
typedef uint64_t (*WeightedSseFn)(const uint8_t *, ptrdiff_t, const uint8_t *, ptrdiff_t, const uint32_t *, ptrdiff_t);
typedef uint64_t (*WeightedSseHBDFn)(const uint16_t *, ptrdiff_t, const uint16_t *, ptrdiff_t, const uint32_t *, ptrdiff_t);
typedef void (*CdefDistKernelFn)(const uint8_t *, ptrdiff_t, const uint8_t *, ptrdiff_t, uint32_t *);
typedef void (*CdefDistKernelHBDFn)(const uint16_t *, ptrdiff_t, const uint16_t *, ptrdiff_t, uint32_t *);
typedef void (*PutFn)(uint8_t *, ptrdiff_t, const uint8_t *, ptrdiff_t, int32_t, int32_t, int32_t, int32_t);
typedef void (*PutHBDFn)(uint16_t *, ptrdiff_t, const uint16_t *, ptrdiff_t, int32_t, int32_t, int32_t, int32_t, int32_t);
typedef void (*PrepFn)(int16_t *, const uint8_t *, ptrdiff_t, int32_t, int32_t, int32_t, int32_t);
typedef void (*PrepHBDFn)(int16_t *, const uint16_t *, ptrdiff_t, int32_t, int32_t, int32_t, int32_t, int32_t);
typedef void (*AvgFn)(uint8_t *, ptrdiff_t, const int16_t *, const int16_t *, int32_t, int32_t);
typedef void (*AvgHBDFn)(uint16_t *, ptrdiff_t, const int16_t *, const int16_t *, int32_t, int32_t, int32_t);
typedef void (*InvTxfmFunc)(uint8_t *, ptrdiff_t, int16_t *, int32_t);
typedef void (*InvTxfmHBDFunc)(uint16_t *, ptrdiff_t, int16_t *, int32_t, int32_t);
typedef void (*CdefFilterFn)(uint8_t *, ptrdiff_t, const uint16_t *, ptrdiff_t, int32_t, int32_t, int32_t, int32_t);
typedef void (*CdefFilterHBDFn)(uint16_t *, ptrdiff_t, const uint16_t *, ptrdiff_t, int32_t, int32_t, int32_t, int32_t, int32_t);
typedef int32_t (*CdefDirLBDFn)(const uint8_t *, ptrdiff_t, uint32_t *);
typedef int32_t (*CdefDirHBDFn)(const uint16_t *, ptrdiff_t, uint32_t *, int32_t);
typedef void (*IpredFn)(uint8_t *, ptrdiff_t, const uint8_t *, int, int, int);
typedef void (*IpredHBDFn)(uint16_t *, ptrdiff_t, const uint16_t *, int, int, int, int, int, int);
typedef void (*IpredZ2Fn)(uint8_t *, ptrdiff_t, const uint8_t *, int, int, int, int, int);
typedef void (*IpredZ2HBDFn)(uint16_t *, ptrdiff_t, const uint16_t *, int, int, int, int, int, int);
typedef void (*CflAcFn)(int16_t *, const uint8_t *, ptrdiff_t, int, int, int, int);
typedef void (*CflAcHBDFn)(int16_t *, const uint16_t *, ptrdiff_t, int, int, int, int);
typedef void (*CflPredFn)(uint8_t *, ptrdiff_t, const uint8_t *, int, int, const int16_t *, int);
typedef void (*CflPredHBDFn)(uint16_t *, ptrdiff_t, const uint16_t *, int, int, const int16_t *, int, int);
typedef void (*DequantizeFn)(uint8_t, const int16_t *, uint16_t, int16_t *, uint8_t, size_t, int8_t, int8_t);

static void *lib;
static int list_only, n_syms, n_checks, n_fail;
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }

static void *sym(const char *name) {
  void *p = dlsym(lib, name);
  n_syms++;
  if (!p) { fprintf(stderr, "MISSING %s\n", name); n_fail++; }
  return p;
}
#define CHECK(cond, name)                                             \
  do {                                                                \
    n_checks++;                                                       \
    if (!(cond)) { n_fail++; fprintf(stderr, "FAIL %s (%s)\n", name, #cond); } \
  } while (0)

/* two planes with different strides, blocks at an odd offset */
#define PW 160
#define PH 160
static uint8_t a8[PH * PW], b8[PH * (PW + 24)];
static uint16_t a16[PH * PW], b16[PH * (PW + 24)];
static void fill(int bd) {
  for (int i = 0; i < PH * PW; i++) { a8[i] = (uint8_t)rnd(); a16[i] = (uint16_t)(rnd() & ((1 << bd) - 1)); }
  for (int i = 0; i < PH * (PW + 24); i++) { b8[i] = (uint8_t)rnd(); b16[i] = (uint16_t)(rnd() & ((1 << bd) - 1)); }
}
#define A8 (a8 + 5 * PW + 7)
#define B8 (b8 + 9 * (PW + 24) + 3)
#define A16 (a16 + 5 * PW + 7)
#define B16 (b16 + 9 * (PW + 24) + 3)
#define SA PW
#define SB (PW + 24)

static void t_sad(const char *nm, int w, int h, int satd, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  for (int bd = hbd ? 10 : 8; bd <= (hbd ? 12 : 8); bd += 2) {
    fill(bd);
    uint32_t got, want;
    if (!hbd) {
      got = ((SadFn)f)(A8, SA, B8, SB);
      want = satd ? r1o_get_satd(A8, SA, B8, SB, w, h, 0) : r1o_get_sad(A8, SA, B8, SB, w, h, 0);
    } else {
      got = satd ? ((SatdHBDFn)f)(A16, SA * 2, B16, SB * 2, (1u << bd) - 1) : ((SadHBDFn)f)(A16, SA * 2, B16, SB * 2);
      want = satd ? r1o_get_satd(A16, SA, B16, SB, w, h, 1) : r1o_get_sad(A16, SA, B16, SB, w, h, 1);
    }
    CHECK(got == want, nm);
  }
}

static void t_wsse(const char *nm, int w, int h, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  fill(hbd ? 10 : 8);
  uint32_t scale[32 * 40];
  const int sstride = 40;   /* entries; the asm takes BYTES */
  for (int i = 0; i < 32 * 40; i++) scale[i] = (1u << 13) + rnd() % (1u << 14);
  uint64_t raw, want;
  if (!hbd) {
    raw = ((WeightedSseFn)f)(A8, SA, B8, SB, scale, sstride * 4);
    want = r1o_get_weighted_sse(A8, SA, B8, SB, scale, sstride, w, h, 0);
  } else {
    raw = ((WeightedSseHBDFn)f)(A16, SA * 2, B16, SB * 2, scale, sstride * 4);
    want = r1o_get_weighted_sse(A16, SA, B16, SB, scale, sstride, w, h, 1);
  }
  CHECK((raw + 32) / 64 == want, nm);   /* the wrapper's (ret + den / 2) / den, sse.rs:123-131 */
  /* the raw sum itself, from the definition */
  uint64_t def = 0;
  for (int cy = 0; cy < h / 4; cy++)
    for (int cx = 0; cx < w / 4; cx++) {
      uint64_t s = 0;
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
          const int64_t d = hbd ? (int64_t)A16[(cy * 4 + y) * SA + cx * 4 + x] - B16[(cy * 4 + y) * SB + cx * 4 + x]
                                : (int64_t)A8[(cy * 4 + y) * SA + cx * 4 + x] - B8[(cy * 4 + y) * SB + cx * 4 + x];
          s += (uint64_t)(d * d);
        }
      def += (s * scale[cy * sstride + cx] + 128) >> 8;
    }
  CHECK(raw == def, nm);
}

static void t_cdk(const char *nm, int w, int h, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  const int bd = hbd ? 10 : 8;
  fill(bd);
  uint32_t ret[3] = {0, 0, 0};
  uint32_t want;
  if (!hbd) {
    ((CdefDistKernelFn)f)(A8, SA, B8, SB, ret);
    want = r1o_cdef_dist_kernel(A8, SA, B8, SB, w, h, bd, 0);
  } else {
    ((CdefDistKernelHBDFn)f)(A16, SA * 2, B16, SB * 2, ret);
    want = r1o_cdef_dist_kernel(A16, SA, B16, SB, w, h, bd, 1);
  }
  CHECK(r1o_apply_ssim_boost(ret[2], ret[0], ret[1], bd) == want, nm);   /* [svar, dvar, sse] */
}

static void t_mc(const char *nm, int mode_x, int mode_y, int kind, int hbd) {   /* kind 0 put 1 prep */
  void *f = sym(nm);
  if (!f || list_only) return;
  static const int sizes[][2] = {{4, 4}, {8, 8}, {16, 8}, {32, 32}, {64, 64}, {128, 16}, {8, 16}, {2, 2}};
  for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); s++) {
    const int w = sizes[s][0], h = sizes[s][1];
    if (w == 2 && kind == 1) continue;
    for (int bd = hbd ? 10 : 8; bd <= (hbd ? 12 : 8); bd += 2) {
      fill(bd);
      const int mx = rnd() % 16 * (rnd() % 4 != 0), my = rnd() % 16 * (rnd() % 4 != 0);
      if (kind == 0) {
        if (!hbd) {
          static uint8_t g[128 * 136], o[128 * 136];
          ((PutFn)f)(g, 136, A8, SA, w, h, mx, my);
          r1o_put_8tap(o, 136, A8, SA, w, h, mx, my, mode_x, mode_y, 8, 0);
          int ok = 1;
          for (int y = 0; y < h; y++) ok &= !memcmp(g + y * 136, o + y * 136, w);
          CHECK(ok, nm);
        } else {
          static uint16_t g[128 * 136], o[128 * 136];
          ((PutHBDFn)f)(g, 136 * 2, A16, SA * 2, w, h, mx, my, (1 << bd) - 1);
          r1o_put_8tap(o, 136, A16, SA, w, h, mx, my, mode_x, mode_y, bd, 1);
          int ok = 1;
          for (int y = 0; y < h; y++) ok &= !memcmp(g + y * 136, o + y * 136, w * 2);
          CHECK(ok, nm);
        }
      } else {
        static int16_t g[128 * 128], o[128 * 128];
        if (!hbd) {
          ((PrepFn)f)(g, A8, SA, w, h, mx, my);
          r1o_prep_8tap(o, A8, SA, w, h, mx, my, mode_x, mode_y, 8, 0);
        } else {
          ((PrepHBDFn)f)(g, A16, SA * 2, w, h, mx, my, (1 << bd) - 1);
          r1o_prep_8tap(o, A16, SA, w, h, mx, my, mode_x, mode_y, bd, 1);
        }
        CHECK(!memcmp(g, o, (size_t)w * h * 2), nm);
      }
    }
  }
}

static void t_avg(const char *nm, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  static int16_t t1[64 * 64], t2[64 * 64];
  for (int bd = hbd ? 10 : 8; bd <= (hbd ? 12 : 8); bd += 2) {
    for (int i = 0; i < 64 * 64; i++) {
      t1[i] = (int16_t)((int)(rnd() % 16000) - (hbd ? 8000 : 0));
      t2[i] = (int16_t)((int)(rnd() % 16000) - (hbd ? 8000 : 0));
    }
    const int w = 32, h = 16;
    if (!hbd) {
      static uint8_t g[64 * 80], o[64 * 80];
      ((AvgFn)f)(g, 80, t1, t2, w, h);
      r1o_mc_avg(o, 80, t1, t2, w, h, 8, 0);
      int ok = 1;
      for (int y = 0; y < h; y++) ok &= !memcmp(g + y * 80, o + y * 80, w);
      CHECK(ok, nm);
    } else {
      static uint16_t g[64 * 80], o[64 * 80];
      ((AvgHBDFn)f)(g, 80 * 2, t1, t2, w, h, (1 << bd) - 1);
      r1o_mc_avg(o, 80, t1, t2, w, h, bd, 1);
      int ok = 1;
      for (int y = 0; y < h; y++) ok &= !memcmp(g + y * 80, o + y * 80, w * 2);
      CHECK(ok, nm);
    }
  }
}

static void t_itx(const char *nm, int ts, int tt, int bpc) {
  void *f = sym(nm);
  if (!f || list_only) return;
  const int w = r1o_tx_width(ts), h = r1o_tx_height(ts);
  const int area = (w < 32 ? w : 32) * (h < 32 ? h : 32);
  const int bd = bpc, hbd = bd > 8;
  fill(bd);
  static int16_t res[64 * 64];
  static int32_t co32[64 * 64], c32[32 * 32];
  static int16_t co16[64 * 64], c16[32 * 32];
  for (int i = 0; i < w * h; i++) res[i] = (int16_t)((int)(rnd() % (2u << bd)) - (1 << bd) + 1);
  /* coefficients of a real residual (forward transform of the oracle), sparse tail */
  if (hbd) r1o_forward_transform(res, co32, w, ts, tt, bd, 1);
  else r1o_forward_transform(res, co16, w, ts, tt, bd, 0);
  for (int i = 0; i < area; i++) {
    const int keep = i < 24 || rnd() % 3 == 0;
    c32[i] = keep ? co32[i] : 0;
    c16[i] = keep ? co16[i] : 0;
  }
  if (!hbd) {
    static uint8_t g[64 * 72], o[64 * 72];
    for (int i = 0; i < 64 * 72; i++) g[i] = o[i] = (uint8_t)rnd();
    int16_t scratch[32 * 32];
    memcpy(scratch, c16, sizeof(scratch));
    ((InvTxfmFunc)f)(g, 72, scratch, area - 1);
    r1o_inverse_transform_add(c16, o, 72, ts, tt, 8, 0, 0);
    CHECK(!memcmp(g, o, sizeof(g)), nm);
  } else {
    static uint16_t g[64 * 72], o[64 * 72];
    for (int i = 0; i < 64 * 72; i++) g[i] = o[i] = (uint16_t)(rnd() & ((1 << bd) - 1));
    int32_t scratch[32 * 32];
    memcpy(scratch, c32, sizeof(scratch));
    ((InvTxfmHBDFunc)f)(g, 72 * 2, (int16_t *)scratch, area - 1, (1 << bd) - 1);
    r1o_inverse_transform_add(c32, o, 72, ts, tt, bd, 1, 1);
    CHECK(!memcmp(g, o, sizeof(g)), nm);
  }
}

static void t_cdef_filter(const char *nm, int xdec, int ydec, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  const int bd = hbd ? 10 : 8, xs = 8 >> xdec, ys = 8 >> ydec;
  for (int rep = 0; rep < 6; rep++) {
    uint16_t tmp[12 * 16];
    for (int i = 0; i < 12 * 16; i++) tmp[i] = (uint16_t)(rnd() & ((1 << bd) - 1));
    if (rep & 1) for (int x = 0; x < 16; x++) tmp[x] = tmp[16 + x] = 0x8000;      /* no top rows */
    if (rep & 2) for (int y = 0; y < 12; y++) tmp[y * 16] = tmp[y * 16 + 1] = 0x8000;   /* no left columns */
    const int pri = (rnd() % 16) << (bd - 8), sec = (rnd() % 5 == 3 ? 4 : rnd() % 3) << (bd - 8);
    const int dir = rnd() % 8, damping = 3 + rnd() % 4 + (bd - 8);
    uint16_t want[8 * 8];
    r1o_cdef_filter_block(want, 8, tmp + 2 * 16 + 2, 16, pri, sec, dir, damping, bd, xdec, ydec, 15, 1);
    int ok = 1;
    if (!hbd) {
      uint8_t g[8 * 24];
      ((CdefFilterFn)f)(g, 24, tmp + 2 * 16 + 2, 16 * 2, pri, sec, dir, damping);
      for (int y = 0; y < ys; y++) for (int x = 0; x < xs; x++) ok &= g[y * 24 + x] == want[y * 8 + x];
    } else {
      uint16_t g[8 * 24];
      ((CdefFilterHBDFn)f)(g, 24 * 2, tmp + 2 * 16 + 2, 16 * 2, pri, sec, dir, damping, (1 << bd) - 1);
      for (int y = 0; y < ys; y++) for (int x = 0; x < xs; x++) ok &= g[y * 24 + x] == want[y * 8 + x];
    }
    CHECK(ok, nm);
  }
}

static void t_cdef_dir(const char *nm, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  for (int bd = hbd ? 10 : 8; bd <= (hbd ? 12 : 8); bd += 2) {
    fill(bd);
    uint32_t v1 = 1, v2 = 2;
    int d1, d2;
    if (!hbd) { d1 = ((CdefDirLBDFn)f)(A8, SA, &v1); d2 = r1o_cdef_find_dir(A8, SA, &v2, 0, 0); }
    else { d1 = ((CdefDirHBDFn)f)(A16, SA * 2, &v1, (1 << bd) - 1); d2 = r1o_cdef_find_dir(A16, SA, &v2, bd - 8, 1); }
    CHECK(d1 == d2 && v1 == v2, nm);
  }
}

static void t_deq(const char *nm) {
  void *f = sym(nm);
  if (!f || list_only) return;
  for (int ts = 0; ts < 19; ts++) {
    const int w = r1o_tx_width(ts), h = r1o_tx_height(ts);
    const int area = (w < 32 ? w : 32) * (h < 32 ? h : 32);
    int16_t q[1024], g[1024], o[1024];
    for (int i = 0; i < area; i++) q[i] = (int16_t)((int)(rnd() % 61) - 30);
    const int qindex = 30 + rnd() % 200;
    ((DequantizeFn)f)((uint8_t)qindex, q, (uint16_t)area, g, (uint8_t)ts, 8, -2, 3);
    r1o_dequantize(q, o, ts, qindex, 8, -2, 3, 0);
    CHECK(!memcmp(g, o, (size_t)area * 2), nm);
  }
}

/* ---- predict:: entries.  The edge buffer is the reference's IntraEdgeBuffer (257 pixels, top-left
 * at index 128, left below it, above after it); the oracle gets the same buffer with the edge
 * lengths the asm contract implies (left h / above w, the zone's far edge w + h) and the whole
 * block inside the frame; z2 additionally with its dx / dy clip. */
static void t_ipred(const char *nm, int mode, int variant, int hbd, int z2) {
  void *f = sym(nm);
  if (!f || list_only) return;
  static const int z1a[] = {36, 45, 54, 67, 76, 81, 87}, z2a[] = {93, 104, 113, 135, 157, 166, 177},
                   z3a[] = {183, 194, 203, 212};
  const int *angs = mode == 3 ? z1a : (mode == 4 ? z2a : z3a);
  const int dirn = mode == 3 || mode == 4 || mode == 7;
  const int nang = !dirn ? 1 : (mode == 7 ? 4 : 7);
  for (int bd = hbd ? 10 : 8; bd <= (hbd ? 12 : 8); bd += 2)
    for (int ts = 0; ts < 19; ts++) {
      const int w = r1o_tx_width(ts), h = r1o_tx_height(ts);
      for (int ai = 0; ai < nang; ai++)
        for (int ief = 0; ief < (dirn ? 3 : 1); ief++) {
          uint8_t e8[257];
          uint16_t e16[257];
          for (int i = 0; i < 257; i++) { e8[i] = (uint8_t)rnd(); e16[i] = (uint16_t)(rnd() & ((1 << bd) - 1)); }
          int angle = 0, left = h, above = w, aw = w, ah = h, dx = 0, dy = 0;
          if (dirn) angle = angs[ai];
          else if (mode == 1) angle = 90;
          else if (mode == 2) angle = 180;
          if (mode == 3) above = w + h;
          if (mode == 7) left = w + h;
          if (z2) {   /* frame edge somewhere inside, or far away */
            dx = (rnd() & 1) ? w + 8 : (w > 4 ? w / 2 : w);
            dy = (rnd() & 1) ? h + 8 : (h > 4 ? h / 2 : h);
            aw = dx < w ? dx : w;
            ah = dy < h ? dy : h;
          }
          const int angle_arg = dirn ? (angle | (ief ? 1 << 10 : 0) | (ief == 2 ? 1 << 9 : 0)) : angle;
          int ok = 1;
          if (!hbd) {
            static uint8_t g[64 * 80], o[64 * 80];
            memset(g, 0x5a, sizeof(g)); memset(o, 0x5a, sizeof(o));
            if (z2) ((IpredZ2Fn)f)(g, 80, e8 + 128, w, h, angle_arg, dx, dy);
            else ((IpredFn)f)(g, 80, e8 + 128, w, h, angle_arg);
            r1o_dispatch_predict_intra(mode, variant, o, 80, ts, 8, NULL, angle, ief, e8, left, above, aw, ah, 0);
            ok = !memcmp(g, o, sizeof(g));
          } else {
            static uint16_t g[64 * 80], o[64 * 80];
            memset(g, 0x5a, sizeof(g)); memset(o, 0x5a, sizeof(o));
            if (z2) ((IpredZ2HBDFn)f)(g, 80 * 2, e16 + 128, w, h, angle_arg, dx, dy, (1 << bd) - 1);
            else ((IpredHBDFn)f)(g, 80 * 2, e16 + 128, w, h, angle_arg, 0, 0, (1 << bd) - 1);
            r1o_dispatch_predict_intra(mode, variant, o, 80, ts, bd, NULL, angle, ief, e16, left, above, aw, ah, 1);
            ok = !memcmp(g, o, sizeof(g));
          }
          CHECK(ok, nm);
        }
    }
}

static void t_cfl(const char *nm, int variant, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  for (int bd = hbd ? 10 : 8; bd <= (hbd ? 12 : 8); bd += 2)
    for (int ts = 0; ts < 19; ts++) {
      const int w = r1o_tx_width(ts), h = r1o_tx_height(ts);
      if (w > 32 || h > 32) continue;   /* CFL blocks are at most 32x32 */
      uint8_t e8[257];
      uint16_t e16[257];
      for (int i = 0; i < 257; i++) { e8[i] = (uint8_t)rnd(); e16[i] = (uint16_t)(rnd() & ((1 << bd) - 1)); }
      int16_t ac[32 * 32];
      for (int i = 0; i < w * h; i++) ac[i] = (int16_t)((int)(rnd() % 1024) - 512);
      const int alpha = (int)(rnd() % 33) - 16;
      int ok;
      if (!hbd) {
        static uint8_t g[32 * 48], o[32 * 48];
        memset(g, 0x5a, sizeof(g)); memset(o, 0x5a, sizeof(o));
        ((CflPredFn)f)(g, 48, e8 + 128, w, h, ac, alpha);
        r1o_dispatch_predict_intra(13, variant, o, 48, ts, 8, ac, alpha, 0, e8, h, w, w, h, 0);
        ok = !memcmp(g, o, sizeof(g));
      } else {
        static uint16_t g[32 * 48], o[32 * 48];
        memset(g, 0x5a, sizeof(g)); memset(o, 0x5a, sizeof(o));
        ((CflPredHBDFn)f)(g, 48 * 2, e16 + 128, w, h, ac, alpha, (1 << bd) - 1);
        r1o_dispatch_predict_intra(13, variant, o, 48, ts, bd, ac, alpha, 0, e16, h, w, w, h, 1);
        ok = !memcmp(g, o, sizeof(g));
      }
      CHECK(ok, nm);
    }
}

static void t_cflac(const char *nm, int xdec, int ydec, int hbd) {
  void *f = sym(nm);
  if (!f || list_only) return;
  static const int dims[][2] = {{4, 4}, {4, 8}, {8, 4}, {8, 8}, {8, 16}, {16, 8}, {16, 16}, {16, 32}, {32, 16},
                                {32, 32}, {4, 16}, {16, 4}, {8, 32}, {32, 8}};
  fill(hbd ? 10 : 8);
  for (unsigned i = 0; i < sizeof(dims) / sizeof(dims[0]); i++)
    for (int pad = 0; pad < 3; pad++) {
      const int w = dims[i][0], h = dims[i][1];
      const int w_pad = pad == 1 ? (w / 4) / 2 : 0, h_pad = pad == 2 ? (h / 4) / 2 : 0;
      if ((w << xdec) + 7 > PW - 8 || (h << ydec) + 5 > PH - 8) continue;
      int16_t g[32 * 32], o[32 * 32];
      memset(g, 0x11, sizeof(g)); memset(o, 0x11, sizeof(o));
      if (!hbd) { ((CflAcFn)f)(g, A8, SA, w_pad, h_pad, w, h); r1o_pred_cfl_ac(o, A8, SA, w, h, w_pad, h_pad, xdec, ydec, 0); }
      else { ((CflAcHBDFn)f)(g, A16, SA * 2, w_pad, h_pad, w, h); r1o_pred_cfl_ac(o, A16, SA, w, h, w_pad, h_pad, xdec, ydec, 1); }
      CHECK(!memcmp(g, o, (size_t)w * h * 2), nm);
    }
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s librav1e_hip.so [list]\n", argv[0]); return 2; }
  lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  list_only = argc > 2 && !strcmp(argv[2], "list");
#define X_SAD(n, w, h) t_sad(#n, w, h, 0, 0);
#define X_SATD(n, w, h) t_sad(#n, w, h, 1, 0);
#define X_SAD_HBD(n, w, h) t_sad(#n, w, h, 0, 1);
#define X_SATD_HBD(n, w, h) t_sad(#n, w, h, 1, 1);
#define X_WSSE(n, w, h) t_wsse(#n, w, h, 0);
#define X_WSSE_HBD(n, w, h) t_wsse(#n, w, h, 1);
#define X_CDK(n, w, h) t_cdk(#n, w, h, 0);
#define X_CDK_HBD(n, w, h) t_cdk(#n, w, h, 1);
#define X_PUT(n, mx, my) t_mc(#n, mx, my, 0, 0);
#define X_PUT_HBD(n, mx, my) t_mc(#n, mx, my, 0, 1);
#define X_PREP(n, mx, my) t_mc(#n, mx, my, 1, 0);
#define X_PREP_HBD(n, mx, my) t_mc(#n, mx, my, 1, 1);
#define X_AVG(n, a, b) t_avg(#n, 0);
#define X_AVG_HBD(n, a, b) t_avg(#n, 1);
#define X_ITX(n, ts, tt) t_itx(#n, ts, tt, 8);
#define X_ITX_HBD(n, ts, tt, bpc) t_itx(#n, ts, tt, bpc);
#define X_CDEFF(n, xd, yd) t_cdef_filter(#n, xd, yd, 0);
#define X_CDEFF_HBD(n, xd, yd) t_cdef_filter(#n, xd, yd, 1);
#define X_CDEFD(n, a, b) t_cdef_dir(#n, 0);
#define X_CDEFD_HBD(n, a, b) t_cdef_dir(#n, 1);
#define X_DEQ(n, a, b) t_deq(#n);
#define X_IPRED(n, mode, variant) t_ipred(#n, mode, variant, 0, 0);
#define X_IPRED_HBD(n, mode, variant) t_ipred(#n, mode, variant, 1, 0);
#define X_IPRED_Z2(n, mode, variant) t_ipred(#n, mode, variant, 0, 1);
#define X_IPRED_Z2_HBD(n, mode, variant) t_ipred(#n, mode, variant, 1, 1);
#define X_CFL(n, mode, variant) t_cfl(#n, variant, 0);
#define X_CFL_HBD(n, mode, variant) t_cfl(#n, variant, 1);
#define X_CFLAC(n, xd, yd) t_cflac(#n, xd, yd, 0);
#define X_CFLAC_HBD(n, xd, yd) t_cflac(#n, xd, yd, 1);
#include "dispatch_list.h"
  printf("%d symbols, %d checks, %d failures\n", n_syms, n_checks, n_fail);
  return n_fail ? 1 : 0;
}
