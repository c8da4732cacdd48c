"""oracle/fwd_tx_np.py -- TEST INFRASTRUCTURE (not product code).

NumPy restatement of the reference's forward transform, written so that each
function can be read side by side with the Rust it follows:

  TxOperations for i32          src/transform/forward.rs:37-65
  forward_transform (2-D)       src/transform/forward.rs:71-161
  Txfm2DFlipCfg::fwd, shifts    src/transform/forward_shared.rs:22-165
  rotation / butterfly kernels  src/transform/forward_shared.rs:220-397
  daala_fdct*/fdst*/fwht4       src/transform/forward_shared.rs:398-1796
  VTX_TAB/HTX_TAB, valid_av1_transform  src/transform/mod.rs:364-417

Every scalar `T` of the reference is an int32 ndarray here (one lane per
independent 1-D transform), so wrapping i32 arithmetic is NumPy's native
behaviour and a whole pass of a 2-D transform is one call.  This is the
second, independent restatement next to oracle/fwd_tx.c (array/index style,
fast); tests/test_oracle_fwd_tx.py requires the two to agree bit for bit.

Parity status: the reference stores no forward-transform coefficients of its own
(SURVEY.md 8c); this file is pinned by tests/golden/fwd_tx_*.npz (vectors obtained in the build container by
executing the reference's own source text, see tests/golden/README.md), by
closeness to the real-valued DCT/ADST and by fwd->inv round trips.
"""
import numpy as np

I32 = np.int32


# ---- symbolic lanes: lets tools/gen_tx1d.py trace a 1-D network into
# straight-line SSA C (one statement per primitive op) ----
class Sym:
    """A traced value; `emit` appends `T tN = <expr>;` to the shared tape."""
    __slots__ = ("name", "tape")

    def __init__(self, name, tape):
        self.name, self.tape = name, tape

    def emit(self, expr):
        n = "t%d" % len(self.tape)
        self.tape.append((n, expr))
        return Sym(n, self.tape)


# ---- TxOperations for i32 (forward.rs:37-65) ----
def tx_mul(a, mul, shift):
    if isinstance(a, Sym):
        return a.emit("TX_MUL(%s, %d, %d)" % (a.name, mul, shift))
    return ((a * I32(mul)) + I32((1 << shift) >> 1)) >> I32(shift)


def rshift1(a):
    if isinstance(a, Sym):
        return a.emit("TX_RSHIFT1(%s)" % a.name)
    return (a + (a < 0).astype(I32)) >> I32(1)


def add(a, b):
    if isinstance(a, Sym):
        return a.emit("TX_ADD(%s, %s)" % (a.name, b.name))
    return a + b


def sub(a, b):
    if isinstance(a, Sym):
        return a.emit("TX_SUB(%s, %s)" % (a.name, b.name))
    return a - b


def add_avg(a, b):
    if isinstance(a, Sym):
        return a.emit("TX_ADD_AVG(%s, %s)" % (a.name, b.name))
    return (a + b) >> I32(1)


def sub_avg(a, b):
    if isinstance(a, Sym):
        return a.emit("TX_SUB_AVG(%s, %s)" % (a.name, b.name))
    return (a - b) >> I32(1)


def copy_fn(a):
    return a


# ---- rotation kernels (forward_shared.rs:220-345) ----
def _pi4(ADD, SUB):
    def kernel(s0, s1, p0, p1, m):
        t = ADD(p1, p0)
        a, out0 = tx_mul(p0, m[0], s0), tx_mul(t, m[1], s1)
        out1 = SUB(a, out0)
        return out0, out1
    return kernel


RotatePi4Add = _pi4(add, sub)
RotatePi4AddAvg = _pi4(add_avg, sub)
RotatePi4Sub = _pi4(sub, add)
RotatePi4SubAvg = _pi4(sub_avg, add)


class _Rot:
    def __init__(self, ADD, SUB, SHIFT):
        self.ADD, self.SUB, self.SHIFT = ADD, SUB, SHIFT

    def half_kernel(self, s0, s1, s2, p0, p1, m):
        t = self.ADD(p1, p0[0])
        a, b, c = tx_mul(p0[1], m[0], s0), tx_mul(p1, m[1], s1), tx_mul(t, m[2], s2)
        out0 = add(b, c)
        shifted = self.SHIFT(c)
        out1 = self.SUB(a, shifted)
        return out0, out1

    def kernel(self, s0, s1, s2, p0, p1, m):
        return self.half_kernel(s0, s1, s2, (p0, p0), p1, m)


RotateAdd = _Rot(add, sub, copy_fn)
RotateAddAvg = _Rot(add_avg, sub, copy_fn)
RotateAddShift = _Rot(add, sub, rshift1)
RotateSub = _Rot(sub, add, copy_fn)
RotateSubAvg = _Rot(sub_avg, add, copy_fn)
RotateSubShift = _Rot(sub, add, rshift1)


class _RotNeg:
    def __init__(self, ADD):
        self.ADD = ADD

    def kernel(self, s0, s1, s2, p0, p1, m):
        t = self.ADD(p0, p1)
        a, b, c = tx_mul(p0, m[0], s0), tx_mul(p1, m[1], s1), tx_mul(t, m[2], s2)
        out0 = sub(b, c)
        out1 = sub(c, a)
        return out0, out1


RotateNeg = _RotNeg(sub)
RotateNegAvg = _RotNeg(sub_avg)


# ---- butterflies (forward_shared.rs:347-396) ----
def butterfly_add(p0, p1):
    p0 = add(p0, p1)
    p0h = rshift1(p0)
    p1h = sub(p1, p0h)
    return (p0h, p0), p1h


def butterfly_sub(p0, p1):
    p0 = sub(p0, p1)
    p0h = rshift1(p0)
    p1h = add(p1, p0h)
    return (p0h, p0), p1h


def butterfly_neg(p0, p1):
    p1 = sub(p0, p1)
    p1h = rshift1(p1)
    p0h = sub(p0, p1h)
    return p0h, (p1h, p1)


def butterfly_add_asym(p0, p1h):
    p1 = add(p1h, p0[0])
    p0 = sub(p0[1], p1)
    return p0, p1


def butterfly_sub_asym(p0, p1h):
    p1 = sub(p1h, p0[0])
    p0 = add(p0[1], p1)
    return p0, p1


def butterfly_neg_asym(p0h, p1):
    p0 = add(p0h, p1[0])
    p1 = sub(p0, p1[1])
    return p0, p1


# ---- 2-point kernels ----
def daala_fdct_ii_2_asym(p0h, p1):
    return butterfly_neg_asym(p0h, p1)


def daala_fdst_iv_2_asym(p0, p1h):
    return RotateAdd.half_kernel(9, 12, 13, p0, p1h, (473, 3135, 4433))


def daala_fdct_ii_2(p0, p1):
    p1, p0 = RotatePi4SubAvg(13, 13, p1, p0, (11585, 11585))
    return p0, p1


def daala_fdst_iv_2(p0, p1):
    return RotateAddAvg.kernel(13, 14, 12, p0, p1, (10703, 8867, 3135))


# ---- 4-point ----
def daala_fdct_ii_4(q0, q1, q2, q3):
    q0h, q3 = butterfly_neg(q0, q3)
    q1, q2h = butterfly_add(q1, q2)
    q0, q1 = daala_fdct_ii_2_asym(q0h, q1)
    q3, q2 = daala_fdst_iv_2_asym(q3, q2h)
    return [q0, q1, q2, q3]


def daala_fdct4(c):
    t = daala_fdct_ii_4(c[0], c[1], c[2], c[3])
    return [t[0], t[2], t[1], t[3]]


def daala_fdst_vii_4(c):
    q0, q1, q2, q3 = c
    t0 = add(q1, q3)
    t1 = add(q1, sub_avg(q0, t0))
    t2 = sub(q0, q1)
    t3 = q2
    t4 = add(q0, q3)
    t0 = tx_mul(t0, 7021, 14)
    t1 = tx_mul(t1, 37837, 15)
    t2 = tx_mul(t2, 21513, 15)
    t3 = tx_mul(t3, 37837, 15)
    t4 = tx_mul(t4, 467, 11)
    t3h = rshift1(t3)
    u4 = add(t4, t3h)
    return [add(t0, u4), t1, add(t0, sub(t2, t3h)), add(t2, sub(t3, u4))]


def daala_fdct_ii_4_asym(q0h, q1, q2h, q3):
    q0, q3 = butterfly_neg_asym(q0h, q3)
    q1, q2 = butterfly_sub_asym(q1, q2h)
    q0, q1 = daala_fdct_ii_2(q0, q1)
    q3, q2 = daala_fdst_iv_2(q3, q2)
    return [q0, q1, q2, q3]


def daala_fdst_iv_4_asym(q0, q1h, q2, q3h):
    q0, q3 = RotateAddShift.half_kernel(14, 13, 15, q0, q3h, (9633, 12873, 12785))
    q2, q1 = RotateSubShift.half_kernel(14, 15, 12, q2, q1h, (11363, 18081, 4551))
    q2, q3 = butterfly_sub_asym((rshift1(q2), q2), q3)
    q0, q1 = butterfly_sub_asym((rshift1(q0), q0), q1)
    q2, q1 = RotatePi4AddAvg(13, 13, q2, q1, (11585, 11585))
    return [q0, q1, q2, q3]


def daala_fdst_iv_4(q0, q1, q2, q3):
    q0, q3 = RotateAddShift.kernel(14, 12, 11, q0, q3, (13623, 4551, 565))
    q2, q1 = RotateSubShift.kernel(14, 15, 11, q2, q1, (16069, 12785, 1609))
    q2, q3 = butterfly_sub_asym((rshift1(q2), q2), q3)
    q0, q1 = butterfly_sub_asym((rshift1(q0), q0), q1)
    q2, q1 = RotatePi4AddAvg(13, 13, q2, q1, (11585, 11585))
    return [q0, q1, q2, q3]


# ---- 8-point ----
def daala_fdct_ii_8(r0, r1, r2, r3, r4, r5, r6, r7):
    r0h, r7 = butterfly_neg(r0, r7)
    r1, r6h = butterfly_add(r1, r6)
    r2h, r5 = butterfly_neg(r2, r5)
    r3, r4h = butterfly_add(r3, r4)
    lo = daala_fdct_ii_4_asym(r0h, r1, r2h, r3)
    hi = daala_fdst_iv_4_asym(r7, r6h, r5, r4h)
    return lo + hi[::-1]


_PERM8 = [0, 4, 2, 6, 1, 5, 3, 7]


def daala_fdct8(c):
    t = daala_fdct_ii_8(*c)
    return [t[i] for i in _PERM8]


def daala_fdst_iv_8(r0, r1, r2, r3, r4, r5, r6, r7):
    r0, r7 = RotateAdd.kernel(14, 14, 13, r0, r7, (17911, 14699, 803))
    r6, r1 = RotateSub.kernel(14, 15, 12, r6, r1, (20435, 21845, 1189))
    r2, r5 = RotateAdd.kernel(14, 13, 15, r2, r5, (22173, 3363, 15447))
    r4, r3 = RotateSub.kernel(14, 14, 13, r4, r3, (23059, 2271, 5197))
    r0, r3h = butterfly_add(r0, r3)
    r2, r1h = butterfly_sub(r2, r1)
    r5, r6h = butterfly_add(r5, r6)
    r7, r4h = butterfly_sub(r7, r4)
    r7, r6 = butterfly_add_asym(r7, r6h)
    r5, r3 = butterfly_add_asym(r5, r3h)
    r2, r4 = butterfly_add_asym(r2, r4h)
    r0, r1 = butterfly_sub_asym(r0, r1h)
    r3, r4 = RotateSubAvg.kernel(13, 14, 12, r3, r4, (10703, 8867, 3135))
    r2, r5 = RotateNegAvg.kernel(13, 14, 12, r2, r5, (10703, 8867, 3135))
    r1, r6 = RotatePi4SubAvg(13, 13, r1, r6, (11585, 11585))
    return [r0, r1, r2, r3, r4, r5, r6, r7]


def daala_fdst8(c):
    t = daala_fdst_iv_8(*c)
    return [t[i] for i in _PERM8]


def daala_fdct_ii_8_asym(r0h, r1, r2h, r3, r4h, r5, r6h, r7):
    r0, r7 = butterfly_neg_asym(r0h, r7)
    r1, r6 = butterfly_sub_asym(r1, r6h)
    r2, r5 = butterfly_neg_asym(r2h, r5)
    r3, r4 = butterfly_sub_asym(r3, r4h)
    lo = daala_fdct_ii_4(r0, r1, r2, r3)
    hi = daala_fdst_iv_4(r7, r6, r5, r4)
    return lo + hi[::-1]


def daala_fdst_iv_8_asym(r0, r1h, r2, r3h, r4, r5h, r6, r7h):
    r0, r7 = RotateAdd.half_kernel(14, 12, 14, r0, r7h, (12665, 5197, 2271))
    r6, r1 = RotateSub.half_kernel(14, 15, 13, r6, r1h, (14449, 30893, 3363))
    r2, r5 = RotateAdd.half_kernel(14, 11, 13, r2, r5h, (15679, 1189, 5461))
    r4, r3 = RotateSub.half_kernel(14, 12, 14, r4, r3h, (16305, 803, 14699))
    r0, r3h = butterfly_add(r0, r3)
    r2, r1h = butterfly_sub(r2, r1)
    r5, r6h = butterfly_add(r5, r6)
    r7, r4h = butterfly_sub(r7, r4)
    r7, r6 = butterfly_add_asym(r7, r6h)
    r5, r3 = butterfly_add_asym(r5, r3h)
    r2, r4 = butterfly_add_asym(r2, r4h)
    r0, r1 = butterfly_sub_asym(r0, r1h)
    r3, r4 = RotateSubAvg.kernel(9, 14, 12, r3, r4, (669, 8867, 3135))
    r2, r5 = RotateNegAvg.kernel(9, 14, 12, r2, r5, (669, 8867, 3135))
    r1, r6 = RotatePi4SubAvg(12, 13, r1, r6, (5793, 11585))
    return [r0, r1, r2, r3, r4, r5, r6, r7]


# ---- 16-point ----
def daala_fdct_ii_16(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, sa, sb, sc, sd, se, sf):
    s0h, sf = butterfly_neg(s0, sf)
    s1, seh = butterfly_add(s1, se)
    s2h, sd = butterfly_neg(s2, sd)
    s3, sch = butterfly_add(s3, sc)
    s4h, sb = butterfly_neg(s4, sb)
    s5, sah = butterfly_add(s5, sa)
    s6h, s9 = butterfly_neg(s6, s9)
    s7, s8h = butterfly_add(s7, s8)
    lo = daala_fdct_ii_8_asym(s0h, s1, s2h, s3, s4h, s5, s6h, s7)
    hi = daala_fdst_iv_8_asym(sf, seh, sd, sch, sb, sah, s9, s8h)
    return lo + hi[::-1]


_PERM16 = [0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15]


def daala_fdct16(c):
    t = daala_fdct_ii_16(*c)
    return [t[i] for i in _PERM16]


def _fdst16_tail(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, sa, sb, sc, sd, se, sf, asym):
    """Stages 1..5 shared in shape by daala_fdst_iv_16 (asym=False,
    forward_shared.rs:872-947) and daala_fdst_iv_16_asym (asym=True,
    1071-1146); only the stage-3/5 rotations differ."""
    # Stage 1
    s0, s7 = butterfly_sub_asym((rshift1(s0), s0), s7)
    s8, sf = butterfly_sub_asym((rshift1(s8), s8), sf)
    s4, s3 = butterfly_add_asym((rshift1(s4), s4), s3)
    sc, sb = butterfly_add_asym((rshift1(sc), sc), sb)
    s2, s5 = butterfly_sub_asym((rshift1(s2), s2), s5)
    sa, sd = butterfly_sub_asym((rshift1(sa), sa), sd)
    s6, s1 = butterfly_add_asym((rshift1(s6), s6), s1)
    se, s9 = butterfly_add_asym((rshift1(se), se), s9)
    # Stage 2
    (_s8h, s8), s4h = butterfly_add(s8, s4)
    (_s7h, s7), sbh = butterfly_add(s7, sb)
    (_sah, sa), s6h = butterfly_sub(sa, s6)
    (_s5h, s5), s9h = butterfly_sub(s5, s9)
    s0, s3h = butterfly_add(s0, s3)
    sd, seh = butterfly_add(sd, se)
    s2, s1h = butterfly_sub(s2, s1)
    sf, sch = butterfly_sub(sf, sc)
    # Stage 3
    if not asym:
        s8, s7 = RotateAddAvg.kernel(8, 11, 15, s8, s7, (301, 1609, 12785))
        s9, s6 = RotateAdd.kernel(13, 15, 13, s9h, s6h, (11363, 9041, 4551))
        s5, sa = RotateNegAvg.kernel(12, 15, 12, s5, sa, (5681, 9041, 4551))
        s4, sb = RotateNeg.kernel(13, 14, 15, s4h, sbh, (9633, 12873, 6393))
    else:
        s8, s7 = RotateAdd.kernel(13, 14, 15, s8, s7, (9633, 12873, 6393))
        s9, s6 = RotateAdd.kernel(14, 15, 13, s9h, s6h, (22725, 9041, 4551))
        s5, sa = RotateNeg.kernel(13, 15, 13, s5, sa, (11363, 9041, 4551))
        s4, sb = RotateNeg.kernel(13, 14, 15, s4h, sbh, (9633, 12873, 6393))
    # Stage 4
    s2, sc = butterfly_add_asym(s2, sch)
    s0, s1 = butterfly_sub_asym(s0, s1h)
    sf, se = butterfly_add_asym(sf, seh)
    sd, s3 = butterfly_add_asym(sd, s3h)
    s7, s6 = butterfly_add_asym((rshift1(s7), s7), s6)
    s8, s9 = butterfly_sub_asym((rshift1(s8), s8), s9)
    sa, sb = butterfly_sub_asym((rshift1(sa), sa), sb)
    s5, s4 = butterfly_add_asym((rshift1(s5), s5), s4)
    # Stage 5
    if not asym:
        sc, s3 = RotateAddAvg.kernel(9, 14, 12, sc, s3, (669, 8867, 3135))
        s2, sd = RotateNegAvg.kernel(9, 14, 12, s2, sd, (669, 8867, 3135))
        sa, s5 = RotatePi4AddAvg(12, 13, sa, s5, (5793, 11585))
        s6, s9 = RotatePi4AddAvg(12, 13, s6, s9, (5793, 11585))
        se, s1 = RotatePi4AddAvg(12, 13, se, s1, (5793, 11585))
    else:
        sc, s3 = RotateAdd.kernel(13, 14, 13, sc, s3, (10703, 8867, 3135))
        s2, sd = RotateNeg.kernel(13, 14, 13, s2, sd, (10703, 8867, 3135))
        sa, s5 = RotatePi4Add(13, 13, sa, s5, (11585, 5793))
        s6, s9 = RotatePi4Add(13, 13, s6, s9, (11585, 5793))
        se, s1 = RotatePi4Add(13, 13, se, s1, (11585, 5793))
    return [s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, sa, sb, sc, sd, se, sf]


def daala_fdst_iv_16(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, sa, sb, sc, sd, se, sf):
    # Stage 0
    s0, sf = RotateAddShift.kernel(15, 13, 14, s0, sf, (24279, 11003, 1137))
    se, s1 = RotateSubShift.kernel(11, 8, 11, se, s1, (1645, 305, 425))
    s2, sd = RotateAddShift.kernel(14, 13, 13, s2, sd, (14053, 8423, 2815))
    sc, s3 = RotateSubShift.kernel(14, 13, 13, sc, s3, (14811, 7005, 3903))
    s4, sb = RotateAddShift.kernel(15, 14, 14, s4, sb, (30853, 11039, 9907))
    sa, s5 = RotateSubShift.kernel(14, 13, 11, sa, s5, (15893, 3981, 1489))
    s6, s9 = RotateAddShift.kernel(15, 11, 14, s6, s9, (32413, 601, 13803))
    s8, s7 = RotateSubShift.kernel(15, 11, 11, s8, s7, (32729, 201, 1945))
    return _fdst16_tail(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, sa, sb, sc, sd, se, sf, False)


def daala_fdst16(c):
    t = daala_fdst_iv_16(*c)
    return [t[i] for i in _PERM16]


def daala_fdct_ii_16_asym(s0h, s1, s2h, s3, s4h, s5, s6h, s7, s8h, s9, sah, sb, sch, sd, seh, sf):
    s0, sf = butterfly_neg_asym(s0h, sf)
    s1, se = butterfly_sub_asym(s1, seh)
    s2, sd = butterfly_neg_asym(s2h, sd)
    s3, sc = butterfly_sub_asym(s3, sch)
    s4, sb = butterfly_neg_asym(s4h, sb)
    s5, sa = butterfly_sub_asym(s5, sah)
    s6, s9 = butterfly_neg_asym(s6h, s9)
    s7, s8 = butterfly_sub_asym(s7, s8h)
    lo = daala_fdct_ii_8(s0, s1, s2, s3, s4, s5, s6, s7)
    hi = daala_fdst_iv_8(sf, se, sd, sc, sb, sa, s9, s8)
    return lo + hi[::-1]


def daala_fdst_iv_16_asym(s0, s1h, s2, s3h, s4, s5h, s6, s7h, s8, s9h, sa, sbh, sc, sdh, se, sfh):
    # Stage 0
    s0, sf = RotateAddShift.half_kernel(11, 15, 11, s0, sfh, (1073, 62241, 201))
    se, s1 = RotateSubShift.half_kernel(15, 15, 11, se, s1h, (18611, 55211, 601))
    s2, sd = RotateAddShift.half_kernel(14, 10, 13, s2, sdh, (9937, 1489, 3981))
    sc, s3 = RotateSubShift.half_kernel(14, 15, 14, sc, s3h, (10473, 39627, 11039))
    s4, sb = RotateAddShift.half_kernel(12, 12, 13, s4, sbh, (2727, 3903, 7005))
    sa, s5 = RotateSubShift.half_kernel(13, 12, 13, sa, s5h, (5619, 2815, 8423))
    s6, s9 = RotateAddShift.half_kernel(12, 15, 8, s6, s9h, (2865, 13599, 305))
    s8, s7 = RotateSubShift.half_kernel(15, 13, 13, s8, s7h, (23143, 1137, 11003))
    return _fdst16_tail(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, sa, sb, sc, sd, se, sf, True)


# ---- 32-point ----
def daala_fdct_ii_32(t):
    n = 32
    u, v = [None] * 16, [None] * 16
    for i in range(8):
        j = 2 * i
        u[j], v[j] = butterfly_neg(t[j], t[n - 1 - j])
        u[j + 1], v[j + 1] = butterfly_add(t[j + 1], t[n - 2 - j])
    lo = daala_fdct_ii_16_asym(*u)
    hi = daala_fdst_iv_16_asym(*v)
    return lo + hi[::-1]


_PERM32 = [0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30,
           1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31]


def daala_fdct32(c):
    t = daala_fdct_ii_32(c)
    return [t[i] for i in _PERM32]


def daala_fdct_ii_32_asym(a):
    # a: t0h, t1, t2h, t3, ... (even = half value, odd = (half, full) pair)
    n = 32
    x = [None] * n
    for j in range(0, 16, 2):
        x[j], x[n - 1 - j] = butterfly_neg_asym(a[j], a[n - 1 - j])
        x[j + 1], x[n - 2 - j] = butterfly_sub_asym(a[j + 1], a[n - 2 - j])
    lo = daala_fdct_ii_16(*x[:16])
    hi = daala_fdst_iv_16(*x[:15:-1])
    return lo + hi[::-1]


def daala_fdst_iv_32_asym(t0, t1h, t2, t3h, t4, t5h, t6, t7h, t8, t9h, ta, tbh, tc, tdh,
                          te, tfh, tg, thh, ti, tjh, tk, tlh, tm, tnh, to, tph, tq, trh,
                          ts, tth, tu, tvh):
    # Stage 0
    t0, tv = RotateAdd.half_kernel(13, 14, 15, t0, tvh, (5933, 22595, 1137))
    tu, t1 = RotateSub.half_kernel(13, 14, 15, tu, t1h, (6203, 21403, 3409))
    t2, tt = RotateAdd.half_kernel(15, 8, 15, t2, tth, (25833, 315, 5673))
    ts, t3 = RotateSub.half_kernel(15, 12, 15, ts, t3h, (26791, 4717, 7923))
    t4, tr = RotateAdd.half_kernel(13, 14, 15, t4, trh, (6921, 17531, 10153))
    tq, t5 = RotateSub.half_kernel(15, 15, 12, tq, t5h, (28511, 32303, 1545))
    t6, tp = RotateAdd.half_kernel(15, 14, 12, t6, tph, (29269, 14733, 1817))
    to, t7 = RotateSub.half_kernel(15, 14, 14, to, t7h, (29957, 13279, 8339))
    t8, tn = RotateAdd.half_kernel(13, 14, 15, t8, tnh, (7643, 11793, 18779))
    tm, t9 = RotateSub.half_kernel(14, 15, 15, tm, t9h, (15557, 20557, 20835))
    ta, tl = RotateAdd.half_kernel(15, 15, 15, ta, tlh, (31581, 17479, 22841))
    tk, tb = RotateSub.half_kernel(13, 15, 12, tk, tbh, (7993, 14359, 3099))
    tc, tj = RotateAdd.half_kernel(14, 13, 15, tc, tjh, (16143, 2801, 26683))
    ti, td = RotateSub.half_kernel(14, 14, 14, ti, tdh, (16261, 4011, 14255))
    te, th = RotateAdd.half_kernel(15, 15, 15, te, thh, (32679, 4821, 30269))
    tg, tf = RotateSub.half_kernel(14, 12, 14, tg, tfh, (16379, 201, 15977))
    # Stage 1
    t0, tfh = butterfly_add(t0, tf)
    tv, tgh = butterfly_sub(tv, tg)
    th, tuh = butterfly_add(th, tu)
    te, t1h = butterfly_sub(te, t1)
    t2, tdh = butterfly_add(t2, td)
    tt, tih = butterfly_sub(tt, ti)
    tj, tsh = butterfly_add(tj, ts)
    tc, t3h = butterfly_sub(tc, t3)
    t4, tbh = butterfly_add(t4, tb)
    tr, tkh = butterfly_sub(tr, tk)
    tl, tqh = butterfly_add(tl, tq)
    ta, t5h = butterfly_sub(ta, t5)
    t6, t9h = butterfly_add(t6, t9)
    tp, tmh = butterfly_sub(tp, tm)
    tn, toh = butterfly_add(tn, to)
    t8, t7h = butterfly_sub(t8, t7)
    # Stage 2
    t0, t7 = butterfly_sub_asym(t0, t7h)
    tv, to = butterfly_add_asym(tv, toh)
    tp, tu = butterfly_sub_asym(tp, tuh)
    t6, t1 = butterfly_add_asym(t6, t1h)
    t2, t5 = butterfly_sub_asym(t2, t5h)
    tt, tq = butterfly_add_asym(tt, tqh)
    tr, ts = butterfly_sub_asym(tr, tsh)
    t4, t3 = butterfly_add_asym(t4, t3h)
    t8, tg = butterfly_add_asym(t8, tgh)
    te, tm = butterfly_sub_asym(te, tmh)
    tn, tf = butterfly_add_asym(tn, tfh)
    th, t9 = butterfly_sub_asym(th, t9h)
    ta, ti = butterfly_add_asym(ta, tih)
    tc, tk = butterfly_sub_asym(tc, tkh)
    tl, td = butterfly_add_asym(tl, tdh)
    tj, tb = butterfly_sub_asym(tj, tbh)
    # Stage 3
    tf, tg = RotateSub.kernel(14, 14, 13, tf, tg, (17911, 14699, 803))
    th, te = RotateAdd.kernel(13, 13, 12, th, te, (10217, 5461, 1189))
    ti, td = RotateAdd.kernel(12, 13, 14, ti, td, (5543, 3363, 7723))
    tc, tj = RotateSub.kernel(13, 14, 13, tc, tj, (11529, 2271, 5197))
    tb, tk = RotateNeg.kernel(13, 14, 13, tb, tk, (11529, 2271, 5197))
    ta, tl = RotateNeg.kernel(12, 13, 14, ta, tl, (5543, 3363, 7723))
    t9, tm = RotateNeg.kernel(13, 13, 12, t9, tm, (10217, 5461, 1189))
    t8, tn = RotateNeg.kernel(14, 14, 13, t8, tn, (17911, 14699, 803))
    # Stage 4
    t3, t0h = butterfly_sub(t3, t0)
    ts, tvh = butterfly_add(ts, tv)
    tu, tth = butterfly_sub(tu, tt)
    t1, t2h = butterfly_add(t1, t2)
    (_toh, to), t4h = butterfly_add(to, t4)
    (_tqh, tq), t6h = butterfly_sub(tq, t6)
    (_t7h, t7), trh = butterfly_add(t7, tr)
    (_t5h, t5), tph = butterfly_sub(t5, tp)
    tb, t8h = butterfly_sub(tb, t8)
    tk, tnh = butterfly_add(tk, tn)
    tm, tlh = butterfly_sub(tm, tl)
    t9, tah = butterfly_add(t9, ta)
    tf, tch = butterfly_sub(tf, tc)
    tg, tjh = butterfly_add(tg, tj)
    ti, thh = butterfly_sub(ti, th)
    td, teh = butterfly_add(td, te)
    # Stage 5
    to, t7 = RotateAdd.kernel(8, 11, 15, to, t7, (301, 1609, 6393))
    tph, t6h = RotateAdd.kernel(13, 15, 13, tph, t6h, (11363, 9041, 4551))
    t5, tq = RotateNeg.kernel(12, 15, 13, t5, tq, (5681, 9041, 4551))
    t4h, trh = RotateNeg.kernel(13, 14, 15, t4h, trh, (9633, 12873, 6393))
    # Stage 6
    t1, t0 = butterfly_add_asym(t1, t0h)
    tu, tv = butterfly_sub_asym(tu, tvh)
    ts, t2 = butterfly_sub_asym(ts, t2h)
    t3, tt = butterfly_sub_asym(t3, tth)
    t5, t4 = butterfly_add_asym((rshift1(t5), t5), t4h)
    tq, tr = butterfly_sub_asym((rshift1(tq), tq), trh)
    t7, t6 = butterfly_add_asym((rshift1(t7), t7), t6h)
    to, tp = butterfly_sub_asym((rshift1(to), to), tph)
    t9, t8 = butterfly_add_asym(t9, t8h)
    tm, tn = butterfly_sub_asym(tm, tnh)
    tk, ta = butterfly_sub_asym(tk, tah)
    tb, tl = butterfly_sub_asym(tb, tlh)
    ti, tc = butterfly_add_asym(ti, tch)
    td, tj = butterfly_add_asym(td, tjh)
    tf, te = butterfly_add_asym(tf, teh)
    tg, th = butterfly_sub_asym(tg, thh)
    # Stage 7
    t2, tt = RotateNeg.kernel(9, 14, 13, t2, tt, (669, 8867, 3135))
    ts, t3 = RotateAdd.kernel(9, 14, 13, ts, t3, (669, 8867, 3135))
    ta, tl = RotateNeg.kernel(9, 14, 13, ta, tl, (669, 8867, 3135))
    tk, tb = RotateAdd.kernel(9, 14, 13, tk, tb, (669, 8867, 3135))
    tc, tj = RotateAdd.kernel(9, 14, 13, tc, tj, (669, 8867, 3135))
    ti, td = RotateNeg.kernel(9, 14, 13, ti, td, (669, 8867, 3135))
    tu, t1 = RotatePi4Add(12, 13, tu, t1, (5793, 5793))
    tq, t5 = RotatePi4Add(12, 13, tq, t5, (5793, 5793))
    tp, t6 = RotatePi4Sub(12, 13, tp, t6, (5793, 5793))
    tm, t9 = RotatePi4Add(12, 13, tm, t9, (5793, 5793))
    te, th = RotatePi4Add(12, 13, te, th, (5793, 5793))
    return [t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, ta, tb, tc, td, te, tf,
            tg, th, ti, tj, tk, tl, tm, tn, to, tp, tq, tr, ts, tt, tu, tv]


# ---- 64-point ----
def daala_fdct64(c):
    asym = [None] * 32
    half = [None] * 32
    for i in range(16):
        j = i * 2
        ah, cc = butterfly_neg(c[j], c[63 - j])
        b, dh = butterfly_add(c[j + 1], c[63 - j - 1])
        half[i] = ah
        half[31 - i] = dh
        asym[i] = b
        asym[31 - i] = cc
    a = []
    for i in range(16):
        a += [half[i], asym[i]]
    lo = daala_fdct_ii_32_asym(a)
    b = []
    for i in range(16):
        b += [asym[31 - i], half[31 - i]]
    hi = daala_fdst_iv_32_asym(*b)
    tmp = lo + hi[::-1]
    out = [None] * 64
    order = [0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15]
    for i, j in enumerate(order):
        out[0 + i * 4] = tmp[0 + j]
        out[1 + i * 4] = tmp[32 + j]
        out[2 + i * 4] = tmp[16 + j]
        out[3 + i * 4] = tmp[48 + j]
    return out


def fidentity(c):
    return list(c)


def fwht4(c):
    x0, x1, x2, x3 = c
    s0 = add(x0, x1)
    s1 = sub(x3, x2)
    s2 = sub_avg(s0, s1)
    q1 = sub(s2, x2)
    q0 = sub(s0, q1)
    q3 = sub(s2, x1)
    q2 = add(s1, q3)
    return [q0, q1, q2, q3]


# ---- 2-D driver ----
# TxfmType order: DCT4,DCT8,DCT16,DCT32,DCT64,ADST4,ADST8,ADST16,Id4,Id8,Id16,Id32,WHT4
TXFM_FUNCS = [daala_fdct4, daala_fdct8, daala_fdct16, daala_fdct32, daala_fdct64,
              daala_fdst_vii_4, daala_fdst8, daala_fdst16,
              fidentity, fidentity, fidentity, fidentity, fwht4]
TXFM_LEN = [4, 8, 16, 32, 64, 4, 8, 16, 4, 8, 16, 32, 4]

# TxSize enum order (transform/mod.rs:101-123) -> (w, h)
TX_DIMS = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8),
           (16, 32), (32, 16), (32, 64), (64, 32), (4, 16), (16, 4), (8, 32), (32, 8),
           (16, 64), (64, 16)]
# 1-D types: 0 DCT, 1 ADST, 2 FLIPADST, 3 IDTX, 4 WHT
VTX_TAB = [0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3, 4]
HTX_TAB = [0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2, 4]
# AV1_TXFM_TYPE_LS[size_idx][1d type] -> TxfmType index (None = invalid)
TXFM_TYPE_LS = [[0, 5, 5, 8, 12], [1, 6, 6, 9, None], [2, 7, 7, 10, None],
                [3, None, None, 11, None], [4, None, None, None, None]]
_S_A = [[4, -1, 0], [2, 0, 1], [0, 0, 3]]
_S_B = [[4, -2, 0], [2, 0, 0], [0, 0, 2]]
_S_C = [[4, -1, -2], [2, 0, -1], [0, 0, 1]]
FWD_SHIFT = [[[3, 0, 0], [2, 0, 1], [0, 0, 3]], _S_A, _S_A, _S_B, _S_C, _S_A, _S_A, _S_A, _S_A,
             _S_B, _S_B, _S_C, _S_C, _S_A, _S_A, _S_A, _S_A, _S_B, _S_B]
UD_FLIP = {4, 8, 14, 6}
LR_FLIP = {5, 7, 15, 6}


def valid_av1_transform(tx_size, tx_type):
    w, h = TX_DIMS[tx_size]
    m = max(w, h)
    if tx_type == 16:
        return (w, h) == (4, 4)
    if m == 64:
        return tx_type == 0
    if m == 32:
        return tx_type in (0, 9)
    return True


def _round_shift_array(a, bit):
    # av1_round_shift_array (transform/mod.rs:317-331)
    if bit == 0:
        return a
    if bit > 0:
        return (a + I32((1 << bit) >> 1)) >> I32(bit)
    return a << I32(-bit)


def forward_transform(residual, tx_size, tx_type, bd):
    """residual: int array (..., H, W) (values as i16).  Returns int32 array
    (..., W*H) in the reference's output order (forward.rs:135-159)."""
    assert valid_av1_transform(tx_size, tx_type)
    w, h = TX_DIMS[tx_size]
    res = np.asarray(residual).astype(I32)
    lead = res.shape[:-2]
    res = res.reshape((-1, h, w))
    n = res.shape[0]
    wi, hi = w.bit_length() - 3, h.bit_length() - 3
    tcol = TXFM_TYPE_LS[hi][VTX_TAB[tx_type]]
    trow = TXFM_TYPE_LS[wi][HTX_TAB[tx_type]]
    shift = [0, 0, 2] if tx_type == 16 else FWD_SHIFT[tx_size][(bd - 8) // 2]
    ud, lr = tx_type in UD_FLIP, tx_type in LR_FLIP
    # columns: one lane per (block, column)
    src = res[:, ::-1, :] if ud else res
    cols = [np.ascontiguousarray(src[:, r, :]).reshape(-1) for r in range(h)]
    cols = [_round_shift_array(c, -shift[0]) for c in cols]
    cols = TXFM_FUNCS[tcol](cols)
    cols = [_round_shift_array(c, -shift[1]) for c in cols]
    buf = np.stack([c.reshape(n, w) for c in cols], axis=1)  # (n, h, w)
    if lr:
        buf = buf[:, :, ::-1]
    # rows: one lane per (block, row)
    rows = [np.ascontiguousarray(buf[:, :, c]).reshape(-1) for c in range(w)]
    rows = TXFM_FUNCS[trow](rows)
    rows = [_round_shift_array(c, -shift[2]) for c in rows]
    out2d = np.stack([c.reshape(n, h) for c in rows], axis=2)  # (n, h, w)
    out = np.zeros((n, w * h), dtype=I32)
    ostride = min(h, 32)
    for r in range(h):
        base = (r >= 32) * ostride * min(w, 32)
        for cg in range(0, w, 32):
            for c in range(min(w, 32)):
                out[:, base + h * cg + c * ostride + (r & 31)] = out2d[:, r, c + cg]
    return out.reshape(lead + (w * h,))
