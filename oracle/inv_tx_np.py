"""oracle/inv_tx_np.py -- TEST INFRASTRUCTURE (not product code).

NumPy restatement of the reference's inverse transform
(src/transform/inverse.rs; AV1 spec 7.13.2 / 7.13.3), written rule-based
instead of stage-by-stage:

  half_btf, clamp_value           src/transform/mod.rs:297-315
  av1_idct4/8/16/32/64            src/transform/inverse.rs:71-91,160-207,306-401,586-884,893-1588
  av1_iadst4/8/16 (+flip)         src/transform/inverse.rs:93-148,209-297,403-577
  av1_iidentity4/8/16/32          src/transform/inverse.rs:150-155,299-304,579-584,886-891
  av1_iwht4                       src/transform/inverse.rs:35-53
  inverse_transform_add (2-D)     src/transform/inverse.rs:1633-1711

The reference spells every butterfly stage out as an array literal; here the
inverse DCT is the recursion those stages implement:

  idct(N):  even half = idct(N/2) of the even-frequency inputs,
            odd half  = odd_part(N/2) of the odd-frequency inputs taken in
                        bit-reversed order,
            out[i] = clamp(even[i] + odd[M-1-i]), out[N-1-i] = clamp(even[i] - odd[M-1-i])

  odd_part(M): a first rotation stage on mirrored pairs (j, M-1-j), then
  alternating clamped add/sub stages on groups of G = 2, 4, .. M/2 and
  rotation stages on the inner halves of groups of 2G (rules in the code).

Rotations are pairs of half_btf (12-bit cosines, rounding add, arithmetic
shift, wrapping i32); add/sub stages clamp to `rng` bits exactly where the
reference calls clamp_value.  Every scalar of the reference is an int32
ndarray lane here.  Pinned bit-exactly by tests/golden/inv_tx_golden.npz
(vectors produced by executing the reference's own source text) and by
forward->inverse round trips (tests/test_oracle_inv_tx.py).

The functions also run on symbolic lanes (`Sym`): tools/gen_inv_tx1d.py traces
them into straight-line SSA for oracle/inv_tx_1d.inc and
rav1e_amd/csrc/inv_tx_1d.inc.
"""
import numpy as np

I32 = np.int32

# cos(j*pi/128) in Q12, j = 0..63 (AV1 spec Cos128 table; inverse.rs:55-62 carries
# the same values) -- generated, not transcribed:
COSPI = [int(round(4096 * np.cos(j * np.pi / 128))) for j in range(64)]
# sin(k*pi/9)*sqrt(2)*2/3 in Q12 (AV1 spec Sinpi table; inverse.rs:64)
SINPI = [0] + [int(round(4096 * np.sin(k * np.pi / 9) * 2 * np.sqrt(2) / 3)) for k in range(1, 5)]
SQRT2 = 5793          # transform/mod.rs:48
INV_SQRT2 = 2896      # transform/mod.rs:49


class Sym:
    """A traced value for the SSA generator."""
    __slots__ = ("name", "tape")

    def __init__(self, name, tape):
        self.name, self.tape = name, tape

    def emit(self, expr):
        n = "t%d" % len(self.tape)
        self.tape.append((n, expr))
        return Sym(n, self.tape)


def _c(j):
    """signed cosine: index j > 0 -> COSPI[j], j < 0 -> -COSPI[-j] (j != 0)"""
    return COSPI[j] if j >= 0 else -COSPI[-j]


def btf(w0, a, w1, b):
    """half_btf(w0, a, w1, b, 12)  (transform/mod.rs:297-307)"""
    if isinstance(a, Sym):
        return a.emit("ITX_BTF(%d, %s, %d, %s)" % (w0, a.name, w1, b.name))
    return ((a * I32(w0)) + (b * I32(w1)) + I32(2048)) >> I32(12)


def cadd(a, b, rng):
    """clamp_value(a + b, range)"""
    if isinstance(a, Sym):
        return a.emit("ITX_CLAMP(ITX_ADD(%s, %s))" % (a.name, b.name))
    return np.clip(a + b, -(1 << (rng - 1)), (1 << (rng - 1)) - 1).astype(I32)


def csub(a, b, rng):
    """clamp_value(a - b, range)"""
    if isinstance(a, Sym):
        return a.emit("ITX_CLAMP(ITX_SUB(%s, %s))" % (a.name, b.name))
    return np.clip(a - b, -(1 << (rng - 1)), (1 << (rng - 1)) - 1).astype(I32)


def neg(a):
    if isinstance(a, Sym):
        return a.emit("ITX_NEG(%s)" % a.name)
    return -a


def add(a, b):
    if isinstance(a, Sym):
        return a.emit("ITX_ADD(%s, %s)" % (a.name, b.name))
    return a + b


def sub(a, b):
    if isinstance(a, Sym):
        return a.emit("ITX_SUB(%s, %s)" % (a.name, b.name))
    return a - b


def mulc(a, m):
    """wrapping i32 multiply by a constant"""
    if isinstance(a, Sym):
        return a.emit("ITX_MUL(%s, %d)" % (a.name, m))
    return a * I32(m)


def rshift_round(a, bit):
    """round_shift(a, bit) (v_frame math::round_shift)"""
    if isinstance(a, Sym):
        return a.emit("ITX_RSHIFT(%s, %d)" % (a.name, bit))
    return (a + I32((1 << bit) >> 1)) >> I32(bit)


def sar(a, bit):
    if isinstance(a, Sym):
        return a.emit("ITX_SAR(%s, %d)" % (a.name, bit))
    return a >> I32(bit)


def brev(nbits, x):
    r = 0
    for i in range(nbits):
        r |= ((x >> i) & 1) << (nbits - 1 - i)
    return r


def ilog2(n):
    return n.bit_length() - 1


# ---------------------------------------------------------------- inverse DCT
def odd_part(e, rng):
    """The odd half of an N = 2M point inverse DCT.  e[j] = the odd-frequency
    input 2*brev(j)+1.  Returns o with out[i] = even[i] +- o[M-1-i]."""
    M = len(e)
    N = 2 * M
    lm = ilog2(M)
    e = list(e)
    # first rotation stage: mirrored pairs (j, M-1-j), angle from the frequency
    for j in range(M // 2):
        m = M - 1 - j
        k = 2 * brev(lm, j) + 1
        a = 64 - (64 // N) * k
        x, y = e[j], e[m]
        e[j] = btf(_c(a), x, -_c(64 - a), y)
        e[m] = btf(_c(64 - a), x, _c(a), y)
    G = 2
    while G <= M // 2:
        # clamped add/sub on groups of G: odd groups are mirrored
        for b in range(0, M, G):
            flipped = (b // G) & 1
            for p in range(G // 2):
                lo, hi = b + p, b + G - 1 - p
                x, y = e[lo], e[hi]
                if not flipped:
                    e[lo], e[hi] = cadd(x, y, rng), csub(x, y, rng)
                else:
                    e[lo], e[hi] = csub(y, x, rng), cadd(x, y, rng)
        # rotations on the inner halves of groups of 2G (lower half of the
        # array), each paired with its mirror image M-1-j
        G2 = 2 * G
        nq2 = M // G2            # twice the number of 2G-groups in the lower half
        for j in range(M // 2):
            p = j % G2
            if not (G2 // 4 <= p < 3 * G2 // 4):
                continue
            if nq2 == 1:
                a = 32
            else:
                nq = nq2 // 2
                a = (16 // nq) * (1 + 4 * brev(ilog2(nq), j // G2))
            m = M - 1 - j
            x, y = e[j], e[m]
            if p < G2 // 2:      # first inner quarter
                e[j] = btf(-_c(a), x, _c(64 - a), y)
                e[m] = btf(_c(64 - a), x, _c(a), y)
            else:                # second inner quarter
                e[j] = btf(-_c(64 - a), x, -_c(a), y)
                e[m] = btf(-_c(a), x, _c(64 - a), y)
        G = G2
    return e


def idct(x, rng):
    """N-point inverse DCT on natural-order frequencies x[0..N-1]."""
    N = len(x)
    if N == 2:
        return [btf(_c(32), x[0], _c(32), x[1]), btf(_c(32), x[0], -_c(32), x[1])]
    M = N // 2
    even = idct([x[2 * i] for i in range(M)], rng)
    lm = ilog2(M)
    o = odd_part([x[2 * brev(lm, j) + 1] for j in range(M)], rng)
    out = [None] * N
    for i in range(M):
        out[i] = cadd(even[i], o[M - 1 - i], rng)
        out[N - 1 - i] = csub(even[i], o[M - 1 - i], rng)
    return out


# --------------------------------------------------------------- inverse ADST
ADST_OUT = {8: [0, 4, 6, 2, 3, 7, 5, 1],
            16: [0, 8, 12, 4, 6, 14, 10, 2, 3, 11, 15, 7, 5, 13, 9, 1]}


def _rot_p(x, y, a):
    return btf(_c(a), x, _c(64 - a), y), btf(_c(64 - a), x, -_c(a), y)


def _rot_q(x, y, a):
    return btf(-_c(64 - a), x, _c(a), y), btf(_c(a), x, _c(64 - a), y)


def iadst(x, rng):
    """8- and 16-point inverse ADST (inverse.rs:218-297, 409-577)."""
    N = len(x)
    e = [None] * N
    for i in range(N // 2):
        e[2 * i], e[2 * i + 1] = x[N - 1 - 2 * i], x[2 * i]
    step = 128 // N
    for i in range(N // 2):
        e[2 * i], e[2 * i + 1] = _rot_p(e[2 * i], e[2 * i + 1], step // 4 + step * i)
    h = N // 2
    while h >= 2:
        B2 = 2 * h
        for b in range(0, N, B2):       # clamped add/sub, stride h inside blocks of 2h
            for p in range(h):
                u, v = e[b + p], e[b + p + h]
                e[b + p], e[b + p + h] = cadd(u, v, rng), csub(u, v, rng)
        # rotations on adjacent pairs in the upper half of every 2h-block
        npairs, base = h // 2, 128 // B2
        for b in range(0, N, B2):
            for i in range(npairs):
                j = b + h + 2 * i
                if npairs == 1:
                    e[j], e[j + 1] = _rot_p(e[j], e[j + 1], 32)
                elif i < npairs // 2:
                    e[j], e[j + 1] = _rot_p(e[j], e[j + 1], base + 4 * base * i)
                else:
                    e[j], e[j + 1] = _rot_q(e[j], e[j + 1], base + 4 * base * (i - npairs // 2))
        h //= 2
    out = [None] * N
    for i, src in enumerate(ADST_OUT[N]):
        out[i] = e[src] if i % 2 == 0 else neg(e[src])
    return out


def iadst4(x, rng=None):
    """inverse.rs:102-148 (sinpi form; no intermediate clamps)"""
    x0, x1, x2, x3 = x
    s0, s1 = mulc(x0, SINPI[1]), mulc(x0, SINPI[2])
    s2 = mulc(x1, SINPI[3])
    s3, s4 = mulc(x2, SINPI[4]), mulc(x2, SINPI[1])
    s5, s6 = mulc(x3, SINPI[2]), mulc(x3, SINPI[4])
    s7 = add(sub(x0, x2), x3)
    s0 = add(add(s0, s3), s5)
    s1 = sub(sub(s1, s4), s6)
    s3 = s2
    s2 = mulc(s7, SINPI[3])
    y0, y1, y2 = add(s0, s3), add(s1, s3), s2
    y3 = sub(add(s0, s1), s3)
    return [rshift_round(y, 12) for y in (y0, y1, y2, y3)]


def iidentity(x, rng=None):
    n = len(x)
    if n == 4:
        return [rshift_round(mulc(v, SQRT2), 12) for v in x]
    if n == 8:
        return [mulc(v, 2) for v in x]
    if n == 16:
        return [rshift_round(mulc(v, 2 * SQRT2), 12) for v in x]
    return [mulc(v, 4) for v in x]


def iwht4(x, rng=None):
    x0, x1, x2, x3 = x
    s0, s2 = add(x0, x1), sub(x2, x3)
    s4 = sar(sub(s0, s2), 1)
    s3, s1 = sub(s4, x3), sub(s4, x1)
    return [sub(s0, s3), s3, s1, add(s2, s1)]


def inv_1d(cls, x, rng):
    """cls: 0 DCT, 1 ADST, 2 FLIPADST, 3 IDTX, 4 WHT (INV_TXFM_FNS rows, inverse.rs:1593-1626)"""
    n = len(x)
    if cls == 0:
        return idct(x, rng)
    if cls in (1, 2):
        o = iadst4(x) if n == 4 else iadst(x, rng)
        return o[::-1] if cls == 2 else o
    if cls == 3:
        return iidentity(x)
    return iwht4(x)


# ------------------------------------------------------------------ 2-D driver
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
VTX = [0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3, 4]
HTX = [0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2, 4]
INV_INTERMEDIATE_SHIFTS = [0, 1, 2, 2, 2, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2]


def _clampv(v, bits):
    return np.clip(v, -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(I32)


def inverse_transform_add(coeffs, pred, tx_size, tx_type, bd):
    """coeffs (n, min(w,32)*min(h,32)) in the forward transform's transposed
    layout, pred (n, h, w) -> reconstruction (n, h, w) int32.
    inverse.rs:1633-1705: rows first (input read with stride min(h,32)),
    rect 2:1 scaling, clamps, intermediate shift, columns, >>4, add, clamp."""
    w, h = TX_W[tx_size], TX_H[tx_size]
    n = coeffs.shape[0]
    hc, wc = min(h, 32), min(w, 32)
    co = coeffs.astype(I32).reshape(n, wc, hc)          # [col][row]: transposed storage
    rect1 = abs(ilog2(w) - ilog2(h)) == 1
    lossless = tx_type == 16
    r1 = bd + 8
    buf = np.zeros((n, h, w), I32)
    for r in range(hc):
        raw = co[:, :, r]
        if rect1:
            raw = (raw * I32(INV_SQRT2) + I32(2048)) >> I32(12)
        elif lossless:
            raw = raw >> I32(2)
        row = np.zeros((n, w), I32)
        row[:, :wc] = _clampv(raw, r1)
        o = inv_1d(HTX[tx_type], [row[:, i] for i in range(w)], r1)
        buf[:, r, :] = np.stack(o, axis=1)
    r2 = max(bd + 6, 16)
    sh = INV_INTERMEDIATE_SHIFTS[tx_size]
    out = pred.astype(I32).copy()
    for c in range(w):
        col = _clampv((buf[:, :, c] + I32((1 << sh) >> 1)) >> I32(sh), r2)
        o = np.stack(inv_1d(VTX[tx_type], [col[:, i] for i in range(h)], r2), axis=1)
        res = o if lossless else (o + I32(8)) >> I32(4)
        out[:, :, c] = np.clip(out[:, :, c] + res, 0, (1 << bd) - 1)
    return out
