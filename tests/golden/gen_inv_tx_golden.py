#!/usr/bin/env python3
"""Generate tests/golden/inv_tx_golden.npz FROM THE REFERENCE'S OWN SOURCE TEXT.

Same method as gen_fwd_tx_golden.py: there is no Rust toolchain here, but the
reference's inverse 1-D transforms (src/transform/inverse.rs:35-1588) are
straight-line integer code -- `let stgN = [ ... ];` array literals of
`half_btf(..)` / `clamp_value(..)` calls.  This script reads that file where
it lies under /root/reference (nothing is copied into this repository),
rewrites each `fn` into a Python `def` with a few regular expressions and
executes it on int32 NumPy lanes (wrapping arithmetic = Rust release mode).
`half_btf` / `clamp_value` restate src/transform/mod.rs:297-315; the cosine
tables are parsed from the reference text.  The 1-D outputs are therefore
produced by the reference's own networks and constants.

Only the 2-D driver (src/transform/inverse.rs:1633-1705) is hand-stated here
(it is generic Rust over iterators), independently of oracle/inv_tx.c.

Run in the build container:   python tests/golden/gen_inv_tx_golden.py
Output: tests/golden/inv_tx_golden.npz (committed).
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/src/transform/inverse.rs"
HERE = os.path.dirname(os.path.abspath(__file__))
I32 = np.int32

SQRT2, INV_SQRT2, SQRT2_BITS = 5793, 2896, 12   # src/transform/mod.rs:47-49


def half_btf(w0, in0, w1, in1, bit):
    # transform/mod.rs:297-307 (wrapping i32)
    r = (in0 * I32(w0)) + (in1 * I32(w1))
    return (r + I32(1 << (bit - 1))) >> I32(bit)


def clamp_value(v, bit):
    return np.clip(v, -(1 << (bit - 1)), (1 << (bit - 1)) - 1).astype(I32)


def round_shift(v, bit):
    return (v + I32((1 << bit) >> 1)) >> I32(bit)


def load_reference_fns():
    src = open(REF).read()
    env = {"half_btf": half_btf, "clamp_value": clamp_value, "round_shift": round_shift,
           "SQRT2": SQRT2, "INV_COS_BIT": 12, "np": np, "I32": I32, "_rng": range}
    for name in ("COSPI_INV", "SINPI_INV"):
        m = re.search(r"static %s: \[i32; \d+\] = \[(.*?)\];" % name, src, re.S)
        env[name] = [int(x) for x in re.findall(r"-?\d+", m.group(1))]
    # cut out every `fn av1_*(input, output, range) { ... }` up to the dispatch table
    body = src[:src.index("type InvTxfmFn")]
    fns = re.findall(r"(?:pub )?fn (av1_\w+)\(\s*input: &\[i32\], output: &mut \[i32\], _?range: usize,?\s*\) \{\n(.*?)\n\}\n",
                     body, re.S)
    names = []
    for name, text in fns:
        names.append(name)
        lines = []
        for ln in text.split("\n"):
            ln = re.sub(r"//.*$", "", ln).rstrip()
            if not ln.strip() or "assert!" in ln:
                continue
            ln = re.sub(r"^(\s*)let mut (\w+): \[i32; (\d+)\] = \[0; \d+\];",
                        r"\1\2 = [None] * \3", ln)
            ln = re.sub(r"^(\s*)let (\w+) =", r"\1\2 =", ln)
            ln = ln.replace("&mut ", "").replace("&", "")
            ln = re.sub(r"output\[\.\.(\d+)\]\.reverse\(\)", r"output[:\1] = output[:\1][::-1]", ln)
            lines.append(ln)
        py = "\n".join(lines)
        # identity kernels are iterator one-liners: rewrite the closure form
        m = re.search(r"output\[\.\.(\d+)\]\s*\.iter_mut\(\)\s*\.zip\(input\[\.\.\d+\]\.iter\(\)\)\s*"
                      r"\.for_each\(\|\(outp, inp\)\| \*outp = (.*?)\);", py, re.S)
        if m:
            n, expr = int(m.group(1)), m.group(2).replace("*inp", "input[i]")
            py = "  for i in _rng(%d):\n    output[i] = %s" % (n, expr)
        code = "def %s(input, output, range):\n%s\n" % (name, py)
        exec(compile(code, "<reference:%s>" % name, "exec"), env)
    return env, names


# TxfmType order shared with the oracle: row = 1-D class, col = log2(n) - 2
FN_TABLE = {
    ("dct", 4): "av1_idct4", ("dct", 8): "av1_idct8", ("dct", 16): "av1_idct16",
    ("dct", 32): "av1_idct32", ("dct", 64): "av1_idct64",
    ("adst", 4): "av1_iadst4", ("adst", 8): "av1_iadst8", ("adst", 16): "av1_iadst16",
    ("flipadst", 4): "av1_iflipadst4", ("flipadst", 8): "av1_iflipadst8",
    ("flipadst", 16): "av1_iflipadst16",
    ("identity", 4): "av1_iidentity4", ("identity", 8): "av1_iidentity8",
    ("identity", 16): "av1_iidentity16", ("identity", 32): "av1_iidentity32",
    ("wht", 4): "av1_iwht4",
}
CLASSES = ["dct", "adst", "flipadst", "identity", "wht"]

# TxSize dims (transform/mod.rs:101-167) and 1-D classes per TxType (364-402)
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
VTX = [0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3, 4]
HTX = [0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2, 4]
INV_INTERMEDIATE_SHIFTS = [0, 1, 2, 2, 2, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2]


def valid(tx_size, tx_type):
    m = max(TX_W[tx_size], TX_H[tx_size])
    if tx_type == 16:
        return tx_size == 0
    if m == 64:
        return tx_type == 0
    if m == 32:
        return tx_type in (0, 9)
    return True


def run_1d(env, cls, n, x, rng_bits):
    """x: (lanes, n) int32 -> (lanes, n)"""
    f = env[FN_TABLE[(CLASSES[cls], n)]]
    inp = [x[:, i].copy() for i in range(n)]
    out = [None] * n
    f(inp, out, rng_bits)
    return np.stack([np.asarray(o, dtype=I32) for o in out], axis=1)


def inverse_transform_add_2d(env, coeffs, dst, tx_size, tx_type, bd):
    """coeffs: (n, min(w,32)*min(h,32)) int32 in the forward transform's
    transposed layout; dst: (n, h, w) int32 prediction -> reconstructed pixels.
    Hand-stated from inverse.rs:1633-1705."""
    w, h = TX_W[tx_size], TX_H[tx_size]
    n = coeffs.shape[0]
    hc, wc = min(h, 32), min(w, 32)
    rect1 = abs(int(np.log2(w)) - int(np.log2(h))) == 1
    lossless = tx_type == 16
    rng1 = bd + 8
    buf = np.zeros((n, h, w), I32)
    for r in range(hc):
        tin = np.zeros((n, w), I32)
        raw = coeffs[:, r::hc][:, :wc].astype(I32)        # input[r..].step_by(min(h,32))
        if rect1:
            raw = round_shift(raw * I32(INV_SQRT2), SQRT2_BITS)
        elif lossless:
            raw = raw >> I32(2)
        tin[:, :raw.shape[1]] = clamp_value(raw, rng1)
        buf[:, r, :] = run_1d(env, HTX[tx_type], w, tin, rng1)
    rng2 = max(bd + 6, 16)
    out = dst.astype(I32).copy()
    for c in range(w):
        tin = clamp_value(round_shift(buf[:, :, c], INV_INTERMEDIATE_SHIFTS[tx_size]), rng2)
        tout = run_1d(env, VTX[tx_type], h, tin, rng2)
        r = tout if lossless else round_shift(tout, 4)
        out[:, :, c] = np.clip(out[:, :, c] + r, 0, (1 << bd) - 1)
    return out


def main():
    env, names = load_reference_fns()
    print("executing reference 1-D kernels:", ", ".join(names))
    rng = np.random.default_rng(20260922)
    out = {}
    # ---- 1-D vectors: moderate and extreme (clamp-exercising) inputs ----
    for (cls, n), fn in FN_TABLE.items():
        ci = CLASSES.index(cls)
        for rb in (16, 18, 20):
            lim = 1 << (rb - 1)
            x = np.concatenate([
                rng.integers(-(lim >> 3), lim >> 3, (24, n)),
                rng.integers(-lim, lim, (24, n)),
                np.full((1, n), lim - 1), np.full((1, n), -lim),
                np.eye(n, dtype=np.int64)[: min(n, 8)] * (lim - 1),
            ]).astype(I32)
            out["d1_%s_%d_r%d_in" % (cls, n, rb)] = x
            out["d1_%s_%d_r%d_out" % (cls, n, rb)] = run_1d(env, ci, n, x, rb)
    # ---- 2-D vectors ----
    ncase = 0
    for ts in range(19):
        w, h = TX_W[ts], TX_H[ts]
        area = min(w, 32) * min(h, 32)
        for tt in range(17):
            if not valid(ts, tt):
                continue
            for bd in (8, 10, 12):
                nb = 4
                amp = 1 << (bd + 3)
                co = rng.integers(-amp, amp, (nb, area)).astype(I32)
                co[1] = rng.integers(-amp * 64, amp * 64, area)          # clamp-exercising
                co[2, 1:] = 0                                            # DC only
                co[3] = (rng.integers(-amp, amp, area) * (rng.random(area) < 0.1)).astype(I32)
                pred = rng.integers(0, 1 << bd, (nb, h, w)).astype(I32)
                rec = inverse_transform_add_2d(env, co, pred, ts, tt, bd)
                k = "d2_%d_%d_%d" % (ts, tt, bd)
                out[k + "_co"] = co
                out[k + "_pred"] = pred.astype(np.uint16)
                out[k + "_rec"] = rec.astype(np.uint16)
                ncase += 1
    path = os.path.join(HERE, "inv_tx_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d 1-D sets, %d 2-D cases, %.1f KiB" %
          (path, len(FN_TABLE) * 3, ncase, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    sys.exit(main())
