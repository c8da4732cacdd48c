#!/usr/bin/env python3
"""Generate tests/golden/fwd_tx_golden.npz FROM THE REFERENCE'S OWN SOURCE TEXT.

There is no Rust toolchain in the build container, so the reference cannot be
compiled.  Its forward 1-D transforms are, however, straight-line integer code
inside one macro body (src/transform/forward_shared.rs:179-1797).  This script
reads that file where it lies under /root/reference (it is NOT copied into
this repository), rewrites each `fn` body into Python statements with a few
regular expressions (Rust `let` tuples, method calls and turbofish constants
map 1:1 onto Python syntax), and executes the result with a value class whose
six primitive ops restate `impl TxOperations for i32`
(src/transform/forward.rs:37-65).  The outputs are therefore produced by the
reference's butterfly networks and constants, not by our restatement.

Only the 2-D driver (column pass, flips, shifts, transposed 32x32-chunked
store; src/transform/forward.rs:83-160) is hand-stated here, because it is
generic Rust (MaybeUninit, slices) rather than straight-line arithmetic.

Run in the build container:   python tests/golden/gen_fwd_tx_golden.py
Output: tests/golden/fwd_tx_golden.npz (committed; the GPU box has no
/root/reference and only reads the .npz).
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/src/transform/forward_shared.rs"
HERE = os.path.dirname(os.path.abspath(__file__))
I32 = np.int32


class V:
    """i32 lanes with the reference's TxOperations (forward.rs:37-65)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = np.asarray(v, dtype=I32)

    def tx_mul(self, shift, mul):
        return V(((self.v * I32(mul)) + I32((1 << shift) >> 1)) >> I32(shift))

    def rshift1(self):
        return V((self.v + (self.v < 0).astype(I32)) >> I32(1))

    def add(self, b):
        return V(self.v + b.v)

    def sub(self, b):
        return V(self.v - b.v)

    def add_avg(self, b):
        return V((self.v + b.v) >> I32(1))

    def sub_avg(self, b):
        return V((self.v - b.v) >> I32(1))

    def copy_fn(self):
        return self


class Buf:
    """A `[T; N]` / `&mut [T]` with sub-slice views."""

    def __init__(self, n=None, store=None, lo=0, hi=None):
        self.s = store if store is not None else [None] * n
        self.lo = lo
        self.hi = hi if hi is not None else len(self.s)

    def view(self, lo, hi):
        return Buf(store=self.s, lo=self.lo + lo, hi=self.lo + hi)

    def __getitem__(self, i):
        return self.s[self.lo + i]

    def __setitem__(self, i, x):
        self.s[self.lo + i] = x

    def __len__(self):
        return self.hi - self.lo

    def reverse(self):
        self.s[self.lo:self.hi] = self.s[self.lo:self.hi][::-1]

    def store(self, vals):
        for i, x in enumerate(vals):
            self.s[self.lo + i] = x


def _match_brace(text, start):
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                return i
    raise ValueError("unbalanced")


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def _expr(e):
    e = re.sub(r"//[^\n]*", "", e)
    e = re.sub(r"\s+", " ", e).strip()
    e = re.sub(r"&mut (\w+)\[(\d+)\.\.(\d+)\]", r"\1.view(\2,\3)", e)
    e = re.sub(r"&mut (\w+)", r"\1", e)
    e = re.sub(r"(\w+)\[(\d+)\.\.(\d+)\]", r"\1.view(\2,\3)", e)
    e = re.sub(r"\[\(?T::zero\(\)(?:, T::zero\(\)\))?; (\d+)\]", r"Buf(\1)", e)
    # Type::kernel::<A, B, C>(args)  ->  Type.kernel(A, B, C, args)
    e = re.sub(r"(\w+)::(\w+)::<([^>]*)>\(", r"\1.\2(\3, ", e)
    e = re.sub(r"\.(\w+)::<([^>]*)>\(", r".\1(\2, ", e)
    e = re.sub(r"Self::(\w+)\(", r"self.\1(", e)
    e = re.sub(r"\bSelf\.", "self.", e)
    # tuple fields p0.0 / m.2
    e = re.sub(r"\b([a-z_]\w*)\.(\d)\b", r"\1[\2]", e)
    return e


def _stmt(st):
    st = re.sub(r"//[^\n]*", "", st)
    st = re.sub(r"#\[[^\]]*\]", "", st)
    st = re.sub(r"\s+", " ", st).strip()
    if not st or st.startswith("assert!"):
        return None
    m = re.match(r"store_coeffs!\( ?(\w+), (.*)\)$", st)
    if m:
        return "%s.store([%s])" % (m.group(1), _expr(m.group(2)))
    m = re.match(r"let (?:mut )?([^=:]+?)(?:: [^=]+)? = (.*)$", st)
    if m:
        return "%s = %s" % (m.group(1).strip(), _expr(m.group(2)))
    return _expr(st)


def _body_to_py(body, indent="    "):
    # drop nested fn definitions (translated separately) and bare blocks
    out = []
    i = 0
    flat = ""
    while i < len(body):
        m = re.compile(r"fn \w+<T: TxOperations>\(").search(body, i)
        if not m:
            flat += body[i:]
            break
        # cut back to the start of the attribute/`$($s)*` prefix on that line
        line_start = body.rfind("\n", 0, m.start()) + 1
        # also drop preceding attribute lines (#[$m], #[inline])
        prefix = body[i:line_start]
        prefix = re.sub(r"(\s*#\[[^\]]*\]\s*)+$", "\n", prefix)
        flat += prefix
        b0 = body.index("{", m.end())
        i = _match_brace(body, b0) + 1
    flat = flat.replace("{", " ").replace("}", " ")
    # array initialisers / type annotations contain ';' -- fold them first
    flat = re.sub(r"\[\(?T::zero\(\)(?:, T::zero\(\)\))?; (\d+)\]", r"Buf(\1)", flat)
    flat = re.sub(r": \[[^\]]*; \d+\]", "", flat)
    for st in flat.split(";"):
        py = _stmt(st)
        if py:
            out.append(indent + py)
    return "\n".join(out)


def translate(src):
    lo = src.index("macro_rules! impl_1d_tx")
    text = src[lo:]
    code = []
    # --- impl blocks: constants per rotation type ---
    consts = {}
    for m in re.finditer(r"impl<T: TxOperations> (\w+)<T> for (\w+) \{", text):
        end = _match_brace(text, m.end() - 1)
        blk = text[m.end():end]
        d = dict(re.findall(r"const (\w+):[^=]*= T::(\w+);", blk))
        consts[m.group(2)] = (m.group(1), d)
    # --- trait default methods ---
    traits = {}
    for m in re.finditer(r"trait (\w+)<T: TxOperations> \{", text):
        end = _match_brace(text, m.end() - 1)
        blk = text[m.end():end]
        meths = []
        for f in re.finditer(r"fn (\w+)<([^>]*)>\(", blk):
            a0 = f.end()
            a1 = blk.index(")", a0)
            # argument list may contain nested parens (tuples)
            depth, k = 1, a0
            while depth:
                if blk[k] == "(":
                    depth += 1
                elif blk[k] == ")":
                    depth -= 1
                k += 1
            args = [a.split(":")[0].strip() for a in _split_args(blk[a0:k - 1])]
            cargs = re.findall(r"const (\w+): i32", f.group(2))
            b0 = blk.index("{", k)
            b1 = _match_brace(blk, b0)
            meths.append((f.group(1), cargs + args, blk[b0 + 1:b1]))
        traits[m.group(1)] = meths
    for tname, meths in traits.items():
        code.append("class %s:" % tname)
        for name, args, body in meths:
            code.append("  def %s(self, %s):" % (name, ", ".join(args)))
            lines = _body_to_py(body, "    ").split("\n")
            # last expression statement is the return value
            lines[-1] = "    return " + lines[-1].strip()
            code.append("\n".join(lines))
    for cname, (tname, d) in consts.items():
        code.append("class _%s(%s):" % (cname, tname))
        for k, fn in d.items():
            n = 1 if k == "SHIFT" else 2
            if n == 2:
                code.append("  def %s(self, a, b): return a.%s(b)" % (k, fn))
            else:
                code.append("  def %s(self, a): return a.%s()" % (k, fn))
        code.append("%s = _%s()" % (cname, cname))
    # --- free functions ---
    for m in re.finditer(r"fn (\w+)<T: TxOperations>\(", text):
        name = m.group(1)
        depth, k = 1, m.end()
        while depth:
            if text[k] == "(":
                depth += 1
            elif text[k] == ")":
                depth -= 1
            k += 1
        args = [a.split(":")[0].strip() for a in _split_args(text[m.end():k - 1])]
        b0 = text.index("{", k)
        has_ret = "->" in text[k:b0]
        b1 = _match_brace(text, b0)
        body = text[b0 + 1:b1]
        lines = _body_to_py(body, "  ").split("\n") if body.strip() else ["  pass"]
        if has_ret:
            lines[-1] = "  return " + lines[-1].strip()
        code.append("def %s(%s):\n%s" % (name, ", ".join(args), "\n".join(lines)))
    return "\n".join(code)


def load_reference_1d():
    src = open(REF).read()
    py = translate(src)
    ns = {"V": V, "Buf": Buf}
    exec(compile(py, "<forward_shared.rs translated>", "exec"), ns)
    return ns, py


# TxfmType order (forward_shared.rs:67-81, get_func 201-218)
TXFM = ["daala_fdct4", "daala_fdct8", "daala_fdct16", "daala_fdct32", "daala_fdct64",
        "daala_fdst_vii_4", "daala_fdst8", "daala_fdst16",
        "fidentity", "fidentity", "fidentity", "fidentity", "fwht4"]
TXFM_LEN = [4, 8, 16, 32, 64, 4, 8, 16, 4, 8, 16, 32, 4]


def run_1d(ns, ttype, x):
    """x: (lanes, n) int32 -> (lanes, n)"""
    n = TXFM_LEN[ttype]
    buf = Buf(n)
    for i in range(n):
        buf[i] = V(np.ascontiguousarray(x[:, i]))
    ns[TXFM[ttype]](buf)
    return np.stack([buf[i].v for i in range(n)], axis=1)


# ---- hand-stated 2-D driver (forward.rs:83-160), shifts/tables from
# forward_shared.rs:22-64 and transform/mod.rs:364-402 ----
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))


def forward_2d(ns, res, tx_size, tx_type, bd):
    import fwd_tx_np as F  # tables only (TX_DIMS, shifts, VTX/HTX)
    w, h = F.TX_DIMS[tx_size]
    wi, hi = w.bit_length() - 3, h.bit_length() - 3
    tcol = F.TXFM_TYPE_LS[hi][F.VTX_TAB[tx_type]]
    trow = F.TXFM_TYPE_LS[wi][F.HTX_TAB[tx_type]]
    shift = [0, 0, 2] if tx_type == 16 else F.FWD_SHIFT[tx_size][(bd - 8) // 2]
    ud, lr = tx_type in F.UD_FLIP, tx_type in F.LR_FLIP

    def rs(a, bit):
        if bit == 0:
            return a
        if bit > 0:
            return (a + I32((1 << bit) >> 1)) >> I32(bit)
        return a << I32(-bit)

    res = res.astype(I32)
    buf = np.zeros((h, w), dtype=I32)
    for c in range(w):
        col = res[::-1, c] if ud else res[:, c]
        col = rs(col.copy(), -shift[0])
        col = run_1d(ns, tcol, col[None, :])[0]
        col = rs(col, -shift[1])
        buf[:, (w - c - 1) if lr else c] = col
    out = np.zeros(w * h, dtype=I32)
    ostride = min(h, 32)
    for r in range(h):
        row = run_1d(ns, trow, buf[r][None, :])[0]
        row = rs(row, -shift[2])
        base = (r >= 32) * ostride * min(w, 32)
        for cg in range(0, w, 32):
            for c in range(min(w, 32)):
                out[base + h * cg + c * ostride + (r & 31)] = row[c + cg]
    return out


def main():
    import fwd_tx_np as F
    ns, py = load_reference_1d()
    rng = np.random.default_rng(20250921)
    out = {}
    # (a) 1-D vectors: small, mid and near-limit magnitudes
    for t in range(13):
        n = TXFM_LEN[t]
        x = np.concatenate([
            rng.integers(-255, 256, size=(64, n)),
            rng.integers(-(1 << 15), 1 << 15, size=(64, n)),
            rng.integers(-(1 << 19), 1 << 19, size=(32, n)),
        ]).astype(I32)
        out["in1d_%d" % t] = x
        out["out1d_%d" % t] = run_1d(ns, t, x)
    # (b) 2-D vectors: every valid (size, type), bd 8/10/12, one block each,
    # residual range = full (bd+1)-bit signed range as the reference's asm
    # tests use (src/asm/shared/transform/forward.rs:53-109 uses +-255 at bd 8)
    keys = []
    for bd in (8, 10, 12):
        lim = (1 << bd) - 1
        for ts in range(19):
            w, h = F.TX_DIMS[ts]
            for tt in range(17):
                if not F.valid_av1_transform(ts, tt):
                    continue
                res = rng.integers(-lim, lim + 1, size=(h, w)).astype(np.int16)
                k = "%d_%d_%d" % (bd, ts, tt)
                out["res2d_" + k] = res
                out["coef2d_" + k] = forward_2d(ns, res, ts, tt, bd)
                keys.append(k)
    out["keys2d"] = np.array(keys)
    path = os.path.join(HERE, "fwd_tx_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(keys), "2-D cases")


if __name__ == "__main__":
    if "--dump" in sys.argv:
        print(translate(open(REF).read()))
    else:
        main()
