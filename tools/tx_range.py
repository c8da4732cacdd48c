#!/usr/bin/env python3
"""Worst-case magnitude of every multiplier input of the forward transform.

Each 1-D network is (up to rounding) linear, so the worst-case |a| at a
multiplier fed by inputs bounded by B is B * (L1 norm of that node's impulse
responses) + a small rounding slack.  We measure the impulse responses by
running the NumPy restatement on scaled unit impulses, one lane per impulse,
and intercepting tx_mul.  Used to justify v_mul_i32_i24 (needs |a| < 2^23) in
the fused kernel, where the residual is bounded by the pixel range.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fwd_tx_np as F

AMP = 1 << 14   # impulse amplitude (keeps relative rounding error ~1e-4)


def l1_gain(ttype):
    """-> (max L1 gain over all tx_mul inputs, max L1 gain over outputs)"""
    n = F.TXFM_LEN[ttype]
    rec = []
    orig = F.tx_mul

    def spy(a, mul, shift):
        rec.append(np.abs(a.astype(np.float64)).sum() / AMP)
        return orig(a, mul, shift)
    F.tx_mul = spy
    try:
        x = (np.eye(n, dtype=np.int32) * AMP)
        outs = F.TXFM_FUNCS[ttype]([np.ascontiguousarray(x[:, i]) for i in range(n)])
    finally:
        F.tx_mul = orig
    og = max(np.abs(o.astype(np.float64)).sum() / AMP for o in outs)
    return (max(rec) if rec else 0.0), og


def analyse():
    rows = []
    for bd in (8, 10, 12):
        for ts, (w, h) in enumerate(F.TX_DIMS):
            for tt in range(16):
                if not F.valid_av1_transform(ts, tt):
                    continue
                sh = F.FWD_SHIFT[ts][(bd - 8) // 2]
                wi, hi = w.bit_length() - 3, h.bit_length() - 3
                tcol = F.TXFM_TYPE_LS[hi][F.VTX_TAB[tt]]
                trow = F.TXFM_TYPE_LS[wi][F.HTX_TAB[tt]]
                b0 = ((1 << bd) - 1) * (1 << sh[0])          # column input bound
                gm, go = l1_gain(tcol)
                col_mul = b0 * gm + 64
                col_out = b0 * go + 64
                b1 = col_out * 2.0 ** sh[1] + 1                 # after shift[1]
                gm2, go2 = l1_gain(trow)
                row_mul = b1 * gm2 + 64
                rows.append((bd, ts, tt, max(col_mul, row_mul, b0, b1)))
    return rows


def shifted_coefficient_bound():
    """{bit depth: max over sizes / types of |c << log_tx_scale| for residuals that come from pixels}: the magnitude
    the quantizer divides (quantize/mod.rs:296-318) -- what lets the fused kernels' quantizer run on 24-bit multipliers
    (csrc/quant_common.hpp, QParams::ac_m22: a + ac_offset < 2^22)"""
    worst = {8: 0.0, 10: 0.0, 12: 0.0}
    for bd in worst:
        for ts, (w, h) in enumerate(F.TX_DIMS):
            for tt in range(16):
                if not F.valid_av1_transform(ts, tt):
                    continue
                sh = F.FWD_SHIFT[ts][(bd - 8) // 2]
                tcol = F.TXFM_TYPE_LS[h.bit_length() - 3][F.VTX_TAB[tt]]
                trow = F.TXFM_TYPE_LS[w.bit_length() - 3][F.HTX_TAB[tt]]
                b0 = ((1 << bd) - 1) * (1 << sh[0])
                _, go = l1_gain(tcol)
                b1 = (b0 * go + 64) * 2.0 ** sh[1] + 1
                _, go2 = l1_gain(trow)
                out = (b1 * go2 + 64) * 2.0 ** sh[2] + 1
                lts = (w * h > 256) + (w * h > 1024)
                worst[bd] = max(worst[bd], out * (1 << lts))
    return worst


if __name__ == "__main__":
    rows = analyse()
    worst = max(rows, key=lambda r: r[3])
    print("worst multiplier input: bd=%d tx_size=%d tx_type=%d |a| <= %.0f = 2^%.2f"
          % (worst[0], worst[1], worst[2], worst[3], np.log2(worst[3])))
    for bd in (8, 10, 12):
        m = max(r[3] for r in rows if r[0] == bd)
        print("bd %d: 2^%.2f" % (bd, np.log2(m)))
