#!/usr/bin/env python3
"""tests/golden/quant_ref.npz: quantize / dequantize vectors and the scan orders,
computed by the REFERENCE'S OWN SOURCE TEXT (src/quantize/mod.rs:30-385:
get_log_tx_scale, dc_q, ac_q, divu_gen, divu_pair, QuantizationContext::{update,
quantize}, rust::dequantize; src/quantize/tables.rs; src/scan_order.rs literal
tables, invert(), av1_scan_orders), transpiled by tools/rustlite and executed here.
Same key layout as quant_golden.npz (+ scan_<ts>_<tt> = the reference's
av1_scan_orders[ts][tt].scan).

Run in the build container:  python tests/golden/gen_quant_ref.py
"""
import numpy as np

import reflib as L
from reflib import R

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


def main():
    c = L.crate("quantize/mod.rs", "quantize/tables.rs", "scan_order.rs", "transform/mod.rs")
    upd = c.get("update", owner="QuantizationContext")
    quant = c.get("quantize", owner="QuantizationContext")
    dflt = c.get("default", owner="QuantizationContext")
    deq = c.get("dequantize")
    ac_q = c.get("ac_q")
    TxSize = [L.enum(c, "TxSize", v[0]) for v in c.enums["TxSize"].variants]
    TxType = [L.enum(c, "TxType", v[0]) for v in c.enums["TxType"].variants]
    assert [t.disc for t in TxSize] == list(range(19)) and [t.disc for t in TxType] == list(range(17))
    orders = c.const_value("av1_scan_orders")
    rng = np.random.default_rng(20260926)
    out = {}
    for ts in range(19):
        n = min(TX_W[ts], 32) * min(TX_H[ts], 32)
        for tt in range(16):
            so = orders[ts][tt]
            scan, iscan = so.scan.tolist(), so.iscan.tolist()
            assert len(scan) == n and sorted(scan) == list(range(n))
            assert all(iscan[p] == i for i, p in enumerate(scan))
            out["scan_%d_%d" % (ts, tt)] = np.array(scan, np.uint16)
    ncase = 0
    for ts in range(19):
        n = min(TX_W[ts], 32) * min(TX_H[ts], 32)
        full = TX_W[ts] * TX_H[ts]
        types = [0, 10, 11, 1, 9] if max(TX_W[ts], TX_H[ts]) <= 16 else [0]
        for tt in types:
            for bd in (8, 10, 12):
                cbits = 16 if bd == 8 else 32
                g = {"T": "i16" if bd == 8 else "i32"}
                for intra in (0, 1):
                    for qindex in ((20, 100, 255) if n <= 256 else (100,)):
                        dcd, acd = ((0, 0), (-3, 5), (7, -9))[ncase % 3]
                        acq = ac_q({}, qindex, acd, bd)
                        amp = max(4, acq * 3)
                        blocks = [rng.integers(-amp, amp + 1, full)]                              # dense
                        blocks.append(rng.integers(-amp, amp + 1, full) * (rng.random(full) < 0.08))  # sparse
                        blocks.append(rng.integers(-2, 3, full) + rng.choice([0, acq // 2, acq, -acq], full))
                        lim = (1 << 15) - 1 if cbits == 16 else (1 << 24)
                        blocks.append(rng.integers(-lim, lim + 1, full))                          # extreme
                        blocks.append(np.zeros(full, np.int64))                                   # all zero
                        b = np.zeros(full, np.int64)
                        b[0] = 1
                        blocks.append(b)                                                          # tiny DC only
                        co = np.stack(blocks).astype(np.int64)
                        qs, es, rs = [], [], []
                        for row in co:
                            qc = dflt({})
                            upd({}, qc, qindex, TxSize[ts], bool(intra), bd, dcd, acd)
                            coeffs = R.RSlice([int(v) for v in row])
                            qco = R.RSlice([0] * n)              # pre-zeroed, as encode_tx_block does
                            eob = quant(g, qc, coeffs, qco, TxSize[ts], TxType[tt])
                            rco = R.RSlice([0] * n)
                            deq(g, qindex, qco, eob, rco, TxSize[ts], bd, dcd, acd, None)
                            qs.append(qco.tolist())
                            es.append(eob)
                            rs.append(rco.tolist())
                        k = "q_%d_%d_%d_%d_%d_%d_%d" % (ts, tt, bd, intra, qindex, dcd, acd)
                        dt = np.int16 if cbits == 16 else np.int32
                        out[k + "_co"] = co[:, :n].astype(dt)     # only the coded area is read
                        out[k + "_q"] = np.asarray(qs, dt)
                        out[k + "_eob"] = np.asarray(es, np.uint16)
                        out[k + "_r"] = np.asarray(rs, dt)
                        ncase += 1
        print("tx size", ts, ncase, flush=True)
    L.save("quant_ref.npz", out)
    print(ncase, "cases")


if __name__ == "__main__":
    main()
