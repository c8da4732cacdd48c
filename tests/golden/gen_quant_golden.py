#!/usr/bin/env python3
"""Generate tests/golden/quant_golden.npz and scan_sha256.json.

The reference holds no quantizer vectors ("parity unpinned" by
constants), so this pins the oracle two ways, both run in the build container
with /root/reference readable:

 1. SCAN TABLES.  Every literal table of src/scan_order.rs (42 tables) and the
    [TxSize][TxType] map av1_scan_orders are parsed from the reference text and
    compared with the generating RULE that oracle/quantize.c and the product
    use; the SHA-256 of each (tx_size, tx_type) scan is committed so the GPU
    box can re-check the rule without the reference.
 2. QUANTIZE / DEQUANTIZE.  An independent NumPy model written from
    src/quantize/mod.rs:219-384 -- true integer division instead of the
    reference's divu_pair reciprocal (which its own test claims is exact),
    Python-int arithmetic, explicit scan loops -- with the quantizer tables
    parsed from src/quantize/tables.rs.  Vectors: 19 sizes x {default, mrow,
    mcol} types x bd {8,10,12} x intra/inter x several qindex, with dense,
    sparse, near-threshold and extreme coefficients.
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
TX_NAMES = ["TX_4X4", "TX_8X8", "TX_16X16", "TX_32X32", "TX_64X64", "TX_4X8", "TX_8X4",
            "TX_8X16", "TX_16X8", "TX_16X32", "TX_32X16", "TX_32X64", "TX_64X32", "TX_4X16",
            "TX_16X4", "TX_8X32", "TX_32X8", "TX_16X64", "TX_64X16"]


def rule_scan(kind, W, H):
    if kind == "mcol":
        return list(range(W * H))
    if kind == "mrow":
        return [c * H + r for r in range(H) for c in range(W)]
    out = []
    for d in range(W + H - 1):
        cells = [(r, d - r) for r in range(H) if 0 <= d - r < W]
        if W > H or (W == H and d % 2 == 0):
            cells = cells[::-1]
        out += [c * H + r for r, c in cells]
    return out


def kind_of(tx_type):
    return "default" if tx_type < 10 else ("mcol" if tx_type & 1 else "mrow")


def check_scans():
    src = open(os.path.join(REF, "scan_order.rs")).read()
    tabs = {}
    for m in re.finditer(r"static (\w+_scan_\w+)\s*: \[u16; (\d+)\] = \[(.*?)\];", src, re.S):
        tabs[m.group(1)] = [int(x) for x in re.findall(r"\d+", m.group(3))]
    assert len(tabs) == 42
    for k, v in tabs.items():
        kind, _, dims = k.split("_")
        W, H = map(int, dims.split("x"))
        assert rule_scan(kind, W, H) == v, k
    mp = re.search(r"pub static av1_scan_orders.*?= \[(.*)\];", src, re.S).group(1)
    rows = dict((n, re.findall(r"scan: &(\w+),", b))
                for n, b in re.findall(r"\[\s*//\s*(TX_\w+)(.*?)\n\t\]", mp, re.S))
    sha = {}
    for ts, name in enumerate(TX_NAMES):
        ents = rows[name]
        assert len(ents) == 16
        W, H = min(TX_W[ts], 32), min(TX_H[ts], 32)
        for tt in range(16):
            mine = rule_scan(kind_of(tt), W, H)
            assert mine == tabs[ents[tt]], (name, tt)
            sha["%d_%d" % (ts, tt)] = hashlib.sha256(
                np.asarray(mine, dtype="<u2").tobytes()).hexdigest()
    json.dump(sha, open(os.path.join(HERE, "scan_sha256.json"), "w"), indent=0, sort_keys=True)
    print("scan rule == all 42 reference tables and the 19x16 map; wrote scan_sha256.json")


def load_qtables():
    src = open(os.path.join(REF, "quantize/tables.rs")).read()
    t = {}
    for m in re.finditer(r"const (\w+)_raw: \[u16; QINDEX_RANGE\] = \[(.*?)\];", src, re.S):
        t[m.group(1)] = [int(x) for x in re.findall(r"\d+", m.group(2))]
    dc = [t["dc_qlookup_Q3"], t["dc_qlookup_10_Q3"], t["dc_qlookup_12_Q3"]]
    ac = [t["ac_qlookup_Q3"], t["ac_qlookup_10_Q3"], t["ac_qlookup_12_Q3"]]
    return dc, ac


def wrap(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def model_quantize(co, ts, tt, qindex, bd, intra, dcd, acd, cbits, dc_tab, ac_tab):
    """co: python ints (>= coded area).  Returns (qcoeffs, eob, rcoeffs)."""
    W, H = min(TX_W[ts], 32), min(TX_H[ts], 32)
    n = W * H
    scan = rule_scan(kind_of(tt), W, H)
    iscan = [0] * n
    for i, p in enumerate(scan):
        iscan[p] = i
    area = TX_W[ts] * TX_H[ts]
    lts = int(area > 256) + int(area > 1024)
    bc = min((bd ^ 8) >> 1, 2)
    dcq = dc_tab[bc][min(max(qindex + dcd, 0), 255)]
    acq = ac_tab[bc][min(max(qindex + acd, 0), 255)]
    dc_off = dcq * (109 if intra else 108) // 256
    off0 = acq * (98 if intra else 97) // 256
    off1 = acq * (109 if intra else 108) // 256
    off_eob = acq * (88 if intra else 44) // 256
    q = [0] * n
    c0 = wrap(co[0] << lts, 32)
    v = (abs(c0) + dc_off) // dcq
    q[0] = wrap(-v if c0 < 0 else v, cbits)
    deadzone = wrap(((acq - off_eob) + (1 << lts) - 1) >> lts, cbits)
    eob_m1 = 0
    for i in range(n):
        a = wrap(abs(co[i]), cbits)           # T::abs wraps at T::MIN
        if a >= deadzone:
            eob_m1 = max(eob_m1, iscan[i])
    eob = eob_m1 + 1 if eob_m1 > 0 else int(q[0] != 0)
    mode = 1
    for i in range(1, eob):
        p = scan[i]
        c = wrap(co[p] << lts, 32)
        a = abs(c)
        l0 = a // acq
        off = off1 if l0 > 1 - mode else off0
        aq = l0 + int(a + off >= (l0 + 1) * acq)
        if mode != 0 and aq == 0:
            mode = 0
        elif aq > 1:
            mode = 1
        q[p] = wrap(-aq if c < 0 else aq, cbits)
    r = []
    for i in range(n):
        qq = dcq if i == 0 else acq
        c = q[i]
        r.append(wrap((wrap(c * qq, 32) + ((-1 if c < 0 else 0) & ((1 << lts) - 1))) >> lts, cbits))
    return q, eob, r


def main():
    check_scans()
    dc_tab, ac_tab = load_qtables()
    rng = np.random.default_rng(7)
    out = {}
    ncase = 0
    for ts in range(19):
        n = min(TX_W[ts], 32) * min(TX_H[ts], 32)
        full = TX_W[ts] * TX_H[ts]
        types = [0, 10, 11] if max(TX_W[ts], TX_H[ts]) <= 16 else [0]
        for tt in types:
            for bd in (8, 10, 12):
                cbits = 16 if bd == 8 else 32
                for intra in (0, 1):
                    for qindex in ((20, 100, 255) if n <= 256 else (100,)):
                        dcd, acd = ((0, 0), (-3, 5))[ncase % 2]
                        bc = min((bd ^ 8) >> 1, 2)
                        acq = ac_tab[bc][min(max(qindex + acd, 0), 255)]
                        blocks = []
                        amp = max(4, acq * 3)
                        blocks.append(rng.integers(-amp, amp + 1, full))                 # dense
                        b = rng.integers(-amp, amp + 1, full) * (rng.random(full) < 0.08)
                        blocks.append(b)                                                # sparse
                        b = rng.integers(-2, 3, full) + rng.choice([0, acq // 2, acq, -acq], full)
                        blocks.append(b)                                                # thresholds
                        lim = (1 << (cbits - 1)) - 1 if cbits == 16 else (1 << 24)
                        b = rng.integers(-lim - 1, lim + 1, full)
                        blocks.append(b)                                                # extreme
                        blocks.append(np.zeros(full, np.int64))                         # all zero
                        b = np.zeros(full, np.int64); b[0] = 1
                        blocks.append(b)                                                # tiny DC only
                        co = np.stack(blocks).astype(np.int64)
                        qs, es, rs = [], [], []
                        for row in co:
                            q, e, r = model_quantize([int(x) for x in row], ts, tt, qindex, bd,
                                                     intra, dcd, acd, cbits, dc_tab, ac_tab)
                            qs.append(q); es.append(e); rs.append(r)
                        k = "q_%d_%d_%d_%d_%d_%d_%d" % (ts, tt, bd, intra, qindex, dcd, acd)
                        dt = np.int16 if cbits == 16 else np.int32
                        out[k + "_co"] = co[:, :n].astype(dt)     # only the coded area is read
                        out[k + "_q"] = np.asarray(qs, dt)
                        out[k + "_eob"] = np.asarray(es, np.uint16)
                        out[k + "_r"] = np.asarray(rs, dt)
                        ncase += 1
    path = os.path.join(HERE, "quant_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d cases, %.1f KiB" % (path, ncase, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    sys.exit(main())
