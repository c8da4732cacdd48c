#!/usr/bin/env python3
"""tests/golden/inv_tx_ref.npz: `inverse_transform_add` computed by the REFERENCE'S OWN SOURCE TEXT.

  transform::inverse::rust::inverse_transform_add   src/transform/inverse.rs:1633-1705
      the WHOLE function as written: the `&input[..min(w,32)*min(h,32)]` cut, the row pass with
      `step_by(min(h,32))` gathering, the rect INV_SQRT2 scaling, `raw >> 2` for WHT, clamp to
      bd + 8, the INV_TXFM_FNS table (:1593-1621), INV_INTERMEDIATE_SHIFTS (:1710-1711), the
      column pass clamp to max(bd + 6, 16), round_shift(.., 4), the add and the pixel clamp,
  and everything it calls, also executed from the text: av1_idct4 .. av1_iidentity32 / av1_iwht4
  (:35-1588), half_btf / clamp_value / round_shift / get_1d_tx_types (transform/mod.rs), TxSize's
  width()/height()/rect_ratio_log2(), PlaneRegionMut::rows_iter_mut (tiling/plane_region.rs).

tools/rustlite transpiles these items to Python at run time; nothing is restated here except the
driver loop that feeds cases in.  (gen_inv_tx_golden.py, round 1, executed only the 1-D networks
and hand-stated the 2-D driver; this file supersedes it for the 2-D cases.)

Coefficients are T::Coeff (i16 for u8 pixels, i32 for u16) as in the reference.  Cases per
(tx_size, valid tx_type, bit depth 8/10/12): random dense, clamp-exercising large amplitudes, DC
only, sparse, and "eob-limited" blocks (non-zero only in the first few scan positions -- what
the encoder's quantizer hands over).

Run in the build container:  python tests/golden/gen_inv_tx_ref.py
"""
import sys

import numpy as np

import reflib as L
from reflib import R

FILES = ["transform/inverse.rs", "transform/mod.rs", "tiling/plane_region.rs", "util/mod.rs"]
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


def main():
    c = L.crate(*FILES, strict=True)
    ita = c.get("inverse_transform_add")
    valid = c.get("valid_av1_transform")
    TxSize = [L.enum(c, "TxSize", v[0]) for v in c.enums["TxSize"].variants]
    TxType = [L.enum(c, "TxType", v[0]) for v in c.enums["TxType"].variants]
    assert len(TxSize) == 19 and len(TxType) == 17
    rng = np.random.default_rng(20260923)
    out = {}
    ncase = 0
    nthin = 0
    for ts in range(19):
        w, h = TX_W[ts], TX_H[ts]
        area = min(w, 32) * min(h, 32)
        for tt in range(17):
            if not valid({}, TxSize[ts], TxType[tt]):
                continue
            if tt == 16 and ts != 0:
                continue    # INV_TXFM_FNS holds `unimplemented!()` for WHT beyond 4 points (:1614-1620)
            for bd in (8, 10, 12):
                g = L.pixel_type(bd)
                dt = L.np_dtype(bd)
                ct = np.int16 if bd == 8 else np.int32
                nb = 5 if w * h <= 1024 else 3
                amp = 1 << (bd + 3)
                co = np.zeros((nb, area), np.int64)
                co[0] = rng.integers(-amp, amp, area)
                # reaches every clamp (bd + 8 bits); raw * INV_SQRT2 stays inside i32, where a debug
                # and a release build of the reference agree
                big = 32768 if bd == 8 else 700000
                co[1] = rng.integers(-big, big, area)
                co[2, 0] = rng.integers(-amp * 4, amp * 4)            # DC only
                if nb > 3:
                    co[3] = rng.integers(-amp, amp, area) * (rng.random(area) < 0.1)
                    k = int(rng.integers(1, 7))                       # eob-limited
                    co[4, :k] = rng.integers(-amp * 2, amp * 2, k)
                else:
                    k = int(rng.integers(1, 12))
                    co[2, 1:k] = rng.integers(-amp, amp, k - 1)
                co = co.astype(ct)
                pred = rng.integers(0, 1 << bd, (nb, h, w)).astype(dt)
                pred[1, : h // 2] = rng.choice([0, (1 << bd) - 1], (h // 2, w))   # pixel clamp
                rec = np.zeros_like(pred)
                for i in range(nb):
                    # The crate is transpiled with strict=True: every `let` is range-checked, so a
                    # block that runs to the end overflowed nowhere and a debug and a release build
                    # of the reference agree on it.  The clamp-exercising block can overflow i32
                    # inside av1_iadst4 / av1_iidentity* at 12 bits (the reference wraps there in
                    # release and panics in debug): thin it out until it does not.
                    for attempt in range(6):
                        # a plane larger than the block, block at (8, 4): the region's stride is real
                        img = np.zeros((h + 8, w + 16), dt)
                        img[4:4 + h, 8:8 + w] = pred[i]
                        p = L.plane_from_array(img, bd, 0, 0)
                        area_e = c.G["_E"]("Area", "StartingAt", 1, (8, 4))
                        reg = p.as_region().subregion(area_e)
                        eob = int(np.flatnonzero(co[i]).max() + 1) if co[i].any() else 0
                        try:
                            ita(g, R.RSlice([int(v) for v in co[i]]), reg, eob, TxSize[ts], TxType[tt], bd, None)
                            break
                        except R.Panic as e:
                            assert i == 1 and "does not fit i32" in str(e), (ts, tt, bd, i, e)
                            nthin += 1
                            base = rng.integers(-(1 << (bd + 5 - attempt)), 1 << (bd + 5 - attempt), area)
                            # (an identity row pass at 12 bits overflows for any input at the row
                            # clamp: 5793 << 19 > i32 -- the last attempts carry no spikes)
                            spikes = (rng.random(area) < (0.12 / (1 + attempt))) & (attempt < 3)
                            spikes[int(rng.integers(0, min(area, 4)))] = attempt < 3
                            co[i] = np.where(spikes, rng.choice([-big, big - 1], area), base).astype(ct)
                    else:
                        raise SystemExit("could not find an overflow-free block for %r" % ((ts, tt, bd),))
                    got = L.plane_to_array(p, dt)
                    rec[i] = got[4:4 + h, 8:8 + w]
                    got[4:4 + h, 8:8 + w] = 0
                    assert not got.any()        # nothing written outside the block
                k = "d2_%d_%d_%d" % (ts, tt, bd)
                out[k + "_co"] = co
                out[k + "_pred"] = pred
                out[k + "_rec"] = rec
                ncase += 1
        print("tx_size", ts, "done,", ncase, "cases", flush=True)
    L.save("inv_tx_ref.npz", out)
    print(ncase, "cases;", nthin, "re-draws of an overflowing block")


if __name__ == "__main__":
    sys.exit(main())
