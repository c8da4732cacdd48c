"""The fused candidate kernel multiplies with v_mad_i32_i24 (rav1e_amd/csrc/tx_common.hpp,
namespace m24): exact only while every multiplier input of the forward-transform networks
stays inside 24 signed bits.  tools/tx_range.py bounds them by the L1 norm of the impulse
responses for residuals that come from pixels, every size / type / bit depth."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_multiplier_inputs_fit_24_bits_for_pixel_residuals():
    import tx_range
    rows = tx_range.analyse()
    assert len(rows) > 400                       # 19 sizes x valid types x 3 bit depths
    worst = max(r[3] for r in rows)
    assert worst < 2 ** 22, np.log2(worst)       # one bit of head-room below the 2^23 limit
    # the multipliers themselves are 24-bit too (largest constant of the networks: 62241)
    import re
    src = open(os.path.join(ROOT, "rav1e_amd", "csrc", "fwd_tx_1d.inc")).read()
    consts = [int(m) for m in re.findall(r"TX_MUL\(\w+, (-?\d+), \d+\)", src)]
    assert consts and 0 < min(consts) and max(consts) < 2 ** 23


def test_transposed_tile_of_blocks_up_to_16x16_fits_int16():
    """The type search keeps the column pass's output (after shift[1]) in an int16 LDS tile shared by the types of a
    column kernel (csrc/rdo_cand.hip, COLSHARE): exact because, for residuals that come from pixels and both sides
    <= 16, that value is bounded by 8193 / 16433 / 16445 at 8 / 10 / 12 bits, every valid type."""
    import tx_range as R
    F = R.F
    worst = {8: 0.0, 10: 0.0, 12: 0.0}
    for bd in worst:
        for ts, (w, h) in enumerate(F.TX_DIMS):
            if max(w, h) > 16:
                continue
            for tt in range(16):
                if not F.valid_av1_transform(ts, tt):
                    continue
                sh = F.FWD_SHIFT[ts][(bd - 8) // 2]
                tcol = F.TXFM_TYPE_LS[h.bit_length() - 3][F.VTX_TAB[tt]]
                b0 = ((1 << bd) - 1) * (1 << sh[0])
                _, go = R.l1_gain(tcol)
                worst[bd] = max(worst[bd], (b0 * go + 64) * 2.0 ** sh[1] + 1)
    assert worst[8] <= 8193 and worst[10] <= 16433 and worst[12] <= 16445 and max(worst.values()) < 32767, worst


def test_shifted_coefficients_of_pixel_residuals_leave_room_for_the_24_bit_quantizer():
    """csrc/quant_common.hpp (MID): the fused kernels divide a = |c << log_tx_scale| + ac_offset by ac_q with
    m = floor(2^s / q) + 1, s = 22 + ceil(log2 q), on v_mul_u32_u24 / v_mul_hi_u32_u24 -- exact for a < 2^22, operands
    below 2^24.  (1) a < 2^22 for every size / type / bit depth when the residual comes from pixels; (2) the magic is
    exact: floor(a * m >> s) is monotone in a, so agreeing with a // q at every k q - 1 and k q below 2^22 is agreeing
    everywhere; every AC quantizer of the 8 / 10 / 12-bit tables."""
    import re
    import tx_range as R
    w = R.shifted_coefficient_bound()
    assert w[8] < 2 ** 17.01 and w[10] < 2 ** 19.01 and w[12] < 2 ** 21.01, {k: np.log2(v) for k, v in w.items()}
    src = open(os.path.join(ROOT, "rav1e_amd", "csrc", "quant_tables.inc")).read()
    tabs = re.findall(r"kR1AcQLookup\[3\]\[256\]\s*=\s*\{(.*?)\};", src, re.S)
    assert tabs
    qs = sorted({int(v) for v in re.findall(r"\d+", tabs[0])})
    assert len(qs) > 300 and qs[0] == 4 and qs[-1] == 29247
    for q in qs:
        off = q * 109 // 256                      # the largest ac_offset (intra, ac_offset1)
        assert w[12] + off < 2 ** 22
        L = (q - 1).bit_length()
        s = 22 + L
        m = (1 << s) // q + 1
        assert m < 1 << 24
        k = np.arange(1, (1 << 22) // q + 1, dtype=np.uint64)
        for a in (k * np.uint64(q) - np.uint64(1), k * np.uint64(q)):
            a = a[a < (1 << 22)]
            assert np.array_equal((a * np.uint64(m)) >> np.uint64(s), a // np.uint64(q)), q
