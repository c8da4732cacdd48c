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
