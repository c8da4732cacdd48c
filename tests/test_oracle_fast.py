"""bench.py's vectorised CPU-baseline leg (oracle/fast_cand.c) against the scalar restatement
(oracle/batch.c r1o_rdo_cand_batch): same planes, same candidates, every value equal."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from rav1e_amd import workload as W


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("size", [8, 16, 32, 64])
def test_fast_leg_equals_scalar_oracle(size, bd):
    L = O.lib()
    fw, fh = 256, 192
    a, b = O.HostPlane(fw, fh, bd), O.HostPlane(fw, fh, bd)
    a.data = W.random_plane_array(fw, fh, bd, 11 + size)
    b.data = W.random_plane_array(fw, fh, bd, 12 + bd)
    if size == 64:   # extreme residuals: flat black against flat white rows
        a.data[88:88 + 64] = 0
        b.data[88 - 40:88 + 110] = (1 << bd) - 1
    c = W.speed6_ladder(fw, fh, 6, seed=size + bd)[size]
    # the four put_8tap cases and every filter pair appear
    c["col_frac"][::5] = 0
    c["row_frac"][::7] = 0
    c["mode_x"] = np.arange(len(c)) % 3
    c["mode_y"] = (np.arange(len(c)) // 3) % 3
    n = len(c)
    ct = np.int16 if bd == 8 else np.int32
    ts = {64: 4, 32: 3, 16: 2, 8: 1}[size]
    pa, pb = a.cstruct(), b.cstruct()
    ref = [np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros((n, size * size), ct)]
    got = [np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros((n, size * size), ct)]
    assert L.r1o_rdo_cand_batch(C.byref(pa), C.byref(pb), size, size, ts, O.ptr(c), n,
                                O.ptr(ref[0]), O.ptr(ref[1]), O.ptr(ref[2]), None) == 0
    legs = [L.r1o_fast_rdo_cand_batch]
    if L.r1o_fast512_available():      # the AVX-512 leg (oracle/fast_cand512.c) where the host has it
        legs.append(L.r1o_fast512_rdo_cand_batch)
    for leg in legs:
        for threads in (1, 3):
            for g in got:
                g[...] = 0
            assert leg(C.byref(pa), C.byref(pb), size, ts, O.ptr(c), n, threads,
                       O.ptr(got[0]), O.ptr(got[1]), O.ptr(got[2])) == 0
            for r, g, name in zip(ref, got, ("sad", "satd", "coeffs")):
                assert np.array_equal(r, g), (name, leg.__name__ if hasattr(leg, "__name__") else leg)


def test_avx512_leg_ran_here_or_says_so():
    """this container's CPU has AVX-512: the wide leg must actually be exercised by the test above"""
    L = O.lib()
    flags = open("/proc/cpuinfo").read()
    assert bool(L.r1o_fast512_available()) == all(f in flags for f in ("avx512f", "avx512bw", "avx512dq", "avx512vl"))


def test_fast_leg_rejects_what_it_does_not_cover():
    L = O.lib()
    a = O.HostPlane(64, 64, 8)
    a.data = W.random_plane_array(64, 64, 8, 1)
    pa = a.cstruct()
    c = W.speed6_ladder(64, 64, 1)[16]
    out = np.zeros(len(c), np.uint32)
    c["tx_type"] = 1
    assert L.r1o_fast_rdo_cand_batch(C.byref(pa), C.byref(pa), 16, 2, O.ptr(c), len(c), 1,
                                     O.ptr(out), O.ptr(out), None) == -1
    c["tx_type"] = 0
    assert L.r1o_fast_rdo_cand_batch(C.byref(pa), C.byref(pa), 4, 0, O.ptr(c), len(c), 1,
                                     O.ptr(out), O.ptr(out), None) == -1
