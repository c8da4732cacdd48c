"""rdo_loop_decision with BOTH filters on (src/rdo.rs:2366-2574; BASELINE configs[3] runs speed 4: cdef and lrf on)
on the device, against the function EXECUTED WHOLE (loop_decision_ref.npz `ldb*`, tests/golden/gen_loop_decision_ref.py):
the host driver rav1e_amd/loop_decision.py over r1_cdef_lrf_trial_batch / r1_cdef_apply_area / r1_lrf_search_batch.
Every rdo_loop_plane_error of every pass is compared in the reference's call order -- the first pass' trials on the
CDEF output, the restoration leg's options on the CDEF working copy, and from the second pass on every trial's error
on the superblock RESTORED with the unit's current choice -- then the final cdef_index per superblock and the final
filter per restoration unit.  The same driver runs on the CPU oracle in tests/test_loop_decision_ref.py."""
import numpy as np
import pytest

import loop_decision_util as U
import oracle_lib as O
from rav1e_amd import loop_decision as LD

pytestmark = pytest.mark.gpu
L = U.load()


def dev_plane(hp):
    from rav1e_amd.api import Plane
    return Plane.from_numpy(hp.data, hp.width, hp.height, hp.bit_depth, hp.xpad, hp.ypad)


def device_backend(ctx, c):
    import torch
    rec = [dev_plane(p) for p in c["rec"]]
    work = [dev_plane(p) for p in c["rec"]]          # starts as the reconstruction: rec_subset.clone() (rdo.rs:2330-2331)
    src = [dev_plane(p) for p in c["src"]]
    return LD.DeviceBackend(ctx, rec, work, src, torch.from_numpy(c["skip"]).cuda(), c["ystr"], c["uvstr"], c["damping"],
                            c["bd"], c["n_idx"], c["xdec"], c["ydec"], c["W"], c["H"], c["area"],
                            torch.from_numpy(c["scales"].view(np.int32)).cuda(), c["dscale"])


@pytest.mark.parametrize("case", U.both_cases(L))
def test_both_filters_iteration_on_the_device(ctx, case):
    c = U.case(L, case)
    ld = U.driver(device_backend(ctx, c), c)
    n = U.check_against_trace(L, case, ld)
    assert n > 100 and ld.passes >= 2
    restored = [ev for evs in ld.events.values() for ev in evs if ev[5] == 1 and (ev[3], ev[4]) == (1, 1)]
    assert len(restored) >= 8          # trials measured on the restored superblock did happen


@pytest.mark.parametrize("case", U.both_cases(L))
def test_trial_and_apply_equal_the_oracle_with_random_choices(ctx, case):
    """Independent of the iteration: random restoration choices over random superblocks (every set kind: both radii,
    radius 2 only, radius 1 only; extreme weights), random index maps -- r1_cdef_lrf_trial_batch's per-plane errors and
    r1_cdef_apply_area's working copy against oracle/loop_decision.c."""
    import torch
    c = U.case(L, case)
    be, ob = device_backend(ctx, c), U.OracleBackend(c)
    drv = U.driver(be, c)
    rng = np.random.default_rng([6, sum(case.encode())])
    for rnd in range(3):
        idx = rng.integers(-1, c["n_idx"], drv.best_index.shape).astype(np.int8)
        be.apply(idx)
        ob.apply(idx)
        torch.cuda.synchronize()
        for pl in range(3):
            got = be.work[pl].data.cpu().numpy()
            got = got if got.dtype == np.uint8 else got.view(np.uint16)
            hp = ob.work[pl]
            xd, yd = drv.dec[pl]
            gw, gh = (drv.mi_cols * 4) >> xd, (drv.mi_rows * 4) >> yd      # whole 8x8 blocks: what the call writes
            want = hp.data[hp.yorigin:hp.yorigin + gh, hp.xorigin:hp.xorigin + gw]
            g = got[hp.yorigin:hp.yorigin + gh, hp.xorigin:hp.xorigin + gw]
            assert np.array_equal(g, want), (case, rnd, pl, np.argwhere(g != want)[:4])
        units = [[], [], []]
        aw, ah = drv.area
        for sy in range(drv.n_sby):
            for sx in range(drv.n_sbx):
                ax, ay = sx // aw * aw, sy // ah * ah
                for pl in range(3):
                    if drv.unit_offset(pl, ax, ay, sx, sy) is None or rng.random() < 0.3:
                        continue
                    x, y, w, h = drv.sb_vis(pl, ax, ay, sx, sy)
                    edges = (1 if sx > ax else 0) | (2 if sy > ay else 0)
                    s = int(rng.integers(0, 16))
                    xq = (int(rng.integers(-96, 32)), int(rng.integers(-32, 96)))
                    units[pl].append((x, y, w, h, s, edges, xq, sy * drv.n_sbx + sx))
        units = [np.array(u, U.TRIAL_UNIT) for u in units]
        sel = (rng.random(drv.best_index.shape) < 0.8).astype(np.uint8)
        got = be.trial(units, sel)
        want = ob.trial(units, sel)
        assert np.array_equal(got, want), (case, rnd, np.argwhere(got != want)[:6])
        assert sum(len(u) for u in units) > 0


@pytest.mark.parametrize("fmt", [(0, 0, 8, 100), (1, 0, 10, 100), (1, 1, 12, 100), (1, 1, 8, 180), (1, 1, 10, 220)])
def test_whole_iteration_device_equals_oracle_on_other_formats(ctx, fmt):
    """Formats and quantizers the executed fixtures do not have (4:4:4, 4:2:2, 12-bit; qindex 180 / 220: restoration
    units of 128 / 256 luma pixels, areas of 2 x 2 / 4 x 4 superblocks, units stretched over the frame's remainder):
    the whole iteration through the SAME host driver on the device and on the oracle -- every event, pick and choice
    equal.  The oracle side is pinned by the fixtures (tests/test_loop_decision_ref.py); this widens the device's
    coverage to geometries only the oracle reaches."""
    xdec, ydec, bd, q = fmt
    c = U.synthetic_case(264, 200, xdec, ydec, bd, q, [9, xdec, ydec, bd, q])
    area = c["area"]
    dev = U.driver(device_backend(ctx, c), c)
    ora = U.driver(U.OracleBackend(c), c)
    bd_, ld_ = dev.run()
    bo_, lo_ = ora.run()
    assert dev.area == area and dev.passes == ora.passes >= 2
    assert np.array_equal(bd_, bo_) and ld_ == lo_, (fmt, bd_, bo_)
    assert set(dev.events) == set(ora.events)
    n = 0
    for a in ora.events:
        assert dev.events[a] == ora.events[a], (fmt, a, [(g, w) for g, w in zip(dev.events[a], ora.events[a]) if g != w][:3])
        n += len(ora.events[a])
    assert n > 300 and any(v is not None for v in lo_.values())


def test_driver_with_one_filter_enabled_on_the_device(ctx):
    """ldc* (CDEF only) and ldl* (restoration only) through the driver on the device: every recorded error, pick, choice"""
    assert U.check_one_filter_cases(L, lambda c: device_backend(ctx, c)) > 300
