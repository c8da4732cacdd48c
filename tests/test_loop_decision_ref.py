"""loop_decision_ref.npz: rdo_loop_decision (src/rdo.rs:2104-2763) EXECUTED WHOLE through tools/rustlite
(tests/golden/gen_loop_decision_ref.py).  The arithmetic of its two legs is checked where the other fixtures of
those legs are (tests/test_oracle_lrf.py, tests/test_oracle_cdef_search.py, and on the GPU tests/test_gpu_parity.py
::test_lrf_search_ref, tests/test_gpu_ref_vectors.py::test_cdef_search_ref).  Here: what the HOST has to get right
around r1_lrf_search_batch -- which units a frame has and in which order, their visible sizes, which of them see
pixels left of / above themselves (R1SgrSolveUnit.edges), the cost of an option and the choice -- rav1e_amd.rdo_glue
against what the executed function did."""
import os

import numpy as np
import pytest

from rav1e_amd import rdo_glue as RG

L = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop_decision_ref.npz"))
LRF_CASES = sorted(k[:-5] for k in L.files if k.startswith("ldl") and k.endswith("_meta"))
# base_q_idx of the cases (gen_loop_decision_ref.py, CASES): the unit sizes follow from it
Q = {"ldl0": 100, "ldl1": 180, "ldl2": 100, "ldl3": 100}


@pytest.mark.parametrize("case", LRF_CASES)
def test_units_edges_costs_and_choices_of_the_restoration_leg(case):
    W, H, xdec, ydec, bd, asw = [int(v) for v in L[case + "_meta"]]
    rows, errs, edges, costs = L[case + "_rows"], L[case + "_err"], L[case + "_edges"], L[case + "_cost"]
    cfgs = RG.restoration_plane_configs(W, H, xdec, ydec, Q[case])
    # the geometry is RestorationState::new's (also pinned by lrf_geometry_ref.npz)
    assert [[c[k] for k in ("unit_size", "sb_h_shift", "sb_v_shift", "stripe_height", "cols", "rows")] for c in cfgs] == \
        L[case + "_geo"].tolist()
    area = RG.restoration_area_sb(cfgs)
    assert area[0] == asw
    rate_none, rate_sgr, rate_per_set = [int(v) for v in L[case + "_rate"]]
    lam = float(L[case + "_lambda"][0])
    seen, choice = set(), {}
    for pli in range(3):
        dx, dy = (0, 0) if pli == 0 else (xdec, ydec)
        want_units = RG.restoration_search_units(cfgs[pli], W, H, dx, dy)
        r = rows[rows[:, 0] == pli]
        got_units = sorted({(int(x), int(y), int(w), int(h)) for (_, x, y, w, h, *_r) in r.tolist()})
        assert got_units == sorted(want_units), (case, pli)          # the same units, each once per option
    i = 0
    while i < len(rows):
        pli, x, y, w, h, set_, _, _ = [int(v) for v in rows[i]]
        assert set_ == 255                                           # every unit opens with the no-filter option
        j = i + 1
        while j < len(rows) and int(rows[j][5]) != 255:
            assert tuple(rows[j][:5]) == tuple(rows[i][:5])          # ... followed by its parameter sets, in order
            j += 1
        dx, dy = (0, 0) if pli == 0 else (xdec, ydec)
        assert all(int(e) == RG.restoration_unit_edges(x, y, dx, dy, area) for e in edges[i:j]), (case, pli, x, y)
        options = [(rate_none if int(s) == 255 else rate_sgr + rate_per_set * int(s), int(e))
                   for s, e in zip(rows[i:j, 5], errs[i:j])]
        for (rate, err), c in zip(options, costs[i:j]):
            assert RG.compute_rd_cost(lam, rate, err) == float(c), (case, pli, x, y, rate, err)
        pick, _ = RG.pick_restoration_filter(options, lam)
        choice[(pli, x, y)] = tuple(int(v) for v in (rows[i + pick][5], rows[i + pick][6], rows[i + pick][7]))
        assert (pli, x, y) not in seen
        seen.add((pli, x, y))
        i = j
    want = {(int(p), int(x), int(y)): (int(s), int(a), int(b)) for (p, x, y, s, a, b) in L[case + "_choice"].tolist()}
    assert choice == want, (case, [(k, choice[k], want[k]) for k in choice if choice[k] != want[k]][:4])
    assert len({v[0] for v in want.values()}) > 1                     # the choices are not all the same option


def test_edges_are_set_only_inside_multi_unit_areas():
    # 64 / 32-pixel units (the configuration of BASELINE config 4): one unit per plane and area -> nothing visible
    c4 = RG.restoration_plane_configs(3840, 2160, 1, 1, 100)
    a = RG.restoration_area_sb(c4)
    assert a == (1, 1)
    assert all(RG.restoration_unit_edges(x, y, 0, 0, a) == 0 for (x, y, _, _) in RG.restoration_search_units(c4[0], 3840, 2160))
    # ldl1: 128-pixel luma units inside 256-pixel areas -> the second luma unit of a row sees its left neighbour
    c = RG.restoration_plane_configs(192, 128, 1, 1, 180)
    a = RG.restoration_area_sb(c)
    assert a == (4, 4)
    assert [RG.restoration_unit_edges(x, y, 0, 0, a) for (x, y, _, _) in RG.restoration_search_units(c[0], 192, 128)] == [0, 1]
    assert set(L["ldl1_edges"].tolist()) == {0, 1}


def test_the_both_filters_trace_alternates_the_legs_as_the_integration_notes_say():
    """ldb0: CDEF leg (trials, then the final pass of the pick) -> restoration leg on the CDEF output -> because a
    restoration choice changed, the CDEF leg again, now with the chosen restoration filter applied inside every trial
    (plane errors taken on the restoration working copy, frame 1) -> until nothing changes"""
    tr = L["ldb0_trace"]
    n_idx = int(L["ldb0_meta"][6])
    starts = [i for i, r in enumerate(tr) if r[0] == 3]
    assert len(starts) == len(L["ldb0_areas"])
    for a, b in zip(starts, starts[1:] + [len(tr)]):
        ev = tr[a + 1:b]
        kinds = ev[:, 0].tolist()
        if 2 not in kinds:
            continue                                  # a skipped superblock: only the restoration leg ran
        first_solve = kinds.index(1) if 1 in kinds else len(kinds)
        cdef1 = ev[:first_solve]
        assert (cdef1[:, 0] == 2).sum() == n_idx + 1                       # n_idx trials + the final pass
        assert cdef1[cdef1[:, 0] == 2][:n_idx, 3].tolist() == list(range(n_idx))
        assert set(cdef1[cdef1[:, 0] == 0][:, 6].tolist()) <= {0}         # first pass: errors on the CDEF output itself
        later = ev[first_solve:]
        if (later[:, 0] == 2).any():                                        # a second CDEF pass happened
            k = int(np.argmax(later[:, 0] == 2))
            after = later[k:]
            assert (after[after[:, 0] == 0][:, 6] == 1).any()               # ... with errors on the restored copy


# ---- round 6: the whole iteration, both filters on, through the host driver (rav1e_amd/loop_decision.py) ----
import loop_decision_util as U   # noqa: E402

BOTH = U.both_cases(L)


@pytest.mark.parametrize("case", BOTH)
def test_both_filters_iteration_on_the_oracle_reproduces_every_recorded_error(case):
    """rdo_loop_decision executed whole (ldb*: 8-bit and 10-bit 4:2:0 with one superblock per area, and 128-pixel luma
    units -- an area of 3 x 2 superblocks whose trials read their neighbours' current CDEF output) against the host
    driver with the CPU oracle behind it: every rdo_loop_plane_error of every pass in the reference's call order
    (CDEF trials on the CDEF output and, from the second pass on, on the RESTORED superblock; the restoration leg's
    options), the final cdef_index per superblock and the final filter per unit.  Pins oracle/loop_decision.c and the
    driver's host logic (areas, unit <-> superblock maps, rates, costs, the alternation)."""
    c = U.case(L, case)
    be = U.OracleBackend(c)
    ld = U.driver(be, c)
    n = U.check_against_trace(L, case, ld)
    assert n > 100 and ld.passes >= 2
    # the later passes really ran with restoration choices in play
    assert any(ev[5] == 1 and ev[3] == 1 and ev[4] == 1 and ld.cfgs[ev[0]]["unit_size"] >= 32
               for evs in ld.events.values() for ev in evs)


def test_trial_without_units_is_the_first_pass_search():
    """r1o_cdef_lrf_trial with no restoration choice = r1o_cdef_strength_search (the fixture's first-pass errors)"""
    import ctypes as C
    import oracle_lib as O
    c = U.case(L, "ldb0")
    be = U.OracleBackend(c)
    be.apply(np.full(c["first_best"].shape, -1, np.int8))
    errp = be.trial([np.zeros(0, U.TRIAL_UNIT)] * 3, np.ones(c["first_best"].shape, np.uint8))
    assert np.array_equal(errp.sum(axis=3), c["first_err"])


def test_driver_with_one_filter_enabled_on_the_oracle():
    """speed settings with only CDEF (ldc0-2) or only restoration (ldl0-3): the same driver, enable_restoration / enable_cdef
    off, against the executed function -- every error in call order, every pick and choice"""
    assert U.check_one_filter_cases(L, lambda c: U.OracleBackend(c)) > 300
