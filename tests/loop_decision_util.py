"""Shared by tests/test_loop_decision_ref.py (CPU: the host driver on the oracle) and tests/test_gpu_loop_decision.py
(the same driver on the device): the `ldb*` cases of loop_decision_ref.npz -- rdo_loop_decision executed whole with
both filters on -- as planes and parameters, the recorded trace grouped by area, and the oracle as a backend of
rav1e_amd.loop_decision.LoopDecision."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O
from rav1e_amd import loop_decision as LD

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop_decision_ref.npz")
TRIAL_UNIT = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("set", "u1"), ("edges", "u1"),
                       ("xqd", "i1", (2,)), ("sb", "<i4")])       # r1o_trial_unit == R1TrialUnit


def load():
    return np.load(GOLD)


def both_cases(L):
    return sorted(k[:-6] for k in L.files if k.startswith("ldb") and k.endswith("_trace"))


def sigs(lib):
    vp, i = C.c_void_p, C.c_int
    lib.r1o_cdef_apply_area.restype, lib.r1o_cdef_apply_area.argtypes = i, [vp, vp, vp, i, i, i, vp, vp]
    lib.r1o_cdef_lrf_trial.restype = i
    lib.r1o_cdef_lrf_trial.argtypes = [vp, vp, vp, vp, i, i, i, vp, i, vp, vp, vp, vp, vp, vp, vp]
    return lib


def case(L, name):
    rec, src, skip, scales, prm, first_err, first_best = O.cdef_search_case(L, name)
    W, H, xdec, ydec, bd, damping, n_idx, asw, ash, planes = [int(v) for v in L[name + "_meta"]]
    rn, rs, rp = [int(v) for v in L[name + "_rate"]]
    q, full = [int(v) for v in L[name + "_q"]]

    def rate_fn(pli, f):       # the generator's STATED rate in place of the entropy coder's
        return rn if f is None else rs + rp * f[0]
    return dict(rec=rec, src=src, skip=skip, scales=scales, prm=prm, W=W, H=H, xdec=xdec, ydec=ydec, bd=bd,
                damping=damping, n_idx=n_idx, area=(asw, ash), lam=float(L[name + "_lambda"][0]), rate_fn=rate_fn, q=q,
                sets=LD.SGR_SETS["Full" if full else "Reduced"], first_err=first_err, first_best=first_best,
                ystr=[int(v) for v in L[name + "_ystr"]], uvstr=[int(v) for v in L[name + "_uvstr"]],
                dscale=[int(v) for v in L[name + "_dscale"]])


def trace_by_area(L, name):
    """{(ax0, ay0): [(pli, loop_sbx, loop_sby, sb_w, sb_h, frame, err)]}: the kind 0 rows, per area in call order; and
    the solves {(ax0, ay0): [(set, px, py, vis_w, vis_h, xqd0, xqd1)]}"""
    tr, te = L[name + "_trace"], L[name + "_trace_err"]
    errs, solves, cur = {}, {}, None
    for r, e in zip(tr.tolist(), te.tolist()):
        if r[0] == 3:
            cur = (r[1], r[2])
            errs[cur], solves[cur] = [], []
        elif r[0] == 0:
            errs[cur].append((r[1], r[2], r[3], r[4], r[5], r[6], int(e)))
        elif r[0] == 1:
            solves[cur].append((r[1], r[2], r[3], r[4], r[5], r[6] >> 8, ((r[6] & 255) ^ 128) - 128))
    return errs, solves


def driver(backend, c):
    return LD.LoopDecision(backend, c["W"], c["H"], c["xdec"], c["ydec"], c["q"], c["skip"], c["lam"], c["rate_fn"],
                           c["n_idx"], c["sets"])


def check_against_trace(L, name, ld):
    """every plane error of every pass, area by area in the reference's call order; the final picks"""
    best, lrf = ld.run()
    want, _ = trace_by_area(L, name)
    assert set(ld.events) <= set(want)
    n = 0
    for a, rows in want.items():
        got = ld.events.get(a, [])
        assert len(got) == len(rows), (name, a, len(got), len(rows))
        for k, (g, w) in enumerate(zip(got, rows)):
            assert g == w, (name, "area", a, "event", k, "got", g, "want", w)
        n += len(rows)
    # a completely skipped superblock is never searched: the fixture holds its blocks' untouched cdef_index there
    want_best = L[name + "_best_final"]
    assert np.array_equal(best[~ld.sb_skip], want_best[~ld.sb_skip]) and (best[ld.sb_skip] == -1).all(), (name, best, want_best)
    for (pli, x, y, s, x0, x1) in L[name + "_choice"].tolist():
        us = ld.cfgs[pli]["unit_size"]
        f = lrf.get((pli, x // us, y // us))
        assert (255, 0, 0) == (s, x0, x1) if f is None else f == (s, x0, x1), (name, pli, x, y, f, (s, x0, x1))
    return n


class OracleBackend:
    """trial / apply / lrf_search of rav1e_amd.loop_decision on the CPU oracle (oracle/loop_decision.c, lrf.c)"""

    def __init__(self, c):
        self.c, self.lib = c, sigs(O.lib())
        self.work = [O.plane_from_image(p.view().astype(np.int64), c["bd"], 16, 16) for p in c["rec"]]
        self.calls = {"trial": 0, "apply": 0, "lrf_search": 0}

    def _p3(self, planes):
        return (O.Plane * 3)(*[p.cstruct() for p in planes])

    def trial(self, units, sb_sel):
        c = self.c
        self.calls["trial"] += 1
        skip, scales = c["skip"], c["scales"]
        n_sby, n_sbx = sb_sel.shape
        allu = np.concatenate([np.ascontiguousarray(u, TRIAL_UNIT) for u in units])
        n_units = (C.c_int32 * 3)(*[len(u) for u in units])
        err = np.zeros((n_sby, n_sbx, 8), np.uint64)
        errp = np.zeros((n_sby, n_sbx, 8, 3), np.uint64)
        best = np.zeros((n_sby, n_sbx), np.int8)
        sel = np.ascontiguousarray(sb_sel, np.uint8)
        rc = self.lib.r1o_cdef_lrf_trial(self._p3(c["rec"]), self._p3(self.work), self._p3(c["src"]), skip.ctypes.data,
                                         skip.shape[1], skip.shape[1], skip.shape[0], scales.ctypes.data, scales.shape[1],
                                         C.byref(c["prm"]), allu.ctypes.data if len(allu) else None, n_units,
                                         sel.ctypes.data, err.ctypes.data, errp.ctypes.data, best.ctypes.data)
        assert rc == 0
        assert np.array_equal(err, errp.sum(axis=3))
        return errp

    def apply(self, index_sb):
        c = self.c
        self.calls["apply"] += 1
        skip = c["skip"]
        idx = np.ascontiguousarray(index_sb, np.int8)
        rc = self.lib.r1o_cdef_apply_area(self._p3(c["rec"]), self._p3(self.work), skip.ctypes.data, skip.shape[1],
                                          skip.shape[1], skip.shape[0], C.byref(c["prm"]), idx.ctypes.data)
        assert rc == 0

    def lrf_search(self, pli, rows):
        c = self.c
        self.calls["lrf_search"] += 1
        xd, yd = (0, 0) if pli == 0 else (c["xdec"], c["ydec"])
        scales = c["scales"]
        xqd = np.zeros((len(rows), 2), np.int8)
        err = np.zeros(len(rows), np.uint64)
        pin, ps = self.work[pli].cstruct(), c["src"][pli].cstruct()
        for k, r in enumerate(rows):
            rc = self.lib.r1o_lrf_search_unit(C.byref(pin), C.byref(ps), int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"]),
                                              int(r["set"]), int(r["edges"]), int(pli > 0), xd, yd, scales.ctypes.data,
                                              scales.shape[1], c["dscale"][pli], c["bd"], xqd[k].ctypes.data,
                                              err[k:].ctypes.data)
            assert rc == 0, (pli, r)
        return xqd, err


def lrf_only_case(L, name):
    """an `ldl*` case (restoration only: speed settings with cdef off) in the form `driver` takes"""
    W, H, xdec, ydec, bd, asw = [int(v) for v in L[name + "_meta"]]
    rec = [O.plane_from_image(L["%s_in%d" % (name, p)].astype(np.int64), bd, 16, 16) for p in range(3)]
    src = [O.plane_from_image(L["%s_src%d" % (name, p)].astype(np.int64), bd, 16, 16) for p in range(3)]
    rn, rs, rp = [int(v) for v in L[name + "_rate"]]
    q, full = [int(v) for v in L[name + "_q"]]
    prm = O.CdefSearchParams()
    prm.damping, prm.bit_depth, prm.n_idx, prm.planes = 3, bd, 1, 3
    prm.xdec, prm.ydec, prm.crop_w, prm.crop_h, prm.area_sb_w, prm.area_sb_h = xdec, ydec, W, H, asw, asw
    dscale = [int(v) for v in L[name + "_dscale"]]
    prm.dist_scale[:] = dscale
    gw, gh = (W + 7) // 8, (H + 7) // 8
    return dict(rec=rec, src=src, skip=np.zeros((2 * gh, 2 * gw), np.uint8), scales=np.ascontiguousarray(L[name + "_scales"]),
                prm=prm, W=W, H=H, xdec=xdec, ydec=ydec, bd=bd, damping=3, n_idx=1, area=(asw, asw),
                lam=float(L[name + "_lambda"][0]), rate_fn=lambda pli, f: rn if f is None else rs + rp * f[0], q=q,
                sets=LD.SGR_SETS["Full" if full else "Reduced"], ystr=[0] * 8, uvstr=[0] * 8, dscale=dscale)


def check_one_filter_cases(L, make_backend):
    """The driver with ONE filter enabled against the executed function: `ldc*` (restoration off: every trial's error,
    the pick per superblock) and `ldl*` (CDEF off: every option's error in call order, the choice per unit)."""
    n = 0
    for name in sorted(k[:-5] for k in L.files if k.startswith("ldc") and k.endswith("_meta")):
        c = case(L, name)
        ld = LD.LoopDecision(make_backend(c), c["W"], c["H"], c["xdec"], c["ydec"], c["q"], c["skip"], c["lam"], c["rate_fn"],
                             c["n_idx"], c["sets"], enable_restoration=False)
        best, lrf = ld.run()
        assert ld.passes == 1 and not lrf and ld.area == c["area"], (name, ld.passes, ld.area, c["area"])
        assert np.array_equal(best, L[name + "_best"]), (name, best, L[name + "_best"])
        err = np.zeros_like(L[name + "_err"])
        for (ax, ay), evs in ld.events.items():
            k = {}
            for (pli, lsx, lsy, _w, _h, fr, e) in evs:
                assert fr == 0
                sb = (ay + lsy, ax + lsx)
                i = k.get((sb, pli), 0)
                err[sb[0], sb[1], i] += np.uint64(e)
                k[(sb, pli)] = i + 1
        assert np.array_equal(err[..., :c["n_idx"]], L[name + "_err"][..., :c["n_idx"]]), name
        n += int((best >= 0).sum()) * c["n_idx"]
    for name in sorted(k[:-5] for k in L.files if k.startswith("ldl") and k.endswith("_meta")):
        c = lrf_only_case(L, name)
        ld = LD.LoopDecision(make_backend(c), c["W"], c["H"], c["xdec"], c["ydec"], c["q"], c["skip"], c["lam"], c["rate_fn"],
                             1, c["sets"], enable_cdef=False)
        best, lrf = ld.run()
        assert ld.passes == 2 and (best == -1).all() and ld.area[0] == c["area"][0], (name, ld.passes, ld.area)
        got = [e[6] for a in ld.areas() for e in ld.events.get(a, [])]
        assert got == [int(v) for v in L[name + "_err"]], (name, len(got), len(L[name + "_err"]))
        for (pli, x, y, s, x0, x1) in L[name + "_choice"].tolist():
            us = ld.cfgs[pli]["unit_size"]
            f = lrf.get((pli, x // us, y // us))
            assert ((255, 0, 0) if f is None else f) == (s, x0, x1), (name, pli, x, y, f, (s, x0, x1))
        n += len(got)
    return n


def synthetic_case(W, H, xdec, ydec, bd, q, seed, n_idx=4, p_skip=0.2, noise=6):
    """a frame the fixtures do not have, in the form `driver` / the backends take: smooth source + coding noise, random
    skip flags (one superblock completely skipped), random strengths, scale grid and plane scales, the stated rate"""
    from rav1e_amd import rdo_glue as RG
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    mx = (1 << bd) - 1
    base = ((np.sin(xx / 7.0) + np.cos((yy + 2 * xx) / 11.0)) * 45 + 128) * (1 << (bd - 8))
    src = [np.clip(base + rng.integers(-3, 4, (H, W)) * (1 << (bd - 8)), 0, mx).astype(np.int64)]
    cw, chh = W >> xdec, H >> ydec
    for k in (1, 2):
        src.append(np.clip(src[0][::1 << ydec, ::1 << xdec][:chh, :cw] // (k + 1) + (30 << (bd - 8)) * k, 0, mx))
    rec = [np.clip(p + rng.integers(-noise << (bd - 8), (noise << (bd - 8)) + 1, p.shape) * (rng.random(p.shape) < 0.6), 0, mx)
           for p in src]
    c = {"rec": [O.plane_from_image(p, bd, 16, 16) for p in rec], "src": [O.plane_from_image(p, bd, 16, 16) for p in src]}
    gw, gh = (W + 7) // 8, (H + 7) // 8
    skip = (rng.random((2 * gh, 2 * gw)) < p_skip).astype(np.uint8)
    if 2 * gw >= 32:
        skip[:16, 16:32] = 1
    prm = O.CdefSearchParams()
    ystr = [0, 9, 22, 63, 5, 40, 17, 50]
    uvstr = [0, 4, 13, 55, 2, 33, 21, 63]
    prm.y_strengths[:] = ystr
    prm.uv_strengths[:] = uvstr
    area = RG.restoration_area_sb(RG.restoration_plane_configs(W, H, xdec, ydec, q))
    damping = int(rng.integers(3, 7))
    prm.damping, prm.bit_depth, prm.n_idx, prm.planes = damping, bd, n_idx, 3
    prm.xdec, prm.ydec, prm.crop_w, prm.crop_h, prm.area_sb_w, prm.area_sb_h = xdec, ydec, W, H, area[0], area[1]
    dscale = [int(v) for v in rng.integers(1 << 13, 1 << 15, 3)]
    prm.dist_scale[:] = dscale
    scales = rng.integers(1 << 12, 1 << 16, (gh, gw)).astype(np.uint32)
    c.update(skip=skip, scales=scales, prm=prm, W=W, H=H, xdec=xdec, ydec=ydec, bd=bd, damping=damping, n_idx=n_idx, area=area,
             lam=90.0 * (1 << (2 * (bd - 8))), rate_fn=lambda pli, f: 24 if f is None else 96 + 8 * f[0], q=q,
             sets=LD.SGR_SETS["Reduced"], ystr=ystr, uvstr=uvstr, dscale=dscale)
    return c
