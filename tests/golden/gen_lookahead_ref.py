#!/usr/bin/env python3
"""tests/golden/lookahead_ref.npz: the lookahead cost maps and the block-importance propagation
computed by the REFERENCE'S OWN SOURCE TEXT, transpiled by tools/rustlite and executed here:

  estimate_intra_costs                   src/api/lookahead.rs:30-123   (the whole function)
  estimate_importance_block_difference   src/api/lookahead.rs:125-180  (the whole function)
  estimate_inter_costs, its cost loop    src/api/lookahead.rs:226-267  (see below)
  ContextInner::update_block_importances src/api/internal.rs:912-1068  (the whole function)
  and what they call: get_intra_edges (partition.rs), PredictionMode::predict_intra (predict.rs),
  get_satd (dist.rs).

estimate_inter_costs first builds two FrameInvariants and runs the whole motion search
(lookahead.rs:186-224: encoder state + compute_motion_vectors -- motion estimation has vectors of
its own, me_ref.npz).  What is pinned here is everything AFTER that: the lines from
"// Estimate inter costs" to the returned mean are taken verbatim from the file at run time and
wrapped in a function whose parameters are the values those lines read (frame, ref_frame,
bit_depth, the FrameMEStats of reference 0, fi).  The one line that takes `stats` out of the
RwLock is replaced by the parameter.  Motion vectors are an INPUT of the vectors.

Hand-stated (plain data, no reference text to run): Frame { planes }, the CodedFrameData fields
update_block_importances reads, FrameMEStats filled from an array.

Run in the build container:  python tests/golden/gen_lookahead_ref.py
"""
import os
import re

import numpy as np

import reflib as L
from reflib import R


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def inter_cost_tail_source():
    """the cost loop of estimate_inter_costs as a function of its own (text of lookahead.rs)"""
    src = open(os.path.join(L.REF_SRC, "api", "lookahead.rs")).read()
    a = src.index("  // Estimate inter costs")
    b = src.index("  inter_costs as f64 / (w_in_imp_b * h_in_imp_b) as f64", a)
    body = src[a:b] + "  inter_costs as f64 / (w_in_imp_b * h_in_imp_b) as f64\n"
    lock_line = re.search(r"\n\s*let stats = &fs\.frame_me_stats\.read\(\)\.expect\(\"poisoned lock\"\)\[0\];", body)
    assert lock_line, "lookahead.rs changed: the stats line is not where it was"
    body = body.replace(lock_line.group(0), "")
    return ("pub(crate) fn estimate_inter_costs_tail<T: Pixel>(\n"
            "  frame: &Frame<T>, ref_frame: &Frame<T>, bit_depth: usize, stats: &FrameMEStats,\n"
            "  fi: &FrameInvariants<T>,\n) -> f64 {\n" + body + "}\n")


def main():
    c = L.crate("api/lookahead.rs", "api/internal.rs", "dist.rs", "predict.rs", "partition.rs", "me.rs")
    c.load_text("<estimate_inter_costs, lines after compute_motion_vectors>", inter_cost_tail_source())
    intra = c.get("estimate_intra_costs")
    blockdiff = c.get("estimate_importance_block_difference")
    inter_tail = c.get("estimate_inter_costs_tail")
    ubi = c.get("update_block_importances", owner="ContextInner")
    MEStats, MV = L.struct(c, "MEStats"), L.struct(c, "MotionVector")
    FrameME = L.struct(c, "FrameMEStats")
    bsize8 = L.enum(c, "BlockSize", "BLOCK_8X8")
    rng = np.random.default_rng(20260925)
    out, keys = {}, []

    def frame_stats(mv, cols, rows):
        """FrameMEStats (4x4 units) whose entries at (2y, 2x) carry the importance blocks' vectors"""
        flat = [MEStats(mv=MV(row=0, col=0), normalized_sad=0) for _ in range(rows * cols)]
        hb, wb = mv.shape[:2]
        for y in range(hb):
            for x in range(wb):
                flat[(2 * y) * cols + 2 * x] = MEStats(mv=MV(row=int(mv[y, x, 0]), col=int(mv[y, x, 1])),
                                                       normalized_sad=0)
        return FrameME(stats=R.RSlice(flat), cols=cols, rows=rows)

    cases = [(8, 72, 40, "noise"), (8, 76, 44, "smooth"), (10, 64, 48, "noise"), (10, 52, 36, "smooth"),
             (12, 40, 40, "noise"), (8, 40, 24, "flat")]
    for (bd, w, h, kind) in cases:
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        if kind == "noise":
            org = rng.integers(0, mx + 1, (h, w))
            ref = rng.integers(0, mx + 1, (h, w))
        elif kind == "smooth":
            yy, xx = np.mgrid[0:h, 0:w]
            org = (mx * (0.5 + 0.4 * np.sin(xx / 9.0) * np.cos(yy / 7.0))).astype(np.int64) + rng.integers(-3, 4, (h, w))
            ref = np.roll(org, (1, -2), (0, 1)) + rng.integers(-2, 3, (h, w))
        else:
            org = np.full((h, w), mx // 3)
            ref = np.full((h, w), mx // 3 + 2)
        org, ref = np.clip(org, 0, mx).astype(dt), np.clip(ref, 0, mx).astype(dt)
        k = "%d_%d_%d_%s" % (bd, w, h, kind)
        keys.append(k)
        out["org_" + k], out["ref_" + k] = org, ref
        # planes with the encoder's luma padding so that vectors may point outside the frame; the
        # padding holds real (replicated-edge) pixels like a padded reference frame
        pad = 16
        ref_p = np.pad(ref, pad, mode="edge")
        out["refpad_" + k] = ref_p
        p_org = L.plane_from_array(org, bd, pad, pad)
        p_ref = L.plane_from_padded(ref_p, bd, pad, pad)
        tmp = L.plane_from_array(np.zeros_like(org), bd, pad, pad)
        f_org, f_ref = Obj(planes=R.RSlice([p_org])), Obj(planes=R.RSlice([p_ref]))
        hb, wb = h // 8, w // 8
        # ---- estimate_intra_costs
        ic = intra(g, tmp, f_org, bd, None)
        out["intra_" + k] = np.array(list(ic), np.uint32).reshape(hb, wb)
        # ---- estimate_importance_block_difference (f64 mean)
        out["blockdiff_" + k] = np.array([blockdiff(g, f_org, f_ref)], np.float64)
        # ---- the cost loop of estimate_inter_costs, vectors in 1/8 pel within the padding
        mv = rng.integers(-8 * (pad - 1), 8 * (pad - 1) + 1, (hb, wb, 2))
        mv[0, 0] = (0, 0)
        if hb > 1 and wb > 1:
            mv[1, 1] = (-7, 7)          # truncation toward zero of a negative position
            mv[hb - 1, wb - 1] = (8 * (pad - 1), 8 * (pad - 1))
        out["mv_" + k] = mv.astype(np.int16)
        cols, rows = (w + 3) // 4, (h + 3) // 4
        fi = Obj(cpu_feature_level=None)
        stats = frame_stats(mv, cols, rows)
        out["inter_mean_" + k] = np.array([inter_tail(g, f_org, f_ref, bd, stats, fi)], np.float64)
        # ---- update_block_importances: intra costs from above, random future importances, len
        fut = (rng.random((hb, wb)) * 3000.0).astype(np.float32)
        out["future_" + k] = fut
        for ln in (1, 4):
            imp = (rng.random((hb, wb)) * 10.0).astype(np.float32)
            out["refimp_in_%d_%s" % (ln, k)] = imp
            acc = R.RSlice([R.F32(float(v)) for v in imp.ravel()])
            coded = Obj(lookahead_intra_costs=R.RSlice([int(v) for v in out["intra_" + k].ravel()]),
                        block_importances=R.RSlice([R.F32(float(v)) for v in fut.ravel()]),
                        w_in_imp_b=wb, h_in_imp_b=hb)
            fi2 = Obj(coded_frame_data=R.Some(coded), cpu_feature_level=None)
            ubi(g, fi2, stats, f_org, f_ref, bd, bsize8, ln, acc)
            assert all(type(v) is R.F32 for v in acc), "an f32 operation escaped the F32 model"
            res = np.array([float(v) for v in acc], np.float32).reshape(hb, wb)
            out["refimp_out_%d_%s" % (ln, k)] = res
    out["keys"] = np.array(keys)

    # ---- adversarial set for update_block_importances (f32, order-of-operation code): reference
    # positions ON and either side of importance-block boundaries in both axes, negative positions
    # (-1, -63, -64, -65, -127, -128 in MV units: the `reference < 0` floor terms of
    # internal.rs:1010-1017), positions at / beyond the right and bottom frame edge (targets partly or
    # wholly off-frame on each side), len in {1, 2, 7}.  The intra costs are GIVEN (large) so that
    # every block propagates; the inter costs come from get_satd inside the executed function.
    adv_keys = []
    for ci, (bd, w, h, kind) in enumerate([(8, 48, 40, "noise"), (8, 56, 32, "smooth"), (10, 48, 40, "noise"),
                                           (10, 40, 48, "smooth"), (12, 48, 32, "noise"), (8, 64, 64, "smooth")]):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        if kind == "noise":
            org = rng.integers(0, mx + 1, (h, w))
            ref = rng.integers(0, mx + 1, (h, w))
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            org = (mx * (0.5 + 0.4 * np.sin(xx / 5.0) * np.cos(yy / 6.0))).astype(np.int64) + rng.integers(-3, 4, (h, w))
            ref = np.roll(org, (2, -1), (0, 1)) + rng.integers(-2, 3, (h, w))
        org, ref = np.clip(org, 0, mx).astype(dt), np.clip(ref, 0, mx).astype(dt)
        k = "%d_%d_%d_adv%d" % (bd, w, h, ci)
        adv_keys.append(k)
        pad = 16
        ref_p = np.pad(ref, pad, mode="edge")
        out["org_" + k], out["ref_" + k], out["refpad_" + k] = org, ref, ref_p
        f_org = Obj(planes=R.RSlice([L.plane_from_array(org, bd, pad, pad)]))
        f_ref = Obj(planes=R.RSlice([L.plane_from_padded(ref_p, bd, pad, pad)]))
        hb, wb = h // 8, w // 8
        mv = np.zeros((hb, wb, 2), np.int64)

        def targets(nb, size_px):
            edge = nb * 64                                       # the frame edge in MV units
            t = [-128, -127, -65, -64, -63, -1, 0, 1, 63, 64, 65, 127, 128, 129]
            t += [edge + d for d in (-129, -128, -127, -65, -64, -63, -1, 0, 1, 63, 64)]
            return [v for v in t if -8 * pad <= v <= (size_px + pad - 8) * 8]
        tx, ty = targets(wb, w), targets(hb, h)
        for y in range(hb):
            for x in range(wb):
                # a position is reachable from any block: the vector is what it takes to get there
                rx = tx[(3 * (y * wb + x) + ci) % len(tx)] if rng.random() < 0.85 else int(rng.integers(-8 * pad, (w + pad - 8) * 8 + 1))
                ry = ty[(5 * (y * wb + x) + 2 * ci + x) % len(ty)] if rng.random() < 0.85 else int(rng.integers(-8 * pad, (h + pad - 8) * 8 + 1))
                mv[y, x] = (ry - y * 64, rx - x * 64)
        assert np.abs(mv).max() < 32768
        out["mv_" + k] = mv.astype(np.int16)
        cols, rows = (w + 3) // 4, (h + 3) // 4
        stats = frame_stats(mv, cols, rows)
        # SATD of an 8x8 block is at most 64 * mx * 8 / 8; intra costs above that always propagate,
        # a few small ones (0, 1) take the `intra_cost <= inter_cost` arm
        intra_g = rng.integers(64 * mx, 200 * mx, (hb, wb)).astype(np.uint32)
        intra_g[rng.random((hb, wb)) < 0.1] = rng.integers(0, 2)
        out["intra_" + k] = intra_g
        fut = (rng.random((hb, wb)) * 3000.0).astype(np.float32)
        fut[rng.random((hb, wb)) < 0.2] = 0.0
        out["future_" + k] = fut
        for ln in (1, 2, 7):
            imp = (rng.random((hb, wb)) * 10.0).astype(np.float32)
            out["refimp_in_%d_%s" % (ln, k)] = imp
            acc = R.RSlice([R.F32(float(v)) for v in imp.ravel()])
            coded = Obj(lookahead_intra_costs=R.RSlice([int(v) for v in intra_g.ravel()]),
                        block_importances=R.RSlice([R.F32(float(v)) for v in fut.ravel()]),
                        w_in_imp_b=wb, h_in_imp_b=hb)
            fi2 = Obj(coded_frame_data=R.Some(coded), cpu_feature_level=None)
            ubi(g, fi2, stats, f_org, f_ref, bd, bsize8, ln, acc)
            assert all(type(v) is R.F32 for v in acc), "an f32 operation escaped the F32 model"
            out["refimp_out_%d_%s" % (ln, k)] = np.array([float(v) for v in acc], np.float32).reshape(hb, wb)
    out["adv_keys"] = np.array(adv_keys)
    L.save("lookahead_ref.npz", out)


if __name__ == "__main__":
    main()
