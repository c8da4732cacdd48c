#!/usr/bin/env python3
"""tests/golden/lookahead_chain_ref.npz: the lookahead's inter-cost pipeline END TO END on the
reference's own outputs -- motion search feeding the cost loop feeding the importance propagation
(src/api/lookahead.rs:186-267: compute_motion_vectors, then the SATD loop over the importance blocks
at the searched vectors; src/api/internal.rs:912-1068: update_block_importances).

The motion search of these frames IS already an executed-reference result: me_ref.npz holds the
FrameMEStats that the reference's estimate_tile_motion (src/me.rs:153-218 and everything below it, run
through tools/rustlite by gen_me_ref.py) wrote for whole-frame tiles.  This generator takes those
statistics AS THE REFERENCE PRODUCED THEM, and executes on them
  * the cost loop of estimate_inter_costs (the lines after compute_motion_vectors, cut out of
    lookahead.rs at run time exactly as gen_lookahead_ref.py does) -> the mean inter cost,
  * estimate_intra_costs (lookahead.rs:30-123) on the source frame,
  * update_block_importances with those intra costs, the searched vectors and a random future
    importance map (len 1 and 3).
A GPU run that searches the same frames itself and feeds ITS statistics into ITS cost kernels must
land on the same numbers: lookahead_ref.npz pins the cost loop with vectors as an input, me_ref.npz pins
the search; this file pins their composition (a vector format or sampling-position mismatch between the
two stages would pass both and fail here).

What is hand-stated: the MEStats rows are read from me_ref.npz ((row, col, normalized_sad) per 4x4
unit); Frame { planes } / CodedFrameData as in gen_lookahead_ref.py.

Run in the build container:  python tests/golden/gen_lookahead_chain_ref.py
"""
import os

import numpy as np

import reflib as L
from reflib import R
from gen_lookahead_ref import Obj, inter_cost_tail_source

CASES = ["p0", "p3", "p4", "p5", "p6", "t0", "t1", "t5"]     # whole-frame tiles of me_ref.npz, 8 and 10 bit


def main():
    M = np.load(os.path.join(L.HERE, "me_ref.npz"))
    c = L.crate("api/lookahead.rs", "api/internal.rs", "dist.rs", "predict.rs", "partition.rs", "me.rs")
    c.load_text("<estimate_inter_costs, lines after compute_motion_vectors>", inter_cost_tail_source())
    intra = c.get("estimate_intra_costs")
    inter_tail = c.get("estimate_inter_costs_tail")
    ubi = c.get("update_block_importances", owner="ContextInner")
    MEStats, MV, FrameME = L.struct(c, "MEStats"), L.struct(c, "MotionVector"), L.struct(c, "FrameMEStats")
    bsize8 = L.enum(c, "BlockSize", "BLOCK_8X8")
    rng = np.random.default_rng(20260929)
    out, keys = {}, []
    for name in CASES:
        w, h, bd, tx, ty, tw, th = [int(v) for v in M[name + "_meta"][:7]]
        assert (tx, ty, tw, th) == (0, 0, w, h), "whole-frame tiles only"
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        org, ref = M[name + "_org0"].astype(dt), M[name + "_ref0_0"].astype(dt)
        st = M[name + "_stats0"]                                  # rows x cols x (row, col, normalized_sad)
        rows, cols = st.shape[:2]
        flat = [MEStats(mv=MV(row=int(st[y, x, 0]), col=int(st[y, x, 1])), normalized_sad=int(st[y, x, 2]))
                for y in range(rows) for x in range(cols)]
        stats = FrameME(stats=R.RSlice(flat), cols=cols, rows=rows)
        pad = 88
        p_org = L.plane_from_padded(np.pad(org, pad, mode="edge"), bd, pad, pad)
        p_ref = L.plane_from_padded(np.pad(ref, pad, mode="edge"), bd, pad, pad)
        f_org, f_ref = Obj(planes=R.RSlice([p_org])), Obj(planes=R.RSlice([p_ref]))
        hb, wb = h // 8, w // 8
        fi = Obj(cpu_feature_level=None)
        mean = inter_tail(g, f_org, f_ref, bd, stats, fi)
        tmp = L.plane_from_array(np.zeros_like(org), bd, pad, pad)
        ic = np.array(list(intra(g, tmp, f_org, bd, None)), np.uint32).reshape(hb, wb)
        fut = (rng.random((hb, wb)) * 2000.0).astype(np.float32)
        out["inter_mean_" + name] = np.array([mean], np.float64)
        out["intra_" + name], out["future_" + name] = ic, fut
        for ln in (1, 3):
            imp = (rng.random((hb, wb)) * 10.0).astype(np.float32)
            acc = R.RSlice([R.F32(float(v)) for v in imp.ravel()])
            coded = Obj(lookahead_intra_costs=R.RSlice([int(v) for v in ic.ravel()]),
                        block_importances=R.RSlice([R.F32(float(v)) for v in fut.ravel()]),
                        w_in_imp_b=wb, h_in_imp_b=hb)
            ubi(g, Obj(coded_frame_data=R.Some(coded), cpu_feature_level=None), stats, f_org, f_ref, bd, bsize8, ln, acc)
            assert all(type(v) is R.F32 for v in acc)
            out["imp_in_%d_%s" % (ln, name)] = imp
            out["imp_out_%d_%s" % (ln, name)] = np.array([float(v) for v in acc], np.float32).reshape(hb, wb)
        keys.append(name)
        print(name, w, h, bd, "mean inter cost", mean, flush=True)
    out["keys"] = np.array(keys)
    L.save("lookahead_chain_ref.npz", out)


if __name__ == "__main__":
    main()
