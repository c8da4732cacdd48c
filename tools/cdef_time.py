#!/usr/bin/env python3
"""Times the frame CDEF path on the GPU: r1_cdef_analyze_frame, r1_cdef_filter_frame_plane_dirs
(luma, 4:2:0 chroma) and the composite r1_cdef_filter_frame_plane, plus the strength search.
usage: python tools/cdef_time.py [--reps 20]  -> one JSON line per case"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from rav1e_amd import workload as W
    from rav1e_amd.api import Context, Plane
    ctx = Context()

    def ev_time(f, reps):
        f(); torch.cuda.synchronize()
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in e:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in e)
        return t[len(t) // 2]

    for (w, h, bd) in ((3840, 2160, 8), (3840, 2160, 10), (1920, 1080, 8)):
        dt = np.uint16 if bd > 8 else np.uint8
        luma = Plane.from_numpy(W.random_plane_array(w, h, bd, 1), w, h, bd, 88, 88)
        out = Plane.from_numpy(W.random_plane_array(w, h, bd, 2), w, h, bd, 88, 88)
        cw, ch = w // 2, h // 2
        cin = Plane.from_numpy(W.random_plane_array(cw, ch, bd, 3, 44, 44), cw, ch, bd, 44, 44)
        cout = Plane.from_numpy(W.random_plane_array(cw, ch, bd, 4, 44, 44), cw, ch, bd, 44, 44)
        mi_c, mi_r = (w + 3) // 4, (h + 3) // 4
        rng = np.random.default_rng(7)
        skip = torch.from_numpy((rng.random((mi_r, mi_c)) < 0.1).astype(np.uint8)).cuda()
        ci = torch.from_numpy(rng.integers(0, 8, ((h + 63) // 64, (w + 63) // 64)).astype(np.uint8)).cuda()
        ys, uvs = [38, 21, 9, 0, 63, 17, 4, 50], [20, 5, 0, 33, 62, 12, 45, 1]
        res = {"case": "%dx%d %d-bit" % (w, h, bd)}
        d, v = ctx.cdef_analyze_frame(luma, w, h, mi_c, mi_r)
        res["analyze_ms"] = ev_time(lambda: ctx.cdef_analyze_frame(luma, w, h, mi_c, mi_r), args.reps)
        res["luma_filter_ms"] = ev_time(lambda: ctx.cdef_filter_frame_plane_dirs(d, v, luma, out, 0, 0, 0, w, h, skip, ci, ys, uvs, 5, bd), args.reps)
        res["chroma420_filter_ms"] = ev_time(lambda: ctx.cdef_filter_frame_plane_dirs(d, v, cin, cout, 1, 1, 1, w, h, skip, ci, ys, uvs, 5, bd), args.reps)
        res["luma_composite_ms"] = ev_time(lambda: ctx.cdef_filter_frame_plane(luma, luma, out, 0, 0, 0, w, h, skip, ci, ys, uvs, 5, bd), args.reps)
        cin2 = Plane.from_numpy(W.random_plane_array(cw, ch, bd, 5, 44, 44), cw, ch, bd, 44, 44)
        cout2 = Plane.from_numpy(W.random_plane_array(cw, ch, bd, 6, 44, 44), cw, ch, bd, 44, 44)
        skip_s = torch.zeros((2 * ((h + 7) // 8), 2 * ((w + 7) // 8)), dtype=torch.uint8, device="cuda")
        presets = [0, 4, 9, 13, 22, 31, 43, 55]
        scales = torch.from_numpy(rng.integers(1 << 12, 1 << 16, ((h + 7) // 8, (w + 7) // 8)).astype(np.int32)).cuda()
        res["strength_search_8_presets_420_ms"] = ev_time(lambda: ctx.cdef_strength_search(
            [luma, cin, cin2], [out, cout, cout2], skip_s, presets, presets, 5, bd, 8, 1, 1, w, h, scales=scales), args.reps)
        bpp = 2 if bd > 8 else 1
        res["luma_filter_GBps"] = round(2 * w * h * bpp / (res["luma_filter_ms"] * 1e-3) / 1e9, 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
