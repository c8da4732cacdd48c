import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from rav1e_amd.api import Context, Plane, CDEF_DIR_CAND
ctx = Context()
G = dict(np.load(os.path.join(ROOT, "tests/golden/cdef_ref.npz")))
def plane_from(arr, bd, pad=16):
    h, w = arr.shape
    hp = O.HostPlane(w, h, bd, pad, pad, rng=np.random.default_rng(1))
    hp.view()[:] = arr
    return hp, Plane.from_numpy(hp.data, w, h, bd, pad, pad)
for c in range(3):
    k = "c%d" % c
    W, H, xdec, ydec, bd, damping = (int(v) for v in G[k + "_meta"])
    dt = np.uint16 if bd > 8 else np.uint8
    planes = [plane_from(G[k + "_in%d" % p].astype(dt), bd) for p in range(3)]
    skip, ci = torch.from_numpy(G[k + "_skip"]).cuda(), torch.from_numpy(G[k + "_ci"]).cuda()
    nby, nbx = H // 8, W // 8
    dc = np.zeros(nby * nbx, CDEF_DIR_CAND)
    dc["x"] = np.tile(np.arange(nbx) * 8, nby); dc["y"] = np.repeat(np.arange(nby) * 8, nbx)
    d, v = ctx.cdef_find_dir_batch(planes[0][1], dc)
    d, v = d.cpu().numpy().reshape(nby, nbx), v.cpu().numpy().reshape(nby, nbx)
    da, va = ctx.cdef_analyze_frame(planes[0][1], W, H, skip.shape[1], skip.shape[0])
    print(k, "W,H,bd,xdec,ydec,damping", W, H, bd, xdec, ydec, damping, "analysis dir mismatches", int((da.cpu().numpy()[:nby, :nbx] != d).sum()),
          "var mismatches", int((va.cpu().numpy()[:nby, :nbx] != v).sum()))
    print(" ystr", G[k + "_ystr"], "uvstr", G[k + "_uvstr"], "ci", G[k + "_ci"].ravel()[:8])
    # filter with the OLD direction search's results in full-grid arrays
    dfull = torch.zeros_like(da); vfull = torch.zeros_like(va)
    dfull[:nby, :nbx] = torch.from_numpy(d).cuda(); vfull[:nby, :nbx] = torch.from_numpy(v).cuda()
    for p in range(3):
        xd, yd = (0, 0) if p == 0 else (xdec, ydec)
        _, dst = plane_from(np.zeros_like(G[k + "_in%d" % p]).astype(dt), bd)
        ctx.cdef_filter_frame_plane_dirs(dfull, vfull, planes[p][1], dst, p, xd, yd, W, H, skip, ci, G[k + "_ystr"], G[k + "_uvstr"], damping, bd)
        got = dst.data.cpu().numpy().view(dt)[16:16 + (H >> yd), dst.xorigin:dst.xorigin + (W >> xd)].astype(np.int64)
        want = G[k + "_out%d" % p].astype(np.int64)
        bad = got != want
        print("  plane", p, "bad px", int(bad.sum()), "of", bad.size, "max |diff|", int(np.abs(got - want).max()))
        if bad.any() and p == 0:
            ys, xs = np.nonzero(bad)
            print("   first bad:", [(int(y), int(x), int(got[y, x]), int(want[y, x]), int(G[k + "_in0"][y, x])) for y, x in list(zip(ys, xs))[:12]])
            print("   bad by (x & 1):", [int(bad[:, i::2].sum()) for i in range(2)], "bad by (x&7):", [int(bad[:, i::8].sum()) for i in range(8)],
                  "bad by y&7:", [int(bad[i::8, :].sum()) for i in range(8)])
            blk = bad.reshape(H // 8, 8, W // 8, 8).sum((1, 3))
            print("   bad per block:\n", blk)
            print("   dir per block:\n", d, "\n   var>>6:\n", v >> 6)
            sk8 = np.array([[G[k + "_skip"][2 * by:2 * by + 2, 2 * bx:2 * bx + 2].all() for bx in range(nbx)] for by in range(nby)])
            print("   skip per block:\n", sk8.astype(int))
