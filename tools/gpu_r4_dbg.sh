#!/bin/bash
cd /root/repo || exit 1
mkdir -p gpurun_out/r04_dbg
timeout 200 python -u - > gpurun_out/r04_dbg/push.log 2>&1 <<'PY'
import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from rav1e_amd import tiles, workload as W
from rav1e_amd.api import Context, Plane
fw, fh, bd = 777, 211, 8
src = W.random_plane_array(fw, fh, bd, 3)
src[src == 0] = 1
ctx = Context(0)
sp = Plane.from_numpy(src, fw, fh, bd, 88, 88)
d = Plane.from_numpy(np.zeros_like(src), fw, fh, bd, 88, 88)
ptrs = (C.c_void_p * 1)(d.data.data_ptr())
for rect in [(0, 10, 20, 300, 60), (0, 8, 0, 24, 4), (0, 0, 0, 16, 1), (0, 3, 5, 100, 6)]:
    d.data.zero_()
    x = np.zeros(1, tiles.PUSH_RECT); x[0] = rect
    p = sp.cstruct()
    rc = ctx.lib.r1_push_rects(ctx.h, C.byref(p), ptrs, 1, x.ctypes.data, 1, None)
    torch.cuda.synchronize()
    g = d.data.cpu().numpy()
    want = np.zeros_like(src)
    _, x0, y0, x1, y1 = rect
    want[88+y0:88+y1, 88+x0:88+x1] = src[88+y0:88+y1, 88+x0:88+x1]
    bad = np.argwhere(g != want)
    print(rect, "rc", rc, "bad", len(bad))
    for y in sorted(set(bad[:, 0]))[:3]:
        cols = bad[bad[:, 0] == y][:, 1]
        print("  row", y, "cols", cols.min(), "..", cols.max(), "n", len(cols), "g nonzero cols",
              np.flatnonzero(g[y])[:3], np.flatnonzero(g[y])[-3:], "want", 88 + x0, 88 + x1 - 1)
PY
cat gpurun_out/r04_dbg/push.log
