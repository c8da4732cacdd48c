#!/usr/bin/env python3
"""Where a wave of the headline kernel spends its life: runs the bench workload once per block
size on the experiment build (make -C rav1e_amd/csrc prof) whose k_rdo_cand stamps s_memtime at
its phase boundaries.  Prints mean cycles per wave per phase (wall clock of the wave, i.e.
including the time it is parked or waiting for an issue slot)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["R1_LIB"] = os.path.join(ROOT, "rav1e_amd", "librav1e_hip_prof.so")


def main():
    import torch
    from rav1e_amd import _lib
    _lib.SO = os.environ["R1_LIB"]
    from rav1e_amd import workload as W
    from rav1e_amd.api import Context, Plane
    bd = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = Context(0)
    lib = ctx.lib
    lib.r1_debug_phase_prof.restype = C.c_int
    lib.r1_debug_phase_prof.argtypes = [C.c_void_p, C.c_int]
    fw, fh = 3840, 2160
    org = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 1), fw, fh, bd, 88, 88)
    ref = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 2), fw, fh, bd, 88, 88)
    lad = W.speed6_ladder(fw, fh, 16)
    names = ["A2 lds+barrier", "B1 mc+resid", "B2 sad+satd", "C col tx", "D row tx+store", "A0 descriptor",
             "A1 pixels"]
    for s in W.LADDER:
        c = lad[s]
        dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
        for _ in range(50):
            ctx.rdo_cand_batch(org, ref, s, s, dc, n=len(c))
        torch.cuda.synchronize()
        buf = np.zeros((4096, 8), np.uint64)
        lib.r1_debug_phase_prof(buf.ctypes.data, 1)
        ctx.rdo_cand_batch(org, ref, s, s, dc, n=len(c))
        torch.cuda.synchronize()
        lib.r1_debug_phase_prof(buf.ctypes.data, 1)
        rows = buf[buf[:, 1] > 0][:, :7].astype(np.float64)
        per = rows.mean(axis=0)
        print(json.dumps({"size": s, "bd": bd, "sampled_waves": len(rows),
                          "cycles_per_wave": {n: round(v) for n, v in zip(names, per)},
                          "total": round(per.sum()), "median_total": round(float(np.median(rows.sum(axis=1))))}))
    ctx.close()


if __name__ == "__main__":
    main()
