#!/usr/bin/env python3
"""Launch time of the fused candidate kernel by 8-tap filter class of the candidates (4K 8-bit, K = 16):
all REGULAR, all SHARP, the nine pairs mixed per candidate (bench.py's filter_pairs line), and the
mixed list sorted by filter pair (what a host that batches by filter class would send)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rav1e_amd import workload as W            # noqa: E402
from rav1e_amd.api import Context, Plane       # noqa: E402


def main():
    bd = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    fw, fh, k = 3840, 2160, 16
    ctx = Context(0)
    po = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 21), fw, fh, bd, 88, 88)
    pr = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 22), fw, fh, bd, 88, 88)
    mixed = W.speed6_ladder(fw, fh, k, seed=7, mix_filters=True)
    res = {}
    for s in W.LADDER:
        base = mixed[s]
        n = len(base)
        variants = {}
        for name, (mx, my) in (("all_regular", (0, 0)), ("all_smooth", (1, 1)), ("all_sharp", (2, 2)),
                               ("sharp_x_only", (2, 0)), ("sharp_y_only", (0, 2))):
            c = base.copy()
            c["mode_x"], c["mode_y"] = mx, my
            variants[name] = c
        variants["mixed"] = base
        order = np.argsort(base["mode_y"].astype(np.int32) * 3 + base["mode_x"], kind="stable")
        variants["mixed_sorted_by_pair"] = base[order]
        order = np.argsort((base["mode_y"] == 2).astype(np.int32), kind="stable")
        variants["mixed_sorted_by_sharp_y"] = base[order]
        o = {"sad": torch.empty(n, dtype=torch.int32, device="cuda"), "satd": torch.empty(n, dtype=torch.int32, device="cuda"),
             "coeffs": torch.empty((n, s * s), dtype=torch.int16 if bd == 8 else torch.int32, device="cuda")}
        row = {}
        for name, c in variants.items():
            dc = torch.from_numpy(np.ascontiguousarray(c).view(np.uint8).reshape(-1).copy()).cuda()
            f = ctx.prepare_rdo_cand(po, pr, s, s, dc, n, o)
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            ev = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            row[name] = round(sum(a.elapsed_time(b) for a, b in ev) / len(ev), 4)
        res[str(s)] = row
        print(json.dumps({"size": s, "bit_depth": bd, "launch_ms": row}))


main()
