#!/usr/bin/env python3
"""Ablation timing of the fused candidate kernel: which outputs are requested
(sad / satd / coeffs) -- a quick way to see where the time goes per size."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rav1e_amd import workload as W
from rav1e_amd.api import Context, Plane
ctx = Context(0)
fw, fh, bd = 3840, 2160, int(os.environ.get("BD", "8"))
org = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 1), fw, fh, bd, 88, 88)
ref = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 2), fw, fh, bd, 88, 88)
cands = W.speed6_ladder(fw, fh, 16)
for s in W.LADDER:
    c = cands[s]
    dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
    n = len(c)
    outs = {"sad": torch.empty(n, dtype=torch.int32, device="cuda"),
            "satd": torch.empty(n, dtype=torch.int32, device="cuda"),
            "coeffs": torch.empty((n, s * s), dtype=torch.int16 if bd == 8 else torch.int32, device="cuda")}
    row = {}
    for name, (a, b, cc) in {"all": (1, 1, 1), "no_tx": (1, 1, 0), "sad_only": (1, 0, 0),
                             "tx_only": (0, 0, 1), "satd_only": (0, 1, 0)}.items():
        f = lambda: ctx.rdo_cand_batch(org, ref, s, s, dc, n=n, want_sad=bool(a), want_satd=bool(b),
                                       want_coeffs=bool(cc), outs=dict(outs))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        row[name] = round(e0.elapsed_time(e1) / 10, 4)
    print(s, json.dumps(row), flush=True)
