#!/usr/bin/env python3
"""Per-pass workgroup lifetimes of k_me_diag (experiment build with -DR1_ME_PROF installed as
rav1e_amd/librav1e_hip.so): which pass bounds a diagonal launch.  python tools/me_prof.py [cfg]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from rav1e_amd.api import Context, Plane, me_lambdas
sys.argv = sys.argv[:1] + sys.argv[1:]
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
import importlib.util
spec = importlib.util.spec_from_file_location("bench_me", os.path.join(ROOT, "tools", "bench_me.py"))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
w, h, bd = 3840, 2160, int(os.environ.get('R1_ME_PROF_BD', '8'))
f = bm.texture(w, h, bd, 1)
rng = np.random.default_rng(2)
org = f[32:32 + h, 32:32 + w]
shifts = [(5, -9), (-3, 2), (12, 7), (0, -1)]
refs = [np.clip(f[32 + dy:32 + dy + h, 32 + dx:32 + dx + w] + rng.integers(-2, 3, (h, w)), 0, (1 << bd) - 1) for dx, dy in shifts]
po = O.me_pyramid(org, bd)
prs = [O.me_pyramid(r, bd) for r in refs]
dev = lambda pyr: [Plane.from_numpy(p.data, p.width, p.height, bd, p.xpad, p.ypad) for p in pyr]
do, drs = dev(po), [dev(p) for p in prs]
lam = me_lambdas(30.0)
ctx = Context(0)
rows, cols = h // 4, w // 4
nx, ny, nref = [(1, 1, 1), (1, 1, 4), (2, 2, 4), (4, 4, 4), (4, 2, 3)][cfg]
tw = -(-(w // nx) // 64) * 64
th = -(-(h // ny) // 64) * 64
tl = [(x, y, min(tw, w - x), min(th, h - y)) for y in range(0, h, th) for x in range(0, w, tw)]
stats = [torch.zeros((rows, cols, 2), dtype=torch.int32, device="cuda") for _ in range(nref)]
jobs = [dict(org=do, ref=drs[r], stats=stats[r], tile=t) for r in range(nref) for t in tl]
lib = ctx.lib
buf = (C.c_ulonglong * 12)()
ctx.estimate_tile_motion(jobs, cols, rows, bd, lam)
torch.cuda.synchronize()
lib.r1_debug_me_prof.argtypes = [C.c_void_p, C.c_int]
lib.r1_debug_me_prof(buf, 1)
if hasattr(lib, "r1_debug_me_step"):
    lib.r1_debug_me_step.argtypes = [C.c_void_p, C.c_int]
    lib.r1_debug_me_step(buf, 1)
if hasattr(lib, "r1_debug_me_fine"):
    lib.r1_debug_me_fine.argtypes = [C.c_void_p, C.c_int]
    lib.r1_debug_me_fine((C.c_ulonglong * 8)(), 1)
for s in stats:
    s.zero_()
ctx.estimate_tile_motion(jobs, cols, rows, bd, lam)
torch.cuda.synchronize()
lib.r1_debug_me_prof(buf, 0)
v = np.array(list(buf), np.float64).reshape(3, 4)
if hasattr(lib, "r1_debug_me_fine"):
    fb = (C.c_ulonglong * 8)()
    lib.r1_debug_me_fine.argtypes = [C.c_void_p, C.c_int]
    lib.r1_debug_me_fine(fb, 0)
    fv = np.array(list(fb), np.float64)
    n = max(fv[4], 1)
    print(json.dumps({"non_extensive_searches": int(fv[4]), "candidate_scan_us": round(fv[1] / n / 100, 2),
                      "diamond_us": round(fv[2] / n / 100, 2), "diamond_iterations": round(fv[3] / n, 2),
                      "predictor_gather_us (all searches)": round(fv[0] / max(n, 1) / 100, 2)}))
if hasattr(lib, "r1_debug_me_step"):
    # the persistent launch (the product path from 8 jobs on): where a block search's time goes
    lib.r1_debug_me_step.argtypes = [C.c_void_p, C.c_int]
    lib.r1_debug_me_step(buf, 0)
    sv = np.array(list(buf), np.float64).reshape(3, 4)
    for q in range(3):
        n = max(sv[q, 0], 1)
        print(json.dumps({"persist_pass": q + 1, "block_searches": int(sv[q, 0]), "setup_wait_us": round(sv[q, 1] / n / 100, 2),
                          "search_us": round(sv[q, 2] / n / 100, 2), "store_us": round(sv[q, 3] / n / 100, 2)}))
for q in range(3):
    n = max(v[q, 0], 1)
    print(json.dumps({"pass": q, "jobs": len(jobs), "workgroups": int(v[q, 0]), "mean_us": round(v[q, 1] / n / 100, 1),
                      "max_us": round(v[q, 2] / 100, 1), "refine_mean_us": round(v[q, 3] / n / 100, 1)}))
