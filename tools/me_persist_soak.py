#!/usr/bin/env python3
"""Soak of r1_estimate_tile_motion_batch's persistent launches (launch_mode 2: spin-waits on progress
words, every job pinned to one XCD; launch_mode 3: not pinned) against the diagonal launches (launch_mode 1, the path the
parity tests pin to the oracle and to the executed reference): random frame sizes (ragged last
tiles, sizes that are not multiples of 64), tile grids, 1..4 reference frames, previous-frame
statistics present or not, 8 / 10-bit, repeated calls on the same ring slot (epoch reuse) and
two calls in flight on two streams.  Every MEStats entry must be identical.  Run on the GPU box:

    python tools/me_persist_soak.py [--seconds 90]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    from rav1e_amd.api import Context, Plane, me_lambdas
    ctx = Context(0)
    rng = np.random.default_rng(args.seed)
    dev = lambda pyr: [Plane.from_numpy(p.data, p.width, p.height, p.bit_depth, p.xpad, p.ypad) for p in pyr]
    side = torch.cuda.Stream()
    t0, cases, calls, bad_cases, entries, moving = time.time(), 0, 0, [], 0, 0
    while time.time() - t0 < args.seconds:
        bd = int(rng.choice([8, 10]))
        w = int(rng.integers(5, 40)) * 32 + int(rng.choice([0, 0, 8, 16, 24]))
        h = int(rng.integers(4, 30)) * 32 + int(rng.choice([0, 0, 8, 16, 24]))
        nref = int(rng.integers(1, 5))
        tw = int(rng.choice([64, 128, 192, 256, 320, 512]))
        th = int(rng.choice([64, 128, 192, 256, 384, 576]))
        mx = (1 << bd) - 1
        f = rng.standard_normal((h + 64, w + 64)).astype(np.float32)
        for _ in range(2):
            f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
            f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
        f = ((f - f.min()) / (f.max() - f.min()) * mx).astype(np.int64)
        org = f[32:32 + h, 32:32 + w]
        do = dev(O.me_pyramid(org, bd))
        drs = []
        for _ in range(nref):
            dx, dy = int(rng.integers(-12, 13)), int(rng.integers(-12, 13))
            ref = np.clip(f[32 + dy:32 + dy + h, 32 + dx:32 + dx + w] + rng.integers(-3, 4, (h, w)), 0, mx)
            drs.append(dev(O.me_pyramid(ref, bd)))
        rows, cols = (h + 3) // 4, (w + 3) // 4
        tiles = [(x, y, min(tw, w - x), min(th, h - y)) for y in range(0, h, th) for x in range(0, w, tw)]
        if len(tiles) * nref > 256:
            continue
        lam = me_lambdas(float(rng.uniform(5, 200)))
        with_prev = bool(rng.integers(0, 2))
        prev = [torch.from_numpy(rng.integers(-64, 65, (rows, cols, 2)).astype(np.int32)).cuda() if with_prev
                else None for _ in range(nref)]
        if with_prev:
            for pv in prev:      # [row | col << 16, normalized_sad]
                pv[..., 0] = (pv[..., 0] & 0xFFFF) | ((pv[..., 1] & 0xFFFF) << 16)
                pv[..., 1] = torch.randint(0, 1 << 20, (rows, cols), device="cuda", dtype=torch.int32)
        kw = dict(allow_hp=bool(rng.integers(0, 2)), allow_full_search=bool(rng.integers(0, 2)),
                  me_range_scale=int(rng.choice([1, 2])))

        def run(mode, stream=None):
            st = [torch.zeros((rows, cols, 2), dtype=torch.int32, device="cuda") for _ in range(nref)]
            jobs = [dict(org=do, ref=drs[r], stats=st[r], prev=prev[r], tile=t) for r in range(nref) for t in tiles]
            if stream is None:
                ctx.estimate_tile_motion(jobs, cols, rows, bd, lam, launch_mode=mode, **kw)
            else:
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):
                    ctx.estimate_tile_motion(jobs, cols, rows, bd, lam, launch_mode=mode, **kw)
            return st
        want = run(1)
        a = run(2)
        b = run(2, side)          # a second persistent launch in flight beside the next one
        c = run(2)
        d = run(3)                # not pinned: any wave, any row, agent-scope write-through
        torch.cuda.synchronize()
        calls += 5
        cases += 1
        entries += sum(int(x[..., 1].numel()) for x in want)
        moving += sum(int((x[..., 0] != 0).sum()) for x in want)
        for got in (a, b, c, d):
            if any(not torch.equal(g, x) for g, x in zip(got, want)):
                n = sum(int((g != x).any(-1).sum()) for g, x in zip(got, want))
                bad_cases.append((w, h, bd, nref, tw, th, with_prev, kw, n))
                break
    print("me_persist_soak: %d geometries (%d calls, persistent launches alone and two in flight), "
          "%d with a MEStats entry different from the diagonal launches; %d entries compared per launch "
          "mode, %d of them with a non-zero vector; %.0f s"
          % (cases, calls, len(bad_cases), entries, moving, time.time() - t0))
    for bc in bad_cases[:10]:
        print("  BAD", bc)
    sys.exit(1 if bad_cases else 0)


if __name__ == "__main__":
    main()
