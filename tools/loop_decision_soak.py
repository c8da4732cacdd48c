#!/usr/bin/env python3
"""rdo_loop_decision's whole iteration, device against oracle through the same host driver, on random frames: sizes that
are and are not multiples of 64, the three chroma formats, bit depths 8 / 10 / 12, quantizers on both sides of the
restoration geometry's thresholds (areas 1x1 .. 8x8 superblocks), 2 / 4 / 8 strength presets, skip densities.
    python tools/loop_decision_soak.py [--seconds 120] [--seed 0]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import loop_decision_util as U
    import test_gpu_loop_decision as T
    from rav1e_amd.api import Context
    ctx = Context(0)
    rng = np.random.default_rng(a.seed)
    t0, n, ev, passes, bad = time.time(), 0, 0, 0, []
    while time.time() - t0 < a.seconds:
        W, H = int(rng.integers(9, 60)) * 8, int(rng.integers(9, 40)) * 8
        xdec, ydec = [(1, 1), (1, 1), (1, 0), (0, 0)][int(rng.integers(0, 4))]
        bd, q = int(rng.choice([8, 10, 12])), int(rng.choice([60, 100, 150, 180, 220]))
        c = U.synthetic_case(W, H, xdec, ydec, bd, q, [a.seed, n], n_idx=int(rng.choice([2, 4, 8])),
                             p_skip=float(rng.choice([0.0, 0.2, 0.6])), noise=int(rng.choice([2, 6, 12])))
        dev, ora = U.driver(T.device_backend(ctx, c), c), U.driver(U.OracleBackend(c), c)
        bd_, ld_ = dev.run()
        bo_, lo_ = ora.run()
        ok = np.array_equal(bd_, bo_) and ld_ == lo_ and dev.events == ora.events and dev.passes == ora.passes
        if not ok:
            bad.append((W, H, xdec, ydec, bd, q, n))
        n += 1
        ev += sum(len(v) for v in ora.events.values())
        passes += ora.passes
    print("loop_decision_soak: %d random frames, %d plane errors compared in call order, %.1f passes per frame, %d mismatching%s; %.0f s"
          % (n, ev, passes / max(n, 1), len(bad), (" " + str(bad[:4])) if bad else "", time.time() - t0))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
