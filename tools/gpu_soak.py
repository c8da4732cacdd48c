#!/usr/bin/env python3
"""Randomised soak of the GPU parity tests: every test of tests/test_gpu_parity.py that draws
its inputs from np.random.default_rng(<int>) is re-run with the seeds shifted, so the kernels
meet inputs the committed tests never produce.  Run on the GPU box:

    python tools/gpu_soak.py [--rounds 8] [-k substring]
"""
import argparse
import inspect
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("-k", default="")
    args = ap.parse_args()
    import oracle_lib as O
    from rav1e_amd.api import Context
    import test_gpu_parity as T
    import test_small_input_intra as S
    ctx, oracle = Context(0), O.lib()
    real = np.random.default_rng
    shift = [0]

    def shifted(seed=None, *a, **kw):
        if isinstance(seed, (int, np.integer)):
            seed = int(seed) + shift[0]
        return real(seed, *a, **kw)
    np.random.default_rng = shifted
    skip = ("golden", "known_answers", "spec_model", "bad_arg", "reject", "einval", "compat")
    tests = [(n, f) for mod in (T, S) for n, f in inspect.getmembers(mod, inspect.isfunction)
             if n.startswith("test_") and args.k in n and not any(s in n for s in skip)]
    failed, ran, t0 = [], 0, time.time()
    for rnd in range(1, args.rounds + 1):
        shift[0] = 1000003 * rnd
        for name, f in tests:
            marks = [m for m in getattr(f, "pytestmark", []) if m.name == "parametrize"]
            sig = [p for p in inspect.signature(f).parameters if p not in ("ctx", "oracle")]
            combos = [{}]
            for m in marks:
                names = [s.strip() for s in m.args[0].split(",")]
                combos = [dict(c, **dict(zip(names, v if len(names) > 1 else (v,)))) for c in combos for v in m.args[1]]
            for c in combos:
                kw = {k: v for k, v in c.items() if k in sig}
                if "ctx" in inspect.signature(f).parameters:
                    kw["ctx"] = ctx
                else:
                    continue                       # CPU-only test
                if "oracle" in inspect.signature(f).parameters:
                    kw["oracle"] = oracle
                try:
                    f(**kw)
                    ran += 1
                except Exception as e:             # noqa: BLE001 -- reported, the soak goes on
                    failed.append((rnd, name, c, repr(e)[:300]))
                    print("FAIL", rnd, name, c, repr(e)[:300], flush=True)
    print("soak: %d test runs over %d rounds, %d failures, %.0f s" % (ran, args.rounds, len(failed), time.time() - t0))
    ctx.close()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
