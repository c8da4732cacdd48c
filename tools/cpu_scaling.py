#!/usr/bin/env python3
"""How the vectorised CPU leg (oracle/fast_cand.c) scales with threads on this host, and what
the container is allowed to use -- the context for bench.py's cpu_baseline.cores."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from rav1e_amd import workload as W  # noqa: E402


def main():
    info = {"sched_affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count()}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            info[f] = open(f).read().strip()
        except OSError:
            pass
    try:
        info["model"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    print(json.dumps(info))
    L = O.lib()
    fw, fh, bd = 3840, 2160, 8
    a, b = O.HostPlane(fw, fh, bd), O.HostPlane(fw, fh, bd)
    a.data, b.data = W.random_plane_array(fw, fh, bd, 1), W.random_plane_array(fw, fh, bd, 2)
    pa, pb = a.cstruct(), b.cstruct()
    c = W.speed6_ladder(fw, fh, 16)[16]
    n = len(c)
    sad, satd = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    co = np.zeros((n, 256), np.int16)
    for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if th > 2 * (os.cpu_count() or 1):
            break
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            L.r1o_fast_rdo_cand_batch(C.byref(pa), C.byref(pb), 16, 2, O.ptr(c), n, th, O.ptr(sad), O.ptr(satd),
                                      O.ptr(co))
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"threads": th, "Mpixels_s": round(n * 256 / best / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
