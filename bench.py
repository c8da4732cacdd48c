#!/usr/bin/env python3
"""bench.py -- RDO-candidate Mpixels/s (dist + fwd_tx + mc) at 4K speed-6.

One "step" = one pass of the fused hot path (put_8tap -> SAD + SATD -> diff ->
forward DCT, one launch per block size) over every candidate of a synthetic 4K
frame: for each block of the speed-6 ladder 64/32/16/8 (rav1e_amd/workload.py)
K candidates with random motion vectors and 1/16-pel fractions.  Inputs are
resident in HBM before the timed region.  N > 1: one process per GPU, rank r
owns tile r of the frame (uniform tiling, src/tiling/tiler.rs:56); every step
the ranks exchange the tile-boundary rectangles (point to point) and their
tiles of the reference plane (all-gather) through the C ABI's r1_comm_* entry
points (RCCL), in stream order with the launches.

Prints ONE JSON line on rank 0 (contract in the task statement), carrying
`roofline` for the dominant kernel (VALU-issue and HBM roofs), `cpu_baseline`
(vectorised and scalar CPU legs timed on the host cores, rank 0 / N=1 only),
`parity_checked` / `parity_ok` (the GPU's buffers of the timed steps against
those CPU legs) and `extra_lines` (chroma planes, mixed filter pairs, mixed
transform types).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--k", type=int, default=16, help="candidates per block")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target duration of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--single-device", action="store_true",
                    help="dry run: every rank uses cuda:0 (to exercise the N > 1 control flow on a "
                         "1-GPU box; numbers are meaningless)")
    ap.add_argument("--chain", choices=("cand", "full", "pixel"), default="cand",
                    help="cand: the headline fused candidate (BASELINE.json north_star); full: the "
                         "same candidate carried through quantize / tx-domain distortion / rate "
                         "(r1_rdo_full_cand_batch, SURVEY 8f N4); pixel: carried through quantize / "
                         "inverse transform / cdef_dist (r1_rdo_pixel_cand_batch) -- supplementary lines")
    ap.add_argument("--qindex", type=int, default=100)
    ap.add_argument("--streams", type=int, default=0,
                    help="0 (default): 1 at N = 1, 4 at N > 1 (a rank's tile makes each of the four launches "
                         "a partial wave of workgroups: +5 %% on a tile-sized workload, same-box A/B). "
                         "> 1: the launches of a step (one per block size, independent of each other) "
                         "go to one HIP stream per size, so that a launch fills the CUs the previous "
                         "one is draining; the steps that carry timing events stay on one stream")
    ap.add_argument("--stream-plan", default="",
                    help="with --streams > 1: which stream runs which sizes, e.g. 64h/32l/16l/8l ('/' separates "
                         "streams, ',' sizes on one stream in order, h / l = high / low queue priority)")
    ap.add_argument("--step-join", action="store_true",
                    help="with --streams > 1: the size streams fork from / join the main stream every step")
    ap.add_argument("--event-every", type=int, default=4,
                    help="every n-th timed step carries the per-launch timing events (and runs on one stream)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps for this long BEFORE the W warm-up steps, so that short runs "
                         "(small W and K) are measured at the clocks a long run settles at; 0 = off")
    ap.add_argument("--exchange", choices=("auto", "push", "p2p"), default="auto",
                    help="N > 1: how tiles travel.  push = direct peer stores into IPC-mapped planes "
                         "(r1_comm_push_*), p2p = grouped RCCL send / receive (r1_comm_exchange_halos + "
                         "r1_comm_allgather_tiles); auto = push if every rank can map its peers' planes and "
                         "the tagged-tile check passes through it, else p2p")
    ap.add_argument("--overlap-exchange", action="store_true",
                    help="N > 1 with peer stores: the halo stores of a step leave on a side stream as soon as the tile's "
                         "border is written, beside the step's launches (tiles.TileRing.begin_overlapped / "
                         "finish_overlapped); off by default -- never timed on hardware")
    ap.add_argument("--verify-exchange", action="store_true",
                    help="N = 1: run the tagged-tile self-check of the exchange through the C-ABI communicator "
                         "(world 1) as the N > 1 runs always do; the result is config.exchange_ok")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the supplementary workloads (chroma planes, mixed filter pairs, mixed "
                         "transform types) reported under `extra_lines`")
    ap.add_argument("--detail-out", default=None,
                    help="where the long document goes (default gpurun_out/bench_detail.json); stdout carries "
                         "only the compact record (< 4 KB)")
    ap.add_argument("--no-events", action="store_true",
                    help="skip per-kernel event timing (roofline.achieved falls back to step time)")
    return ap.parse_args()


def physical_cores():
    """(cores the CPU legs may use, logical CPUs visible, note): physical cores in the affinity
    mask, capped by the container's CPU quota (cgroup cpu.max / cfs_quota) -- more runnable
    threads than the quota allows only get throttled"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    sib = set()
    for c in cpus:
        try:
            sib.add(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip())
        except OSError:
            sib.add(str(c))
    phys, note = max(1, len(sib)), "physical cores in the affinity mask"
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < phys:
        phys, note = max(1, int(quota)), "cgroup CPU quota of this container (%g CPUs; %d physical cores visible)" % (
            quota, len(sib))
    return phys, len(cpus), note


def reference_probe():
    """rav1e's own asm path can only be timed where cargo and nasm exist (neither
    is in this image).  Probe at run time; when both are there and RAV1E_SRC names a checkout that
    builds offline, run the reference's criterion groups for this path (benches/bench.rs:199-224)
    and report their medians -- the >= 10x target is then evaluated against THAT figure."""
    import re
    import shutil
    import subprocess
    cargo, nasm, src = shutil.which("cargo"), shutil.which("nasm"), os.environ.get("RAV1E_SRC")
    out = {"cargo": cargo, "nasm": nasm, "rav1e_src": src, "ran": False}
    if not (cargo and nasm and src and os.path.isdir(src)):
        out["note"] = ("rav1e's asm path not timed: needs cargo + nasm on PATH and RAV1E_SRC=<checkout "
                       "with vendored dependencies>; cpu_baseline.value is the vectorised port")
        return out
    try:
        r = subprocess.run([cargo, "bench", "--offline", "--features", "bench", "--bench", "bench", "--",
                            "get_sad|get_satd|forward_transform|put_8tap|prep_8tap"],
                           cwd=src, timeout=1500, capture_output=True, text=True)
        medians, name = {}, None
        for line in r.stdout.splitlines():
            m = re.match(r"^(\S.*?)\s+time:\s+\[\S+ \S+ (\S+) (\S+) ", line)
            if m:
                medians[m.group(1)] = m.group(2) + " " + m.group(3)
                continue
            if line and not line.startswith(" "):
                name = line.strip()
            m = re.match(r"^\s+time:\s+\[\S+ \S+ (\S+) (\S+) ", line)
            if m and name:
                medians[name] = m.group(1) + " " + m.group(2)
        out.update(ran=r.returncode == 0, criterion_medians=medians, returncode=r.returncode)
    except Exception as e:      # the probe never takes the bench down
        out["note"] = "cargo bench failed: %s" % (str(e)[:120],)
    return out


def cpu_baseline(args, host_org, host_ref, cands, outs):
    """Time the CPU side on bounded samples of the step that was just timed on the GPU, and check
    the GPU's buffers against it:
      scalar    oracle/batch.c r1o_rdo_cand_batch -- the restatement every parity test uses;
                a strided sample, one thread per usable core (this is the checker of record);
      vector    oracle/fast_cand.c -- the same candidate written for the compiler's SIMD
                (tests/test_oracle_fast.py holds it equal to the scalar one), on every physical
                core (-> `value`) and on one thread.
    Neither is rav1e's nasm code; reference_probe() says what would be needed for that."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = O.lib()
    phys, logical, cores_note = physical_cores()
    ho = O.HostPlane(args.width, args.height, args.bit_depth)
    hr = O.HostPlane(args.width, args.height, args.bit_depth)
    ho.data, hr.data = host_org, host_ref
    pa, pb = ho.cstruct(), hr.cstruct()
    ct = np.int16 if ho.bpp == 1 else np.int32
    TS = {64: 4, 32: 3, 16: 2, 8: 1}

    def leg(kind, threads, frac, reps=1):
        """-> (pixels, seconds inside the CPU calls, candidates compared, sizes that differ)"""
        px, dt, n_cmp, bad = 0, 0.0, 0, []
        for s, c in cands.items():
            if not len(c):
                continue
            n = max(1, int(len(c) * frac))
            idx = np.arange(0, len(c), max(1, len(c) // n))[:n]
            sub = np.ascontiguousarray(c[idx])
            sad, satd = np.zeros(len(sub), np.uint32), np.zeros(len(sub), np.uint32)
            co = np.zeros((len(sub), s * s), ct)
            L.r1o_set_threads(threads)
            t0 = time.perf_counter()
            for _ in range(reps):
                if kind == "scalar":
                    rc = L.r1o_rdo_cand_batch(C.byref(pa), C.byref(pb), s, s, TS[s], O.ptr(sub), len(sub),
                                              O.ptr(sad), O.ptr(satd), O.ptr(co), None)
                elif kind == "vector512":
                    rc = L.r1o_fast512_rdo_cand_batch(C.byref(pa), C.byref(pb), s, TS[s], O.ptr(sub), len(sub),
                                                      threads, O.ptr(sad), O.ptr(satd), O.ptr(co))
                else:
                    rc = L.r1o_fast_rdo_cand_batch(C.byref(pa), C.byref(pb), s, TS[s], O.ptr(sub), len(sub),
                                                   threads, O.ptr(sad), O.ptr(satd), O.ptr(co))
            dt += time.perf_counter() - t0
            assert rc == 0
            px += reps * len(sub) * s * s
            # what the timed launches left in HBM for these very candidates
            ix = torch.from_numpy(idx.astype(np.int64)).cuda()
            same = (np.array_equal(outs[s]["sad"].index_select(0, ix).cpu().numpy().view(np.uint32), sad)
                    and np.array_equal(outs[s]["satd"].index_select(0, ix).cpu().numpy().view(np.uint32), satd)
                    and np.array_equal(outs[s]["coeffs"].index_select(0, ix).cpu().numpy(), co))
            n_cmp += len(idx)
            if not same:
                bad.append(s)
        return px, dt, n_cmp, bad

    def sized(kind, threads, seconds):
        """grow the strided sample until the CPU calls take about `seconds` (small samples are
        dominated by waking the thread team); a step shorter than that is repeated whole"""
        frac = 0.002
        px, dt, n_cmp, bad = leg(kind, threads, frac)
        while dt < 0.6 * seconds and frac < 1.0:
            frac = min(1.0, frac * min(16.0, max(2.0, seconds / max(dt, 1e-3))))
            px, dt, n_cmp, bad = leg(kind, threads, frac)
        reps = 1
        if dt < 0.6 * seconds:
            reps = int(min(64, max(2, round(seconds / max(dt, 1e-3)))))
            px, dt, n_cmp, bad = leg(kind, threads, 1.0, reps)
        return {"px": px, "dt": dt, "frac": frac, "reps": reps, "n": n_cmp, "bad": bad, "mpx": px / dt / 1e6}

    T = args.cpu_seconds
    wide = bool(L.r1o_fast512_available())
    sc = sized("scalar", phys, 0.15 * T)
    vN = sized("vector", phys, (0.25 if wide else 0.45) * T)
    v1 = sized("vector", 1, (0.15 if wide else 0.3) * T)
    w_legs = {}
    if wide:
        wN = sized("vector512", phys, 0.3 * T)
        w1 = sized("vector512", 1, 0.15 * T)
        w_legs = {"avx2_leg": {"value": round(vN["mpx"], 2), "cores": phys, "one_thread": round(v1["mpx"], 2),
                               "impl": "oracle/fast_cand.c: the same port on 8 x i32 lanes (AVX2, -march=x86-64-v3)"}}
        top, top1, impl = wN, w1, ("oracle/fast_cand512.c: 16 x i32 lanes (AVX-512 F/BW/DQ/VL, -march=x86-64-v4): gcc vector "
                                   "extensions over the same generated transform networks as the scalar oracle, the vertical "
                                   "8-tap pass on vpmaddwd, two 8x8 candidates per vector -- a compiler-vectorised port, NOT "
                                   "rav1e's avx512icl nasm kernels")
    else:
        top, top1, impl = vN, v1, ("oracle/fast_cand.c: gcc vector extensions (AVX2, 8 x i32) over the same generated "
                                   "transform networks as the scalar oracle -- a compiler-vectorised port, NOT rav1e's "
                                   "nasm kernels (this host reports no AVX-512)")
    res = {"value": round(top["mpx"], 2), "unit": "Mpixels/s", "cores": phys, "kind": "port",
           "sample": "%.2f%% of the step's candidates (every ladder size, strided) x %d, %.1f s, one "
                     "OpenMP thread per core" % (100 * top["frac"], top["reps"], top["dt"]),
           "impl": impl,
           "one_thread": {"value": round(top1["mpx"], 2),
                          "sample": "%.2f%% of the candidates x %d, %.1f s" % (100 * top1["frac"], top1["reps"], top1["dt"])},
           **w_legs,
           "scalar_port": {"value": round(sc["mpx"], 2), "threads": phys,
                           "sample": "%.3f%% of the candidates x %d, %.1f s; oracle/batch.c, the parity checker"
                                     % (100 * sc["frac"], sc["reps"], sc["dt"])},
           "logical_cpus": logical, "cores_note": cores_note,
           "reference_probe": reference_probe()}
    legs = [sc, vN, v1] + ([wN, w1] if wide else [])
    parity = {"parity_checked": sc["n"], "parity_checked_vector_leg": sum(l["n"] for l in legs[1:]),
              "parity_ok": not any(l["bad"] for l in legs),
              "bad": sorted(set(sum((l["bad"] for l in legs), [])))}
    return res, parity


def sustain(one_pass, ms):
    """untimed passes for `ms` milliseconds: a line is timed at the clocks a long run settles at, not at
    what the host work before it (candidate lists, the previous line's CPU parity leg) left"""
    import torch
    if ms <= 0:
        return
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            one_pass()
        torch.cuda.synchronize()


def _latest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def _kernel_key(d, bd, size):
    lg = {64: 6, 32: 5, 16: 4, 8: 3, 4: 2}[size]
    base = "k_rdo_cand<%d,%d,%d,%s" % (bd, lg, lg, "short" if bd == 8 else "int")
    # QM = 0: the headline variant (round 5 added a trailing template flag to the kernel's name)
    return next((k for k in (base + ",0,false>", base + ",0>", base + ">") if k in d), None)


def pmc_counters(bd, size, fw, fh, k):
    """Per-launch PMC counters of the dominant kernel (SQ_INSTS_VALU, FETCH_SIZE / WRITE_SIZE ...),
    collected in separate `rocprofv3 --pmc` passes of this same workload (tools/gpu_pmc.sh) and
    summarised by tools/pmc_summary.py under profiles/.  A counter pass cannot run inside the
    timed bench, so the figures are read from the committed summary and only used when it was
    taken on the default workload."""
    f = _latest("r*_pmc_summary.json")
    if (fw, fh, k) != (3840, 2160, 16) or not f:
        return None, None
    d = json.load(open(f))
    key = _kernel_key(d, bd, size)
    return (d[key], os.path.basename(f)) if key else (None, None)


def valu_issue_model(bd, size):
    """Issue cycles per wave64 VALU instruction for this kernel's static instruction mix
    (tools/isa_hist.py --json, priced with tools/ubench/valu_rate.hip's measurements)."""
    f = _latest("r*_isa_mix.json")
    if not f:
        return None
    d = json.load(open(f))
    key = _kernel_key(d, bd, size)
    if not key:
        return None
    return dict(d[key], source=os.path.basename(f), model=d.get("_model"))


N_SIMD = 256 * 4          # MI355X: 256 CUs x 4 SIMDs
NOMINAL_GHZ = 2.4         # the clock tools/ubench/valu_rate.hip's cycle counts are expressed in


INFINITY_CACHE_BYTES = 256 << 20   # MI355X_MICROARCH.md: 256 MiB memory-side cache in front of HBM


def launch_pmc(key):
    """FETCH_SIZE / WRITE_SIZE of one launch of an extra / config line.  key = (kernel name prefix as
    tools/pmc_summary.py shortens it, grid size in work-items): the counter pass (tools/gpu_pmc_lines.sh:
    this bench with its lines under rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes) is
    summarised per (kernel, grid) under profiles/ -- a counter pass cannot run inside a timed bench."""
    f = _latest("r*_pmc_launches.json")
    if not f or not key:
        return None, None
    d = json.load(open(f))
    prefix, grid = key
    if grid is None:       # the kernel instantiation has ONE launch geometry in the counter pass: take it
        hits = [k for k in d if k.startswith(prefix + "@")]
        if len(hits) != 1:
            return None, None
        grid = int(hits[0].rsplit("@", 1)[1])
    for k, v in d.items():
        if k.startswith(prefix) and k.endswith("@%d" % grid) and ("hbm_traffic_bytes" in v or "SQ_INSTS_VALU" in v):
            return {"dominant_traffic_bytes": v.get("hbm_traffic_bytes"), "kernel": k,
                    "FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "WRITE_SIZE_KiB": v.get("WRITE_SIZE"),
                    "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_WAVES": v.get("SQ_WAVES"),
                    "dispatches_averaged": v.get("n")}, os.path.basename(f)
    return None, None


def rdo_launch_key(bd, size, qm, n):
    lg = {64: 6, 32: 5, 16: 4, 8: 3, 4: 2}[size]
    nc = 64 // size
    # the grid is rounded up to a multiple of 8 workgroups (XCD-aware mapping, csrc/rdo_cand.hip)
    return ("k_rdo_cand<%d,%d,%d,%s,%d" % (bd, lg, lg, "short" if bd == 8 else "int", qm),
            ((((n + nc - 1) // nc) + 7) & ~7) * 64)


def build_roofline(kname, abytes, launch_ms, pmc, pmc_src, mix, working_set=None, write_bytes=None,
                   line_counters=None, line_src=None):
    """The dominant launch against its roofs.
    Top level = the bench contract's HBM roofline: `achieved` = ALGORITHMIC bytes per launch (SURVEY 8(d)
    per-candidate bytes x candidates) / the launch's live time, `peak` = 8 TB/s, `traffic` = HBM bytes per
    launch by the PMC counters (2 x FETCH_SIZE + WRITE_SIZE KiB, MI355X_MICROARCH.md), or null.
    `binding_roof` says which roof actually limits the launch; for the fused kernels that is VALU issue
    (`valu`: three yardsticks), so `frac` -- kept as the contract defines it -- is NOT how close the
    kernel is to its limit.
    working_set / write_bytes: bytes of the input planes the launch reads from / bytes it writes.  A
    read working set below the 256 MiB Infinity Cache is served from that cache from the second step
    on: `hbm.reads_cache_resident`, and `hbm.frac_unique` prices only what must move (each input byte
    once + the writes)."""
    sec = launch_ms * 1e-3
    hbm_ach = abytes / sec / 1e9
    traffic, traffic_src = None, None
    if pmc and "hbm_traffic_bytes" in pmc:
        traffic, traffic_src = int(pmc["hbm_traffic_bytes"]), pmc_src
    elif line_counters and line_counters.get("dominant_traffic_bytes") is not None:
        traffic, traffic_src = int(line_counters["dominant_traffic_bytes"]), line_src
    hbm = {"achieved": round(hbm_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(hbm_ach / HBM_PEAK_GBS, 4), "traffic": traffic,
           "frac_by_counters": round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
           "traffic_note": ("%s: 2*FETCH_SIZE + WRITE_SIZE KiB per dispatch (gfx950 FETCH_SIZE correction, "
                            "MI355X_MICROARCH.md); below the algorithmic bytes where the K candidates of a "
                            "block share window rows in L2 / Infinity Cache" % traffic_src) if traffic else None,
           "algorithmic_bytes_per_launch": int(abytes)}
    if line_counters:
        hbm["counters_of_the_line"] = {k: line_counters[k] for k in line_counters if k != "dominant_traffic_bytes"}
    if working_set is not None:
        uniq = min(int(abytes), int(working_set)) if write_bytes is None else \
            min(int(abytes) - int(write_bytes), int(working_set)) + int(write_bytes)
        hbm.update({"read_working_set_bytes": int(working_set),
                    "reads_cache_resident": bool(working_set < INFINITY_CACHE_BYTES),
                    "unique_bytes_per_launch": int(uniq),
                    "frac_unique": round(uniq / sec / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_note": ("the input planes (%.1f MB) stay in the 256 MiB Infinity Cache between steps: "
                                  "`frac` is algorithmic bytes over time, served mostly by that cache, NOT an HBM "
                                  "fraction; `frac_unique` counts each input byte once plus the writes"
                                  % (working_set / 1e6)) if working_set < INFINITY_CACHE_BYTES else
                                 "the read working set exceeds the Infinity Cache"})
    common = {"bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": hbm["frac"], "traffic": traffic,
              "kernel": kname, "avg_launch_ms": round(launch_ms, 4), "hbm": hbm,
              "algorithmic_bytes_per_launch": int(abytes)}
    valu = None
    if pmc and mix and "SQ_INSTS_VALU" in pmc:
        cost = mix["issue_cycles_per_valu"]
        ach = pmc["SQ_INSTS_VALU"] / sec / 1e9
        # clock the chip held during the counter pass: GRBM_GUI_ACTIVE is summed over the 8 XCDs
        # (tools/pmc_summary.py divides by the dispatch's own duration in that pass)
        clk = pmc.get("clock_ghz_by_counters")
        peak_ubench = N_SIMD * NOMINAL_GHZ / cost
        peak_guide = N_SIMD * NOMINAL_GHZ / 2.0
        valu = {"achieved": round(ach, 2), "unit": "G wave64-instr/s",
                "insts_per_launch": int(pmc["SQ_INSTS_VALU"]),
                # 1: this kernel's static mix priced with the issue cycles measured on the chip
                #    (tools/ubench/valu_rate{,2}.hip: add / sub / shifts / logic 2.5, everything else 4.5)
                "peak_ubench_mix": round(peak_ubench, 2), "frac_ubench_mix": round(ach / peak_ubench, 4),
                "issue_cycles_per_inst": cost, "fast_slow_static": [mix["fast"], mix["slow"]],
                # 2: MI355X_MICROARCH.md's constant -- every wave64 VALU instruction in 2 cycles at 2.4 GHz
                "peak_guide": round(peak_guide, 2), "frac_guide": round(ach / peak_guide, 4),
                # 3: the same 2-cycle rate at the clock the counters say the chip held
                "clock_ghz_by_counters": round(clk, 3) if clk else None,
                "frac_guide_at_measured_clock": round(ach / (N_SIMD * clk / 2.0), 4) if clk else None,
                "sources": [pmc_src, mix["source"], "tools/ubench/valu_rate.hip", "tools/ubench/valu_rate2.hip"]}
    if not valu and line_counters and line_counters.get("SQ_INSTS_VALU"):
        # a line's own launch: instruction counters of the same (kernel, grid) from the counter pass over this bench
        # (tools/gpu_lease.sh pmc_lines), priced with the guide's constant only (no static mix for these kernels)
        ach = line_counters["SQ_INSTS_VALU"] / sec / 1e9
        peak_guide = N_SIMD * NOMINAL_GHZ / 2.0
        valu = {"achieved": round(ach, 2), "unit": "G wave64-instr/s", "insts_per_launch": int(line_counters["SQ_INSTS_VALU"]),
                "peak_guide": round(peak_guide, 2), "frac_guide": round(ach / peak_guide, 4), "frac_ubench_mix": None,
                "sources": [line_src]}
    if valu and valu.get("frac_ubench_mix") is not None:
        binding = "valu" if valu["frac_ubench_mix"] >= (hbm["frac_by_counters"] or hbm["frac"]) else "hbm"
    elif valu:
        hb = hbm["frac_by_counters"] if hbm["frac_by_counters"] is not None else (hbm.get("frac_unique") or hbm["frac"])
        binding = "launch" if launch_ms < 0.02 else ("valu" if valu["frac_guide"] >= hb else "hbm")
        if binding != "launch" and max(valu["frac_guide"], hb) < 0.15:
            binding = "latency"       # neither roof within a factor of six: waves wait (dependent chains, barriers, occupancy)
    elif launch_ms < 0.02:
        binding = "launch"        # a launch of a few microseconds: neither roof is in sight
    elif "k_rdo_cand" in kname or "pixel" in kname or "x" in kname.split(":")[-1] and "fused" in kname:
        binding = "valu (not measured for this launch)"
    else:
        binding = "not measured"
    return dict(common, binding_roof=binding, valu=valu,
                binding_note={"valu": "VALU issue bound: see valu.frac_ubench_mix / frac_guide; HBM is not the limit",
                              "hbm": "HBM / cache bandwidth",
                              "launch": "the launch lasts microseconds: launch-bound, neither roof applies",
                              "latency": "neither the VALU-issue nor the HBM fraction reaches 0.15: the waves wait (dependent "
                              "chains, barriers, low occupancy); see valu.frac_guide and hbm.frac_by_counters",
                              "valu (not measured for this launch)": "a fused-candidate launch: the headline's instruction counters say "
                              "this kernel family is VALU issue bound; no instruction counters were taken for this launch",
                              "not measured": "no instruction counters for this launch; `frac` / `frac_unique` / `traffic` are what "
                              "is known"}[binding])


COMPACT_LIMIT = 4096   # bytes: the driver's parser lost the 27 KB line of round 5 (BENCH_r05.json.parsed = null)


def _num(v, nd=4):
    return round(float(v), nd) if isinstance(v, (int, float)) and not isinstance(v, bool) else v


def compact_record(res, detail_path=None):
    """The ONE line bench.py prints on stdout: the contract's keys as numbers, no prose, < COMPACT_LIMIT bytes.
    `res` is the long document (kept: stderr + `detail_path`).  Nothing here measures anything."""
    cfg, roof, cpu = res.get("config", {}), res.get("roofline") or {}, res.get("cpu_baseline")
    xok = cfg.get("exchange_ok")
    if isinstance(xok, dict):     # N > 1: the booleans of the tagged-tile checks (before the run: halo, gather; after it), not their notes
        xok = {k: xok.get(k) for k in ("halo", "gather", "after_run") if k in xok} or None
    valu = roof.get("valu") or {}
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                   "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": str(cfg.get("workload", ""))[:200],
                     "candidates_per_step": cfg.get("candidates_per_step"), "tiles": cfg.get("tiles"),
                     "exchange": str(cfg["exchange"])[:60] if cfg.get("exchange") else None, "exchange_ok": xok,
                     "parallelism": cfg.get("parallelism"), "compute_ms": cfg.get("compute_ms"),
                     "exchange_ms": cfg.get("exchange_ms")}
    out["roofline"] = {"bound": roof.get("bound"), "achieved": roof.get("achieved"), "peak": roof.get("peak"),
                       "unit": roof.get("unit"), "frac": roof.get("frac"), "traffic": roof.get("traffic"),
                       "kernel": roof.get("kernel"), "avg_launch_ms": roof.get("avg_launch_ms"),
                       "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
                       "frac_unique": (roof.get("hbm") or {}).get("frac_unique"),
                       "binding_roof": str(roof.get("binding_roof", ""))[:24],
                       "valu_frac": valu.get("frac_guide"), "valu_frac_ubench_mix": valu.get("frac_ubench_mix")}
    if cpu:
        out["cpu_baseline"] = {"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"),
                               "kind": cpu.get("kind"), "sample": str(cpu.get("sample", ""))[:100],
                               "impl": str(cpu.get("impl", ""))[:80],
                               "one_thread": (cpu.get("one_thread") or {}).get("value"),
                               "reference_asm_timed": bool((cpu.get("reference_probe") or {}).get("ran"))}
    for k in ("kernel_ms", "parity_checked", "parity_ok"):
        if k in res:
            out[k] = res[k]
    if "pipelined" in res:
        out["pipelined_value"] = res["pipelined"].get("value")
    if "rdo_only" in res:
        out["rdo_only"] = {"value": res["rdo_only"].get("value"), "ms_per_step": res["rdo_only"].get("ms_per_step")}
    lines = []
    for e in res.get("extra_lines", []):
        r = e.get("roofline") or {}
        v = r.get("valu") or {}
        c = {"name": e.get("name"), "value": _num(e.get("value"), 2), "unit": e.get("unit"),
             "ms_per_step": _num(e.get("ms_per_step")), "dtype": e.get("dtype", res.get("dtype")),
             "roofline_frac": r.get("frac"), "valu_frac": v.get("frac_guide", r.get("valu_frac")),
             "parity_ok": e.get("parity_ok")}
        if c["unit"] == res.get("unit"):      # said once at the top
            del c["unit"]
        if c["valu_frac"] is None:
            del c["valu_frac"]
        cb = e.get("cpu_baseline")
        if cb:
            c["cpu"] = _num(cb.get("value"), 1)
            c["cpu_cores"] = cb.get("cores")
        if isinstance(e.get("pipelined"), dict):
            c["pipelined"] = _num(e["pipelined"].get("value"), 2)
        if "stage_ms" in e:          # the config-4 frame: its stages as numbers
            c["stage_ms"] = {k[:28]: _num(v_, 3) for k, v_ in e["stage_ms"].items()}
        lines.append(c)
    out["extra_lines"] = lines
    if detail_path:
        out["detail"] = detail_path
    s = json.dumps(out, separators=(",", ":"))
    # belt and braces: drop the widest optional parts until the line fits
    for drop in ("stage_ms", "cpu_cores", "valu_frac", "unit"):
        if len(s) < COMPACT_LIMIT:
            break
        for c in lines:
            c.pop(drop, None)
        s = json.dumps(out, separators=(",", ":"))
    if len(s) >= COMPACT_LIMIT:
        out["extra_lines"] = [{"name": c["name"], "value": c["value"], "parity_ok": c["parity_ok"]} for c in lines]
        s = json.dumps(out, separators=(",", ":"))
    if len(s) >= COMPACT_LIMIT:
        out.pop("extra_lines")
        s = json.dumps(out, separators=(",", ":"))
    return s


def emit(res, detail_out=None):
    """Long document -> stderr and `detail_out` (default gpurun_out/bench_detail.json, scratch); compact record
    -> stdout, the last and ONLY stdout line."""
    detail_path = detail_out or os.path.join("gpurun_out", "bench_detail.json")
    try:
        full = os.path.join(ROOT, detail_path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "w") as f:
            json.dump(res, f)
    except OSError:
        detail_path = None
    for e in res.get("extra_lines", []):
        print("extra_line " + json.dumps(e), file=sys.stderr)
    print("detail " + json.dumps({k: v for k, v in res.items() if k != "extra_lines"}), file=sys.stderr)
    sys.stderr.flush()
    print(compact_record(res, detail_path))
    sys.stdout.flush()


def extra_lines(ctx, args):
    """SURVEY 8(d) configs 3 / 4 beside the headline (same frame size, bit depth and K): the two
    4:2:0 chroma planes at half size with the 32/16/8/4 ladder, the nine 8-tap filter pairs mixed
    per candidate, and the RDO transform types mixed per candidate.  Each line: a few timed steps,
    HIP events per launch for its own roofline, and a strided sample of every launch checked
    against the scalar oracle.  (The 10-bit config-4 proxy is `--chain pixel --bit-depth 10`.)"""
    import torch
    from rav1e_amd import workload as W
    from rav1e_amd.api import Plane
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = O.lib()
    L.r1o_set_threads(os.cpu_count() or 1)
    fw, fh, bd, k = args.width, args.height, args.bit_depth, args.k
    bpp = 1 if bd == 8 else 2
    ct, tct = (np.int16, torch.int16) if bpp == 1 else (np.int32, torch.int32)
    specs = [
        ("chroma_420", "the U and V planes of 4:2:0 (%dx%d each), ladder 32/16/8/4, MV +-16 px, REGULAR, DCT_DCT",
         dict(planes=2, w=fw // 2, h=fh // 2, pad=44, sizes=(32, 16, 8, 4), mv=16, kw={})),
        ("filter_pairs", "luma %dx%d, ladder 64/32/16/8, one of the 9 REGULAR/SMOOTH/SHARP pairs per candidate",
         dict(planes=1, w=fw, h=fh, pad=88, sizes=W.LADDER, mv=32, kw={"mix_filters": True})),
        ("tx_types", "luma %dx%d, ladder 64/32/16/8, one of the RDO transform types the size admits per "
                     "candidate (7 up to 16x16, DCT/IDTX at 32, DCT at 64)",
         dict(planes=1, w=fw, h=fh, pad=88, sizes=W.LADDER, mv=32, kw={"mix_tx_types": True})),
    ]
    STEPS, WARM, NCHK = 20, 2, 48
    lines = []
    for name, desc, sp in specs:
        w, h, pad = sp["w"], sp["h"], sp["pad"]
        cands = W.speed6_ladder(w, h, k, seed=7, mv_range=sp["mv"], sizes=sp["sizes"], **sp["kw"])
        dc = {s: torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda() for s, c in cands.items()}
        planes, launches, outs = [], [], []
        for p in range(sp["planes"]):
            ho = W.random_plane_array(w, h, bd, 21 + 2 * p, pad, pad)
            hr = W.random_plane_array(w, h, bd, 22 + 2 * p, pad, pad)
            po, pr = Plane.from_numpy(ho, w, h, bd, pad, pad), Plane.from_numpy(hr, w, h, bd, pad, pad)
            planes.append((ho, hr, po, pr))
            for s, c in cands.items():
                n = len(c)
                o = {"sad": torch.empty(n, dtype=torch.int32, device="cuda"),
                     "satd": torch.empty(n, dtype=torch.int32, device="cuda"),
                     "coeffs": torch.empty((n, s * s), dtype=tct, device="cuda")}
                outs.append((p, s, o))
                launches.append((s, ctx.prepare_rdo_cand(po, pr, s, s, dc[s], n, o)))
        ev = {s: [] for s in cands}
        for _ in range(WARM):
            for _, f in launches:
                f()
        torch.cuda.synchronize()
        # the CPU parity leg of the line before let the clocks drop: without this the 64x64 launch of the
        # tx_types line (DCT only at that size -- the headline's own launch) read 0.277 ms instead of 0.233
        sustain(lambda: [f() for _, f in launches], args.prewarm_ms)
        t0 = time.perf_counter()
        for _ in range(STEPS):
            for s, f in launches:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                f()
                e1.record()
                ev[s].append((e0, e1))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS
        per = {s: sum(a.elapsed_time(b) for a, b in v) / len(v) for s, v in ev.items()}
        dom = max(per, key=lambda s: per[s])
        px = sp["planes"] * sum(len(c) * s * s for s, c in cands.items())
        abytes = W.algorithmic_bytes_per_cand(dom, dom, bpp) * len(cands[dom])
        # parity: NCHK strided candidates of every launch against the scalar oracle
        n_chk, bad = 0, []
        for p, s, o in outs:
            ho, hr = planes[p][0], planes[p][1]
            a, b = O.HostPlane(w, h, bd, pad, pad), O.HostPlane(w, h, bd, pad, pad)
            a.data, b.data = ho, hr
            pa, pb = a.cstruct(), b.cstruct()
            c = cands[s]
            idx = np.arange(0, len(c), max(1, len(c) // NCHK))[:NCHK]
            sub = np.ascontiguousarray(c[idx])
            sad, satd = np.zeros(len(sub), np.uint32), np.zeros(len(sub), np.uint32)
            co = np.zeros((len(sub), s * s), ct)
            ts = {64: 4, 32: 3, 16: 2, 8: 1, 4: 0}[s]
            rc = L.r1o_rdo_cand_batch(C.byref(pa), C.byref(pb), s, s, ts, O.ptr(sub), len(sub),
                                      O.ptr(sad), O.ptr(satd), O.ptr(co), None)
            ix = torch.from_numpy(idx.astype(np.int64)).cuda()
            ok = (rc == 0
                  and np.array_equal(o["sad"].index_select(0, ix).cpu().numpy().view(np.uint32), sad)
                  and np.array_equal(o["satd"].index_select(0, ix).cpu().numpy().view(np.uint32), satd)
                  and np.array_equal(o["coeffs"].index_select(0, ix).cpu().numpy(), co))
            n_chk += len(idx)
            if not ok:
                bad.append("plane %d %dx%d" % (p, s, s))
        lc, lsrc = launch_pmc(rdo_launch_key(bd, dom, 0, len(cands[dom])))
        roof = build_roofline("k_rdo_cand<bd=%d,%dx%d> (%s)" % (bd, dom, dom, name), abytes,
                              per[dom], None, None, None,
                              working_set=planes[0][0].nbytes + planes[0][1].nbytes,
                              write_bytes=len(cands[dom]) * (8 + dom * dom * (2 if bpp == 1 else 4)),
                              line_counters=lc, line_src=lsrc)
        lines.append({"name": name, "metric": "RDO-candidate Mpixels/s (dist+fwd_tx+mc)",
                      "value": round(px / dt / 1e6, 2), "unit": "Mpixels/s", "steps": STEPS,
                      "ms_per_step": round(dt * 1e3, 4),
                      "config": {"workload": desc % (w, h), "bit_depth": bd, "k": k,
                                 "candidates_per_step": sp["planes"] * int(sum(len(c) for c in cands.values()))},
                      "kernel_ms": {str(s): round(v, 4) for s, v in per.items()},
                      "kernel_ms_note": "per launch; timed with events around every launch, so the step "
                                        "time carries the event overhead the headline avoids",
                      "roofline": roof, "parity_checked": n_chk, "parity_ok": not bad, "parity_bad": bad})
        del planes, launches, outs, dc
        torch.cuda.empty_cache()
    return lines


def config_lines(ctx, args):
    """BASELINE.json configs 2-4 in the driver-run record, a few timed launches each,
    HIP events per launch, algorithmic bytes of SURVEY 8(d), a strided sample of every launch
    checked against the scalar oracle:
      config2_dist_1080p      1920x1080 8-bit: get_sad (K = 32) + get_satd (K = 8) over the ladder
                              (benches/dist.rs), r1_dist_batch
      config2_fwd_dct_1080p   forward DCT_DCT of every transform block of a 1080p frame at 64/32/16/8
                              (benches/transform.rs), r1_fwd_txfm_batch
      config3_mc_1080p        put_8tap + prep_8tap over the ladder, random 1/16-pel fractions
                              (benches/mc.rs), r1_mc_put_batch / r1_mc_prep_batch
      fused_4k_10bit          the headline step on 3840x2160 10-bit planes
      config4_proxy_4k_10bit  4K 10-bit: the pixel-domain candidate chain (-> quantize -> inverse ->
                              cdef_dist) + the CDEF luma pass + the CDEF strength search (8 presets)"""
    import torch
    from rav1e_amd import workload as W
    from rav1e_amd.api import DIST_CAND, MC_CAND, Plane
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = O.lib()
    L.r1o_set_threads(os.cpu_count() or 1)
    NCHK, REPS, WARM = 48, 20, 5
    lines = []

    launch_how = []

    def timed(fns, graph_ok=True):
        """fns: [(tag, callable)] -> ({tag: ms per launch}, ms per pass)"""
        for _ in range(WARM):          # clocks ramp over the first passes: time the sustained state
            for _, f in fns:
                f()
        torch.cuda.synchronize()
        sustain(lambda: [f() for _, f in fns], args.prewarm_ms)
        ev = []
        t0 = time.perf_counter()
        for _ in range(REPS):
            for tag, f in fns:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                f()
                e1.record()
                ev.append((tag, e0, e1))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / REPS * 1e3
        per = {}
        for tag, a, b in ev:
            per.setdefault(tag, []).append(a.elapsed_time(b))
        per = {t: sum(v) / len(v) for t, v in per.items()}
        # A pass of these lines is 2-16 launches of a few microseconds each: issued one by one the host
        # (ctypes call + launch) is what is timed.  The same launches captured once in a hipGraph and
        # replayed are the pass as a resident encoder would run it; entry points that synchronise or
        # allocate (the composite CDEF call's scratch ring) cannot be captured and keep the plain figure.
        how = "one call per launch"
        if graph_ok:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side):
                        for _, f in fns:
                            f()
                    for _ in range(WARM):
                        g.replay()
                    side.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(REPS):
                        g.replay()
                    side.synchronize()
                    dtg = (time.perf_counter() - t0) / REPS * 1e3
                torch.cuda.current_stream().wait_stream(side)
                if dtg < dt:
                    dt, how = dtg, "hipGraph replay of the pass (captured once)"
            except Exception as e:      # capture refused: keep the plain pass
                how = "one call per launch (graph capture refused: %s)" % str(e).splitlines()[0][:80]
                torch.cuda.synchronize()
        launch_how.append(how)
        return per, dt

    def line(name, desc, px, per, step_ms, abytes_by_tag, n_chk, bad, extra=None, working_set=None, wbytes=None,
             kkeys=None):
        dom = max(per, key=lambda t: per[t])
        lc, lsrc = launch_pmc((kkeys or {}).get(dom))
        roof = build_roofline("%s: %s" % (name, dom), abytes_by_tag[dom], per[dom], None, None, None,
                              working_set=working_set, write_bytes=(wbytes or {}).get(dom),
                              line_counters=lc, line_src=lsrc)
        d = {"name": name, "metric": "Mpixels/s", "value": round(px / (step_ms * 1e-3) / 1e6, 2), "unit": "Mpixels/s",
             "steps": REPS, "ms_per_step": round(step_ms, 4),
             "config": {"workload": desc, "launch": launch_how[-1] if launch_how else None},
             "kernel_ms": {str(t): round(v, 4) for t, v in per.items()}, "roofline": roof,
             "parity_checked": n_chk, "parity_ok": not bad, "parity_bad": bad}
        if extra:
            d.update(extra)
        lines.append(d)

    def sample(n):
        return np.arange(0, n, max(1, n // NCHK))[:NCHK]

    # ---------------- 1080p 8-bit planes (configs 2 and 3)
    w, h, bd = 1920, 1080, 8
    ho, hr = W.random_plane_array(w, h, bd, 1), W.random_plane_array(w, h, bd, 2)
    po, pr = Plane.from_numpy(ho, w, h, bd, 88, 88), Plane.from_numpy(hr, w, h, bd, 88, 88)
    a, b = O.HostPlane(w, h, bd), O.HostPlane(w, h, bd)
    a.data, b.data = ho, hr
    pa, pb = a.cstruct(), b.cstruct()
    ws_1080 = ho.nbytes + hr.nbytes
    # config 2: SAD K = 32, SATD K = 8
    fns, abytes, px, chk, wbytes, kkeys = [], {}, 0, [], {}, {}
    for kind, k, nm in ((0, 32, "sad"), (1, 8, "satd")):
        cands = W.speed6_ladder(w, h, k, seed=3, mv_range=32)
        for sz, c in cands.items():
            dcand = np.zeros(len(c), DIST_CAND)
            for f in ("ox", "oy", "rx", "ry"):
                dcand[f] = c[f]
            dev = torch.from_numpy(dcand.view(np.uint8).reshape(-1).copy()).cuda()
            out = torch.empty(len(c), dtype=torch.int32, device="cuda")
            tag = "%s %dx%d" % (nm, sz, sz)
            fns.append((tag, lambda kind=kind, sz=sz, dev=dev, n=len(c), out=out: ctx.dist_batch(kind, po, pr, sz, sz, dev, n=n, out=out)))
            abytes[tag] = (2 * sz * sz + 4) * len(c)
            wbytes[tag] = 4 * len(c)
            kkeys[tag] = ("k_dist<1,8,%s>" % ("true" if kind else "false"),
                          ((((len(c) * (sz // 8) ** 2 + 255) // 256) + 7) & ~7) * 256)
            px += len(c) * sz * sz
            chk.append((kind, sz, dcand, out))
    per, step = timed(fns)
    n_chk, bad = 0, []
    for kind, sz, dcand, out in chk:
        idx = sample(len(dcand))
        sub = np.ascontiguousarray(dcand[idx])
        want = np.zeros(len(sub), np.uint32)
        assert L.r1o_dist_batch(kind, C.byref(pa), C.byref(pb), sz, sz, O.ptr(sub), len(sub), O.ptr(want)) == 0
        got = out.index_select(0, torch.from_numpy(idx.astype(np.int64)).cuda()).cpu().numpy().view(np.uint32)
        n_chk += len(idx)
        if not np.array_equal(got, want):
            bad.append("%s %d" % ("satd" if kind else "sad", sz))
    line("config2_dist_1080p", "1920x1080 8-bit luma, speed-6 ladder 64/32/16/8: get_sad K=32 + get_satd K=8 per block, "
         "MV +-32 px (benches/dist.rs)", px, per, step, abytes, n_chk, bad, working_set=ws_1080, wbytes=wbytes, kkeys=kkeys)
    # config 2: forward DCT of every transform block of the frame
    fns, abytes, px, chk, wbytes, kkeys = [], {}, 0, [], {}, {}
    from rav1e_amd.types import TxSize
    rng = np.random.default_rng(5)
    for sz in W.LADDER:
        nb = (w // sz) * (h // sz)
        res = rng.integers(-255, 256, (nb, sz, sz)).astype(np.int16)
        dres = torch.from_numpy(res).cuda()
        out = torch.empty((nb, sz * sz), dtype=torch.int16, device="cuda")
        ts = int(TxSize.by_dims(sz, sz))
        tag = "fdct %dx%d" % (sz, sz)
        fns.append((tag, lambda dres=dres, ts=ts, out=out: ctx.forward_transform_batch(dres, ts, 0, 8, out=out)))
        abytes[tag] = nb * (2 * sz * sz + 2 * sz * sz)
        wbytes[tag] = nb * 2 * sz * sz
        lg_ = sz.bit_length() - 1
        kkeys[tag] = ("k_fwd_tx<%d,%d,short>" % (lg_, lg_), ((nb + 64 // sz - 1) // (64 // sz)) * 64)
        px += nb * sz * sz
        chk.append((sz, ts, res, out))
    per, step = timed(fns)
    n_chk, bad = 0, []
    for sz, ts, res, out in chk:
        idx = sample(len(res))
        sub = np.ascontiguousarray(res[idx])
        want = np.zeros((len(sub), sz * sz), np.int16)
        assert L.r1o_fwd_txfm_batch(O.ptr(sub), O.ptr(want), len(sub), ts, 0, 8, 2) == 0
        got = out.index_select(0, torch.from_numpy(idx.astype(np.int64)).cuda()).cpu().numpy()
        n_chk += len(idx)
        if not np.array_equal(got, want):
            bad.append("fdct %d" % sz)
    line("config2_fwd_dct_1080p", "every transform block of a 1920x1080 frame at 64/32/16/8, DCT_DCT, residual uniform "
         "[-255, 255] (benches/transform.rs)", px, per, step, abytes, n_chk, bad,
         # the residual blocks ARE the input here: every byte is read once per pass
         working_set=sum(v for v in wbytes.values()), wbytes=wbytes, kkeys=kkeys)
    # config 3: put_8tap / prep_8tap
    fns, abytes, px, chk, wbytes, kkeys = [], {}, 0, [], {}, {}
    cands = W.speed6_ladder(w, h, 8, seed=4, mv_range=32)
    for sz, c in cands.items():
        mc = np.zeros(len(c), MC_CAND)
        for f in ("rx", "ry", "col_frac", "row_frac", "mode_x", "mode_y"):
            mc[f] = c[f]
        dev = torch.from_numpy(mc.view(np.uint8).reshape(-1).copy()).cuda()
        o_put = torch.empty((len(c), sz, sz), dtype=torch.uint8, device="cuda")
        o_prep = torch.empty((len(c), sz, sz), dtype=torch.int16, device="cuda")
        fns.append(("put %dx%d" % (sz, sz), lambda sz=sz, dev=dev, n=len(c), o=o_put: ctx.put_8tap_batch(pr, sz, sz, dev, n=n, out=o)))
        fns.append(("prep %dx%d" % (sz, sz), lambda sz=sz, dev=dev, n=len(c), o=o_prep: ctx.prep_8tap_batch(pr, sz, sz, dev, n=n, out=o)))
        abytes["put %dx%d" % (sz, sz)] = len(c) * ((sz + 7) * (sz + 7) + sz * sz)
        abytes["prep %dx%d" % (sz, sz)] = len(c) * ((sz + 7) * (sz + 7) + 2 * sz * sz)
        wbytes["put %dx%d" % (sz, sz)], wbytes["prep %dx%d" % (sz, sz)] = len(c) * sz * sz, 2 * len(c) * sz * sz
        lg_ = sz.bit_length() - 1
        for nm_, pf_ in (("put", "false"), ("prep", "true")):
            kkeys["%s %dx%d" % (nm_, sz, sz)] = ("k_mc_fast<1,%d,%d,%s>" % (lg_, lg_, pf_),
                                                ((((len(c) + 64 // sz - 1) // (64 // sz)) + 7) & ~7) * 64)
        px += 2 * len(c) * sz * sz
        chk.append((sz, mc, o_put, o_prep))
    per, step = timed(fns)
    n_chk, bad = 0, []
    for sz, mc, o_put, o_prep in chk:
        idx = sample(len(mc))
        sub = np.ascontiguousarray(mc[idx])
        w_put, w_prep = np.zeros((len(sub), sz, sz), np.uint8), np.zeros((len(sub), sz, sz), np.int16)
        assert L.r1o_mc_put_batch(C.byref(pb), sz, sz, O.ptr(sub), len(sub), O.ptr(w_put)) == 0
        assert L.r1o_mc_prep_batch(C.byref(pb), sz, sz, O.ptr(sub), len(sub), O.ptr(w_prep)) == 0
        ix = torch.from_numpy(idx.astype(np.int64)).cuda()
        n_chk += 2 * len(idx)
        if not (np.array_equal(o_put.index_select(0, ix).cpu().numpy(), w_put) and
                np.array_equal(o_prep.index_select(0, ix).cpu().numpy(), w_prep)):
            bad.append("mc %d" % sz)
    line("config3_mc_1080p", "1920x1080 8-bit luma, ladder 64/32/16/8, K=8: put_8tap + prep_8tap REGULAR with random "
         "1/16-pel fractions, MV +-32 px (benches/mc.rs)", px, per, step, abytes, n_chk, bad,
         working_set=hr.nbytes, wbytes=wbytes, kkeys=kkeys)
    # config 3: intra prediction (benches/predict.rs: edges uniform 0..255; here every transform block of the frame at
    # 64/32/16/8/4, the 13 luma modes mixed per block -- DC variants, V, H, the six directional modes at their base
    # angles with the edge filter on, the three smooth modes, Paeth)
    from rav1e_amd.api import INTRA_CAND
    fns, abytes, px, chk, wbytes, kkeys = [], {}, 0, [], {}, {}
    rng = np.random.default_rng(6)
    for sz in (64, 32, 16, 8, 4):
        nb = (w // sz) * (h // sz)
        ts = int(TxSize.by_dims(sz, sz))
        edges = np.zeros((nb, 257), np.uint8)
        il = ia = min(2 * sz, 128)
        edges[:, 128 - il:129 + ia] = rng.integers(0, 256, (nb, il + ia + 1))
        lens = np.tile(np.array([il, ia], np.uint8), (nb, 1))
        ic = np.zeros(nb, INTRA_CAND)
        pm = rng.integers(0, 13, nb)
        bx, by = np.arange(nb) % (w // sz), np.arange(nb) // (w // sz)
        var = np.where((bx == 0) & (by == 0), 0, np.where(by == 0, 1, np.where(bx == 0, 2, 3)))
        pm = np.where((pm == 12) & (var == 0), 0, np.where((pm == 12) & (var == 2), 1, np.where((pm == 12) & (var == 1), 2, pm)))
        ic["mode"], ic["variant"] = pm, var
        ic["angle"] = np.array([0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0])[pm]
        ic["ief"] = np.where((pm >= 1) & (pm <= 8), 1, 0)
        ic["avail_w"] = ic["avail_h"] = sz
        dic = torch.from_numpy(ic.view(np.uint8).reshape(-1).copy()).cuda()
        de, dl = torch.from_numpy(edges).cuda(), torch.from_numpy(lens).cuda()
        tag = "predict %dx%d" % (sz, sz)
        lg_ = sz.bit_length() - 1
        kkeys[tag] = ("k_intra_predict<1,false,%d,%d>" % ((lg_, lg_) if sz <= 32 else (-1, -1)), None)
        keep = {}
        fns.append((tag, lambda ts=ts, dic=dic, de=de, dl=dl, nb=nb, keep=keep: keep.__setitem__("o", ctx.predict_intra_batch(ts, dic, de, dl, 8, n=nb))))
        abytes[tag] = nb * ((2 * (2 * sz) + 1) + sz * sz)          # (2 (W + H) + 1) bpp read + W H bpp write
        wbytes[tag] = nb * sz * sz
        px += nb * sz * sz
        chk.append((sz, ts, ic, edges, keep))
    per, step = timed(fns, graph_ok=False)      # the call allocates its output: not capturable
    n_chk, bad = 0, []
    for sz, ts, ic, edges, keep in chk:
        got = keep["o"].cpu().numpy()
        for i in sample(len(ic)):
            out = np.zeros((sz, sz), np.uint8)
            il = ia = min(2 * sz, 128)
            assert L.r1o_dispatch_predict_intra(int(ic["mode"][i]), int(ic["variant"][i]), O.ptr(out), sz, ts, 8, None,
                                                int(ic["angle"][i]), int(ic["ief"][i]), O.ptr(edges[i]), il, ia, sz, sz, 0) == 0
            n_chk += 1
            if not np.array_equal(got[i], out):
                bad.append("predict %d mode %d" % (sz, int(ic["mode"][i])))
                break
    line("config3_predict_1080p", "every transform block of a 1920x1080 8-bit frame at 64/32/16/8/4: dispatch_predict_intra, the 13 "
         "luma modes mixed per block, edges uniform 0..255 (benches/predict.rs)", px, per, step, abytes, n_chk, bad,
         working_set=sum(v for v in wbytes.values()), wbytes=wbytes, kkeys=kkeys)
    del po, pr, fns, chk
    torch.cuda.empty_cache()

    # ---------------- 4K 10-bit: the fused candidate and the config-4 proxy
    w, h, bd, k = args.width, args.height, 10, args.k
    ho, hr = W.random_plane_array(w, h, bd, 1), W.random_plane_array(w, h, bd, 2)
    po, pr = Plane.from_numpy(ho, w, h, bd, 88, 88), Plane.from_numpy(hr, w, h, bd, 88, 88)
    a, b = O.HostPlane(w, h, bd), O.HostPlane(w, h, bd)
    a.data, b.data = ho, hr
    pa, pb = a.cstruct(), b.cstruct()
    cands = W.speed6_ladder(w, h, k)
    dc = {s: torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda() for s, c in cands.items()}
    TS = {64: 4, 32: 3, 16: 2, 8: 1}
    fns, abytes, px, outs = [], {}, 0, {}
    for s, c in cands.items():
        n = len(c)
        outs[s] = {"sad": torch.empty(n, dtype=torch.int32, device="cuda"), "satd": torch.empty(n, dtype=torch.int32, device="cuda"),
                   "coeffs": torch.empty((n, s * s), dtype=torch.int32, device="cuda")}
        fns.append(("%dx%d" % (s, s), ctx.prepare_rdo_cand(po, pr, s, s, dc[s], n, outs[s])))
        abytes["%dx%d" % (s, s)] = W.algorithmic_bytes_per_cand(s, s, 2) * n
        px += n * s * s
    per, step = timed(fns)
    n_chk, bad = 0, []
    for s, c in cands.items():
        idx = sample(len(c))
        sub = np.ascontiguousarray(c[idx])
        sad, satd = np.zeros(len(sub), np.uint32), np.zeros(len(sub), np.uint32)
        co = np.zeros((len(sub), s * s), np.int32)
        assert L.r1o_rdo_cand_batch(C.byref(pa), C.byref(pb), s, s, TS[s], O.ptr(sub), len(sub), O.ptr(sad), O.ptr(satd),
                                    O.ptr(co), None) == 0
        ix = torch.from_numpy(idx.astype(np.int64)).cuda()
        n_chk += len(idx)
        if not (np.array_equal(outs[s]["sad"].index_select(0, ix).cpu().numpy().view(np.uint32), sad) and
                np.array_equal(outs[s]["satd"].index_select(0, ix).cpu().numpy().view(np.uint32), satd) and
                np.array_equal(outs[s]["coeffs"].index_select(0, ix).cpu().numpy(), co)):
            bad.append("fused %d" % s)
    line("fused_4k_10bit", "%dx%d 10-bit luma, speed-6 ladder, K=%d fused candidates (the headline step at the config-4 "
         "pixel format)" % (w, h, k), px, per, step, abytes, n_chk, bad, {"dtype": "u16"},
         working_set=ho.nbytes + hr.nbytes,
         wbytes={"%dx%d" % (s_, s_): len(c_) * (8 + 4 * s_ * s_) for s_, c_ in cands.items()},
         kkeys={"%dx%d" % (s_, s_): rdo_launch_key(10, s_, 0, len(c_)) for s_, c_ in cands.items()})
    del outs, fns
    torch.cuda.empty_cache()
    # config-4 proxy: pixel-domain chain + CDEF luma pass + CDEF strength search
    scales = torch.from_numpy(np.random.default_rng(9).integers(1 << 12, 1 << 16, ((h + 7) // 8, (w + 7) // 8)).astype(np.int32)).cuda()
    fns, abytes, px, pouts = [], {}, 0, {}
    for s, c in cands.items():
        n = len(c)
        pouts[s] = {"eob": torch.empty(n, dtype=torch.int16, device="cuda"), "dist": torch.empty(n, dtype=torch.int64, device="cuda"),
                    "sad": torch.empty(n, dtype=torch.int32, device="cuda"), "satd": torch.empty(n, dtype=torch.int32, device="cuda")}
        fns.append(("pixel %dx%d" % (s, s), lambda s=s, n=n: ctx.rdo_pixel_cand_batch(po, pr, s, s, dc[s], args.qindex, 3, scales=scales,
                                                                                         n=n, outs=pouts[s])))
        abytes["pixel %dx%d" % (s, s)] = (2 * ((s + 7) * (s + 7) + 2 * s * s) + 18) * n
        px += n * s * s
    # CDEF over the luma plane (every 8x8 block coded, strengths of rav1e's preset 2) and the strength search
    cw, chh = w // 2, h // 2
    chroma = [Plane.from_numpy(W.random_plane_array(cw, chh, bd, 30 + i, 44, 44), cw, chh, bd, 44, 44) for i in range(4)]
    rec3, src3 = [pr, chroma[0], chroma[2]], [po, chroma[1], chroma[3]]
    dst = Plane(w, h, bd)
    skip = torch.zeros((h // 4, w // 4), dtype=torch.uint8, device="cuda")
    ci = torch.zeros(((h + 63) // 64, (w + 63) // 64), dtype=torch.uint8, device="cuda")
    skip_s = torch.zeros((2 * ((h + 7) // 8), 2 * ((w + 7) // 8)), dtype=torch.uint8, device="cuda")
    presets = [0, 4, 9, 13, 22, 31, 43, 55]    # fi.cdef_y_strengths, src/encoder.rs:897-916
    fns.append(("cdef luma pass", lambda: ctx.cdef_filter_frame_plane(pr, pr, dst, 0, 0, 0, w, h, skip, ci, [36] * 8, [36] * 8, 5, bd)))
    fns.append(("cdef strength search", lambda: ctx.cdef_strength_search(rec3, src3, skip_s, presets, presets, 5, bd, 8, 1, 1, w, h,
                                                                          scales=scales)))
    abytes["cdef luma pass"] = 2 * w * h * 2
    abytes["cdef strength search"] = 2 * (w * h + 2 * cw * chh) * 2

    def cdef_chk(L):
        """the two CDEF entry points against the oracle on a 512 x 256 frame of the same kind (the
        whole-frame comparison at 4K is tests/test_gpu_fullsize.py)"""
        sw, sh = 512, 256
        rng = np.random.default_rng(77)
        imgs = [rng.integers(0, 1 << bd, ((sh >> (1 if i else 0)), (sw >> (1 if i else 0)))) for i in range(3)]
        recs = [np.clip(im + rng.integers(-20, 21, im.shape), 0, (1 << bd) - 1) for im in imgs]
        hp_s = [O.plane_from_image(im, bd, 16, 16) for im in imgs]
        hp_r = [O.plane_from_image(im, bd, 16, 16) for im in recs]
        dev = lambda hp: Plane.from_numpy(hp.data, hp.width, hp.height, bd, hp.xpad, hp.ypad)
        d_s, d_r = [dev(x) for x in hp_s], [dev(x) for x in hp_r]
        mi_cols, mi_rows = 2 * ((sw + 7) // 8), 2 * ((sh + 7) // 8)
        sk = np.zeros((mi_rows, mi_cols), np.uint8)
        sc = rng.integers(1 << 12, 1 << 16, ((sh + 7) // 8, (sw + 7) // 8)).astype(np.uint32)
        prm = O.CdefSearchParams()
        prm.y_strengths[:] = presets
        prm.uv_strengths[:] = presets
        prm.damping, prm.bit_depth, prm.n_idx, prm.planes = 5, bd, 8, 3
        prm.xdec, prm.ydec, prm.crop_w, prm.crop_h, prm.area_sb_w, prm.area_sb_h = 1, 1, sw, sh, 1, 1
        prm.dist_scale[:] = [1 << 14, 1 << 14, 1 << 14]
        n_sbx, n_sby = (mi_cols + 15) // 16, (mi_rows + 15) // 16
        want_err, want_best = np.zeros((n_sby, n_sbx, 8), np.uint64), np.zeros((n_sby, n_sbx), np.int8)
        pr3 = (O.Plane * 3)(*[x.cstruct() for x in hp_r])
        ps3 = (O.Plane * 3)(*[x.cstruct() for x in hp_s])
        assert L.r1o_cdef_strength_search(pr3, ps3, sk.ctypes.data, mi_cols, mi_cols, mi_rows, sc.ctypes.data, sc.shape[1],
                                          C.byref(prm), want_err.ctypes.data, want_best.ctypes.data) == 0
        err, best = ctx.cdef_strength_search(d_r, d_s, torch.from_numpy(sk).cuda(), presets, presets, 5, bd, 8, 1, 1, sw, sh,
                                             scales=torch.from_numpy(sc.view(np.int32)).cuda())
        ok = (np.array_equal(err.cpu().numpy().view(np.uint64), want_err) and np.array_equal(best.cpu().numpy(), want_best))
        return n_sbx * n_sby * 8, ([] if ok else ["cdef strength search"])
    per, step = timed(fns, graph_ok=False)   # the composite CDEF call waits on its scratch ring: not capturable
    n_chk, bad = 0, []
    for s, c in cands.items():
        idx = sample(len(c))[:16]
        sub = np.ascontiguousarray(c[idx])
        sad, satd = np.zeros(len(sub), np.uint32), np.zeros(len(sub), np.uint32)
        eob, dist = np.zeros(len(sub), np.uint16), np.zeros(len(sub), np.uint64)
        hs = scales.cpu().numpy().view(np.uint32)
        assert L.r1o_rdo_pixel_cand_batch(C.byref(pa), C.byref(pb), s, s, TS[s], O.ptr(sub), len(sub), args.qindex, 0, 0, 0, 3,
                                          O.ptr(hs), hs.shape[1], 0, 0, O.ptr(sad), O.ptr(satd), O.ptr(eob), O.ptr(dist),
                                          None, None, None) == 0
        ix = torch.from_numpy(idx.astype(np.int64)).cuda()
        n_chk += len(idx)
        if not (np.array_equal(pouts[s]["dist"].index_select(0, ix).cpu().numpy().view(np.uint64), dist) and
                np.array_equal(pouts[s]["eob"].index_select(0, ix).cpu().numpy().view(np.uint16), eob)):
            bad.append("pixel %d" % s)
    c_n, c_bad = cdef_chk(L)
    n_chk += c_n
    bad += c_bad
    # CPU beside it: the scalar oracle of the same chain (oracle/batch.c, OpenMP over candidates) on a strided sample of
    # every launch, sized for a few seconds on the host cores the bench may use
    proxy_cpu = None
    if args.cpu_seconds > 0:
        cores, _, _ = physical_cores()
        L.r1o_set_threads(cores)
        hs = scales.cpu().numpy().view(np.uint32)

        def cpu_chain(frac):
            t_px, t0_ = 0, time.perf_counter()
            for s_, c_ in cands.items():
                sub = np.ascontiguousarray(c_[::max(1, int(round(1 / frac)))])
                eob, dist = np.zeros(len(sub), np.uint16), np.zeros(len(sub), np.uint64)
                assert L.r1o_rdo_pixel_cand_batch(C.byref(pa), C.byref(pb), s_, s_, TS[s_], O.ptr(sub), len(sub), args.qindex, 0, 0, 0,
                                                  3, O.ptr(hs), hs.shape[1], 0, 0, None, None, O.ptr(eob), O.ptr(dist), None,
                                                  None, None) == 0
                t_px += len(sub) * s_ * s_
            return t_px, time.perf_counter() - t0_
        p0, d0 = cpu_chain(1 / 256)                                   # calibrate
        want_s = min(6.0, max(2.0, args.cpu_seconds / 3))
        frac = min(1.0, max(1 / 256, (1 / 256) * want_s / max(d0, 1e-3)))
        p1, d1 = cpu_chain(frac)
        proxy_cpu = {"value": round(p1 / d1 / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                     "sample": "every %dth candidate of every ladder size, %.1f s" % (max(1, int(round(1 / frac))), d1),
                     "impl": "oracle/batch.c r1o_rdo_pixel_cand_batch: scalar C restatement, one OpenMP thread per core"}
        L.r1o_set_threads(os.cpu_count() or 1)
    line("config4_proxy_4k_10bit", "%dx%d 10-bit: pixel-domain candidate chain over the ladder (K=%d: mc -> dist -> fwd -> quantize "
         "-> inverse -> cdef_dist) + CDEF luma pass + CDEF strength search over rav1e's 8 presets (4:2:0)" % (w, h, k),
         px, per, step, abytes, n_chk, bad, {"dtype": "u16", "px_note": "Mpixels/s counts the candidate pixels of the chain; the "
                                             "two CDEF launches are inside the step time", "cpu_baseline": proxy_cpu},
         working_set=ho.nbytes + hr.nbytes,
         wbytes={"pixel %dx%d" % (s_, s_): 18 * len(c_) for s_, c_ in cands.items()},
         kkeys={"pixel %dx%d" % (s_, s_): rdo_launch_key(10, s_, 2, len(c_)) for s_, c_ in cands.items()})
    del pouts, fns
    torch.cuda.empty_cache()
    # ---------------- the transform-type search (rdo_tx_type_decision): one prediction per block, every TxType of
    # RAV1E_TX_TYPES the tx set allows, pixel-domain distortion (rav1e's default) -- the fan-out launch, and beside it
    # the same evaluations as independent single-type candidates (what r1_rdo_pixel_cand_batch alone would cost)
    from rav1e_amd import rdo_glue as RG
    fns, fns_ind, abytes, px, touts, tcands = [], [], {}, 0, {}, {}
    for s in (32, 16, 8):
        c1 = np.ascontiguousarray(cands[s][::k])               # one prediction per block of the size
        n = len(c1)
        mask = ctx.tx_type_mask(TS[s], True)
        types = RG.tx_type_slots(mask)
        nt = len(types)
        tcands[s] = (c1, mask, types)
        dev1 = torch.from_numpy(c1.view(np.uint8).reshape(-1).copy()).cuda()
        touts[s] = {"eob": torch.empty((n, nt), dtype=torch.int16, device="cuda"),
                    "dist": torch.empty((n, nt), dtype=torch.int64, device="cuda")}
        tag = "txsearch %dx%d x%d" % (s, s, nt)
        fns.append((tag, lambda s=s, n=n, dev1=dev1, mask=mask: ctx.rdo_txsearch_batch(
            po, pr, s, s, dev1, mask, args.qindex, 3, scales=scales, n=n, outs=touts[s])))
        abytes[tag] = (2 * ((s + 7) * (s + 7) + s * s) + 16 + 10 * nt) * n
        px += n * nt * s * s
        for t in types:
            ct_ = c1.copy()
            ct_["tx_type"] = t
            devt = torch.from_numpy(ct_.view(np.uint8).reshape(-1).copy()).cuda()
            o1 = {"eob": torch.empty(n, dtype=torch.int16, device="cuda"), "dist": torch.empty(n, dtype=torch.int64, device="cuda")}
            fns_ind.append(("single %dx%d" % (s, s), lambda s=s, n=n, devt=devt, o1=o1: ctx.rdo_pixel_cand_batch(
                po, pr, s, s, devt, args.qindex, 3, scales=scales, n=n, outs=o1, want_sad=False, want_satd=False)))
    per_ind, step_ind = timed(fns_ind)
    per, step = timed(fns)
    n_chk, bad = 0, []
    hs = scales.cpu().numpy().view(np.uint32)
    for s, (c1, mask, types) in tcands.items():
        idx = sample(len(c1))[:16]
        sub = np.ascontiguousarray(c1[idx])
        nt = len(types)
        eob, dist = np.zeros((len(sub), nt), np.uint16), np.zeros((len(sub), nt), np.uint64)
        assert L.r1o_rdo_txsearch_batch(C.byref(pa), C.byref(pb), None, s, s, TS[s], O.ptr(sub), len(sub), mask, args.qindex, 0, 0, 0,
                                        3, O.ptr(hs), hs.shape[1], 0, 0, None, None, O.ptr(eob), O.ptr(dist), None, None, None) == 0
        ix = torch.from_numpy(idx.astype(np.int64)).cuda()
        n_chk += len(idx) * nt
        if not (np.array_equal(touts[s]["dist"].index_select(0, ix).cpu().numpy().view(np.uint64), dist) and
                np.array_equal(touts[s]["eob"].index_select(0, ix).cpu().numpy().view(np.uint16), eob)):
            bad.append("txsearch %d" % s)
    ratio = {}
    for s, (c1, mask, types) in tcands.items():
        tag = "txsearch %dx%d x%d" % (s, s, len(types))
        # the events of the independent launches are averaged per launch: the search costs len(types) of them
        ratio["%dx%d" % (s, s)] = round(per[tag] / (per_ind["single %dx%d" % (s, s)] * len(types)), 4)
    line("tx_search_4k_10bit", "%dx%d 10-bit luma: transform-type search of ONE inter prediction per block, 32x32 (DCT_DCT, IDTX), "
         "16x16 and 8x8 (the 7 RAV1E_TX_TYPES): mc -> diff once, then fwd -> quantize -> dequantize -> inverse -> cdef_dist per "
         "type (r1_rdo_txsearch_batch)" % (w, h), px, per, step, abytes, n_chk, bad,
         {"dtype": "u16", "px_note": "Mpixels/s counts (block, type) evaluations",
          "independent_single_type_launches": {"ms_per_launch": {t: round(v, 4) for t, v in per_ind.items()},
                                               "ms_per_pass": round(step_ind, 4),
                                               "fanout_over_independent": ratio,
                                               "pass_ratio": round(step / step_ind, 4)}},
         working_set=ho.nbytes + hr.nbytes,
         wbytes={"txsearch %dx%d x%d" % (s_, s_, len(t_[2])): 10 * len(t_[2]) * len(t_[0]) for s_, t_ in tcands.items()},
         # the fan-out launches of 8x8 / 16x16 are the MT instantiations (",2,true"); 32x32 runs plain launches per type
         kkeys={"txsearch %dx%d x%d" % (s_, s_, len(t_[2])): (rdo_launch_key(10, s_, 2, len(t_[0]))[0] + (",true" if s_ <= 16 else ",false"),
                                                              rdo_launch_key(10, s_, 2, len(t_[0]))[1]) for s_, t_ in tcands.items()})
    del touts, fns, fns_ind
    torch.cuda.empty_cache()
    lines.append(frame_line(ctx, args, timed, launch_how))
    return lines


def frame_line(ctx, args, timed, launch_how):
    """BASELINE.json configs[3] as ONE measured workload: every stage of a coded 3840x2160 10-bit 4:2:0 frame
    (tools/frame_stages.py: tile ME 8 tiles x 3 references, sub-pel search, 13-mode intra pre-screen, pixel-domain
    chain on luma AND both chroma planes, the 7-type transform search, deblock level search + filter, CDEF search +
    filter, restoration search + filter), HIP events per stage, a strided parity sample per stage against the CPU
    oracle (outside the timed region), the dominant stage against its roofline."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import frame_stages
    F = frame_stages.build(ctx, 10, k=args.k, qindex=args.qindex)
    fw, fh, bd = F["frame"]
    parity = F["verify"]()
    per, step = timed(F["stages"], graph_ok=False)     # the ME and the composite CDEF call synchronise: not capturable
    per = {n: v for n, v in per.items() if not n.endswith("_untimed")}   # the restore of the in-place filtered planes
    ok_me = bool(ctx.me_status(wait=True)[0])
    bad = [n for n, (c, ok) in parity.items() if not ok] + ([] if ok_me else ["tile ME flagged a timed-out wait"])
    n_chk = sum(c for c, _ in parity.values())
    dom = max(per, key=lambda t: per[t])
    kkeys = {"rdo_pixel_luma_%dx%d_K%d" % (s, s, args.k): rdo_launch_key(10, s, 2, n) for s, n in F["luma_launch_n"].items()}
    kkeys["estimate_tile_motion_8tiles_x_3refs"] = ("k_me_persist<2,true>", 2048 * 64)     # one persistent launch, 2048 waves
    lc, lsrc = launch_pmc(kkeys.get(dom))
    roof = build_roofline("config4_frame: %s" % dom, F["algorithmic_bytes"][dom], per[dom], None, None, None,
                          working_set=F["working_set_bytes"], line_counters=lc, line_src=lsrc)
    if dom.startswith("estimate_tile_motion"):
        roof["binding_roof"] = "latency"
        roof["binding_note"] = ("the hierarchical ME is a dependent chain (block rows x pyramid levels per tile, ~128 steps of ~10 us): "
                                "neither HBM nor VALU issue limits it; counters: profiles/r05_pmc_me.json")
    # the same work as a frame PIPELINE keeps it in flight: four streams (frame n + 1's motion search and pre-screens |
    # frame n's luma chain | its chroma chain + type search | frame n - 1's post-filter decisions), wall clock per pass
    p4 = F["pipelined4"]
    for _ in range(3):
        p4()
    torch.cuda.synchronize()
    sustain(p4, args.prewarm_ms)
    t4 = time.perf_counter()
    for _ in range(20):
        p4()
    torch.cuda.synchronize()
    four_ms = (time.perf_counter() - t4) / 20 * 1e3
    if not bool(ctx.me_status(wait=True)[0]) and "tile ME flagged a timed-out wait" not in bad:
        bad.append("tile ME flagged a timed-out wait")
    cand_px = sum(F["candidate_pixels"].values())
    cand_ms = sum(per[n] for n in F["candidate_pixels"])
    return {"name": "config4_frame_4k_10bit", "metric": "frames/s", "value": round(1e3 / sum(per.values()), 2), "unit": "frames/s",
            "steps": 20, "ms_per_step": round(sum(per.values()), 4), "ms_per_pass_wall": round(step, 4), "dtype": "u16",
            "config": {"workload": "%dx%d 10-bit 4:2:0, every device-resident stage of one coded frame (BASELINE configs[3]: speed-4 "
                                   "full RDO + CDEF, all kernels): tile ME 8 tiles x 3 refs, sub-pel search, 13-mode intra pre-screen, "
                                   "pixel-domain chain K=%d on luma and both chroma planes, 7-type transform search, deblock level "
                                   "search + filter, CDEF strength search + filter, restoration search (8 sets) + filter, and rdo_loop_decision's "
                                   "iteration with both filters on: working copy, restoration leg on it, the second pass of both legs" % (fw, fh, args.k),
                       "loop_decision": F.get("loop_decision"),
                       "launch": "one call per stage, serialized; stage_ms by HIP events on the launch stream"},
            "stage_ms": {n: round(v, 4) for n, v in per.items()},
            "pipelined": {"value": round(1e3 / four_ms, 2), "unit": "frames/s", "ms_per_frame": round(four_ms, 4),
                          "plan": {k_: len(v_) for k_, v_ in F["plan4"].items()},
                          "note": "the same stages on four free-running streams (groups of a frame pipeline: the next frame's "
                                  "motion search + pre-screens | luma chain | chroma chain + type search | the previous frame's "
                                  "post-filter decisions), wall clock; `value` stays the serialized sum of stage times"},
            "rdo_candidate_Mpixels_s": round(cand_px / (cand_ms * 1e-3) / 1e6, 1),
            "rdo_candidate_note": "luma + chroma candidates and (block, type) evaluations of the type search over their stages' time",
            "roofline": roof, "dominant_stage": dom,
            "algorithmic_bytes_by_stage": {n: int(v) for n, v in F["algorithmic_bytes"].items()},
            "parity_by_stage": {n: {"checked": c, "ok": ok} for n, (c, ok) in parity.items()},
            "parity_not_sampled_here": {"lookahead_intra_costs": "tests/test_gpu_ref_vectors.py (lookahead_ref.npz)",
                                        "update_block_importances_3refs": "tests/test_gpu_ref_vectors.py (lookahead_chain_ref.npz)",
                                        "intra_prescreen_16x16_13modes": "tests/test_gpu_parity.py (predict_ref.npz, 4 992 dispatches)",
                                        "cdef_strength_search_8_presets_420": "tests/test_gpu_fullsize.py at 3840x2160 10-bit, whole frame",
                                        "cdef_luma": "tests/test_gpu_fullsize.py at 3840x2160 10-bit, whole plane",
                                        "lrf_sgrproj_luma": "tests/test_gpu_parity.py (lrf_ref.npz)"},
            "parity_checked": n_chk, "parity_ok": not bad, "parity_bad": bad}


# N = 1: one stream per block size, the 64x64 launch (the longest workgroups) at high queue priority;
# the streams free-run, so the launches of consecutive steps overlap and the CUs / register files a
# draining launch leaves idle are filled by the other sizes (same-box A/B: profiles/r04_ab_notes.md, ab6)
PIPELINE_PLAN = "64h/32l/16l/8l"


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-exec this very command
    line under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1, a free port) --
    the launch the task statement's driver uses.  Under a launcher (WORLD_SIZE set) this is a
    no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    respawn_under_launcher(args)
    import torch
    import torch.distributed as dist
    from rav1e_amd import tiles
    from rav1e_amd import workload as W
    from rav1e_amd.api import Context, Plane

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev_index = 0 if args.single_device else local_rank
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.cuda.current_device()
    ctx = Context(dev)

    fw, fh, bd = args.width, args.height, args.bit_depth
    bpp = 1 if bd == 8 else 2
    # ---- synthetic planes (uniform random, seeds 1 / 2), resident in HBM ----
    host_org = W.random_plane_array(fw, fh, bd, 1)
    host_ref = W.random_plane_array(fw, fh, bd, 2)
    org = Plane.from_numpy(host_org, fw, fh, bd, 88, 88)
    ref = Plane.from_numpy(host_ref, fw, fh, bd, 88, 88)

    # ---- candidates: whole frame at N=1, tile `rank` otherwise ----
    cands = tiles.shard_candidates(fw, fh, args.k, rank, world)
    dcands = {s: torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
              for s, c in cands.items()}
    outs = {}
    for s, c in cands.items():
        n = len(c)
        outs[s] = {"sad": torch.empty(n, dtype=torch.int32, device="cuda"),
                   "satd": torch.empty(n, dtype=torch.int32, device="cuda"),
                   "coeffs": torch.empty((n, s * s), dtype=torch.int16 if bpp == 1 else torch.int32,
                                         device="cuda")}
    my_px = sum(len(c) * s * s for s, c in cands.items())

    # ---- exchange step for N > 1: the tile-boundary rectangles the neighbours'
    # post filters read (point to point), then every rank's tile of the reference plane to
    # everybody (all-gather) -- through the C ABI (csrc/comm.hip links RCCL), on the SAME
    # stream as the launches: the next step's motion compensation reads the plane the
    # all-gather writes, so the exchange is on the critical path, not hidden behind it ----
    comm, exch_note, rects = None, None, None
    if world > 1:
        rects = W.tile_rects(world, fw, fh)
        x0, y0, x1, y1 = rects[rank]
        my_tile = ref.data[ref.yorigin + y0:ref.yorigin + y1, ref.xorigin + x0:ref.xorigin + x1]
        orig_vis = tiles.visible(ref).clone()      # what the after-run check recomputes the tiles from
        try:
            comm = tiles.Comm(ctx, rank, world)
            exch_note = "r1_comm_exchange_halos (64 px) + r1_comm_allgather_tiles per step, in stream order"
        except Exception as e:   # keep the scaling run alive; the JSON says which path ran
            exch_note = ("torch.distributed all_gather_into_tensor of row slabs, NO halo leg -- exchange_ms is not comparable with "
                         "the peer-store / RCCL p2p paths (C-ABI comm failed: %s)" % (str(e)[:80],))
            send, gathered = tiles.make_exchange_buffers(ref.data, rank, world)
    push_why = None   # why the peer-store exchange is not the one in use (None: it is, or N = 1)

    full = args.chain != "cand"
    pixel = args.chain == "pixel"
    if pixel:
        scales = torch.from_numpy(np.random.default_rng(9).integers(
            1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8)).astype(np.int32)).cuda()
    for s, c in cands.items():
        n = len(c)
        if pixel and n:
            del outs[s]["coeffs"]
            outs[s].update(eob=torch.empty(n, dtype=torch.int16, device="cuda"),
                           dist=torch.empty(n, dtype=torch.int64, device="cuda"))
        elif full and not pixel:
            del outs[s]["coeffs"]
            outs[s].update(eob=torch.empty(n, dtype=torch.int16, device="cuda"),
                           tx_dist=torch.empty(n, dtype=torch.int64, device="cuda"),
                           est_rate=torch.empty(n, dtype=torch.int64, device="cuda"))

    def build_launches(refp):
        """one callable per ladder size: the step's launch of that size reading reference plane `refp`"""
        if pixel:
            return {s: (lambda s=s, n=len(c): ctx.rdo_pixel_cand_batch(
                        org, refp, s, s, dcands[s], args.qindex, 3, scales=scales, n=n, outs=outs[s]))
                    for s, c in cands.items() if len(c)}
        if full:
            return {s: ctx.prepare_rdo_full_cand(org, refp, s, s, dcands[s], len(cands[s]), args.qindex, outs[s])
                    for s in cands if len(cands[s])}
        return {s: ctx.prepare_rdo_cand(org, refp, s, s, dcands[s], len(cands[s]), outs[s])
                for s in cands if len(cands[s])}
    launches = build_launches(ref)
    # N > 1 with the exchange as peer stores: the reconstruction of a step goes into the OTHER plane of
    # a two-plane ring (a new buffer per frame, as in the reference), so nobody stores into a plane
    # a peer may still be reading; ring[cur[0]] is the plane the step's launches read
    ring, ring_launches, ring_peers, cur = [ref], [launches], [], [0]
    tile_ring, xsteps = [None], [0]     # tiles.TileRing when the peer stores are in use; exchange steps done

    def abytes_per_cand(s):
        if pixel:  # window + source (read twice: residual, distortion) + sad/satd/eob/dist
            return bpp * ((s + 7) * (s + 7) + 2 * s * s) + 4 + 4 + 2 + 8
        if full:   # window + source + sad/satd/eob/tx_dist/est_rate; no coefficient store
            return bpp * ((s + 7) * (s + 7) + s * s) + 4 + 4 + 2 + 8 + 8
        return W.algorithmic_bytes_per_cand(s, s, bpp)
    ev = {s: [] for s in cands}
    # per-kernel events feed `roofline` (N = 1); at N > 1 they would only add
    # host work to steps that are a fraction of a millisecond long
    use_events = not args.no_events and world == 1
    # N > 1: three events on every EV_EVERY-th timed step split the step into its launches
    # (`compute_ms`) and the exchange behind them (`exchange_ms`: stand-in reconstruction write +
    # halo p2p + tile all-gather); those steps run their launches on the main stream
    split_events = not args.no_events and world > 1
    xev = []

    # A timing-event pair costs ~0.5 % of a step (the event drains the stream), so every
    # EV_EVERY-th timed step carries them (around each size's launch), the others run bare.
    EV_EVERY = max(1, args.event_every)
    nstep = [0]

    if args.streams <= 0:
        args.streams = 4 if world > 1 else 1
    fan = args.streams > 1
    fan_order = list(W.LADDER)
    if fan and args.stream_plan:
        # "64h/32l/16l/8l": streams separated by "/", the sizes a stream runs in order, h / l = the
        # stream's queue priority.  The launch with the long workgroups goes out first at high
        # priority; the other sizes' workgroups fill the CUs it leaves idle while it drains.
        least, greatest = torch.cuda.Stream.priority_range()
        size_streams, fan_order = {}, []
        for grp in args.stream_plan.split("/"):
            items = grp.split(",")
            st = torch.cuda.Stream(priority=greatest if items[0].endswith("h") else least)
            for it in items:
                size_streams[int(it.rstrip("hl"))] = st
                fan_order.append(int(it.rstrip("hl")))
        assert sorted(fan_order) == sorted(W.LADDER), args.stream_plan
    else:
        size_streams = {s: torch.cuda.Stream() for s in W.LADDER} if fan else {}

    overlap_side = [None]

    def step(timed, exchange=True):
        mark = timed and use_events and nstep[0] % EV_EVERY == 0
        split = timed and split_events and exchange and nstep[0] % EV_EVERY == 0
        if timed:
            nstep[0] += 1
        main = torch.cuda.current_stream()
        overlapped = bool(args.overlap_exchange and world > 1 and exchange and ring_peers)
        if overlapped:
            # the border of this rank's new tile and its halo stores FIRST (side stream), the step's launches beside them
            if overlap_side[0] is None:
                overlap_side[0] = torch.cuda.Stream()
            tile_ring[0].begin_overlapped(overlap_side[0])
        if fan and not split and (not mark or args.step_join):
            # independent launches, one stream per block size (each stream is in order with the
            # same size's launch of the previous step, which wrote the same output buffers).
            # --step-join: the streams fork from and join the main stream every step (a step is
            # finished before the next one starts) and the timing events of a marked step bracket
            # each launch ON ITS OWN STREAM -- co-scheduled durations, what rocprofv3 sees too
            for s in fan_order:
                if len(cands[s]):
                    st = size_streams[s]
                    if args.step_join:
                        st.wait_stream(main)
                    with torch.cuda.stream(st):
                        if mark:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                        ring_launches[cur[0]][s]()
                        if mark:
                            e1.record()
                            ev[s].append((e0, e1))
            if args.step_join:
                for st in set(size_streams.values()):
                    main.wait_stream(st)
        else:
            for st in size_streams.values():
                main.wait_stream(st)
            if split:
                xe = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                xe[0].record()
            for s in W.LADDER:
                n = len(cands[s])
                if n == 0:
                    continue
                if mark:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ring_launches[cur[0]][s]()
                if mark:
                    e1.record()
                    ev[s].append((e0, e1))
            if split:
                xe[1].record()
            for st in size_streams.values():
                st.wait_stream(main)
        if world > 1 and exchange:
            for st in size_streams.values():
                main.wait_stream(st)
            # stand-in for the reconstruction write: this rank's tile of the plane differs every
            # step, so the exchange never moves bytes the peers already hold
            if ring_peers:
                # peer stores: the new tile goes into the other plane of the ring, its borders and
                # then the whole tile straight into the peers' copies of that plane (tiles.TileRing)
                if overlapped:
                    tile_ring[0].finish_overlapped()
                else:
                    tile_ring[0].advance()
                cur[0] = tile_ring[0].cur
            elif comm is not None:
                my_tile.bitwise_xor_(tiles.ring_delta(xsteps[0]))
                comm.exchange_tile_halos(ref, rects)
                comm.allgather_tiles(ref, rects)
            else:
                my_tile.bitwise_xor_(tiles.ring_delta(xsteps[0]))
                tiles.exchange_rows(send, gathered)
            xsteps[0] += 1
            if split:
                xe[2].record()
                xev.append(xe)
            for st in size_streams.values():
                st.wait_stream(main)

    def fence():
        for st in size_streams.values():
            torch.cuda.current_stream().wait_stream(st)
        if world > 1:
            # drain this rank's stream (the C-ABI communicator's collectives included) before
            # torch's own communicator runs its barrier: the two never have work in flight together
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    # ---- N > 1 (and --verify-exchange at N = 1): did the exchange move the right bytes?  Tagged
    # tiles through one halo exchange and one tile gather, checked on every rank, BEFORE the pre-warm and the warm-up (the check's copies and host round trips let
    # the clocks drop; measured: -12 % on the following 20 steps when it sat after the warm-up) (rav1e_amd/tiles.py verify_exchange).  A wrong rectangle would not slow the
    # candidates down -- they would read garbage at full speed -- so the number alone cannot tell.
    exchange_ok = None
    if world > 1 or args.verify_exchange:
        vrects = rects if rects is not None else W.tile_rects(world, fw, fh)
        # the communicator the exchange goes through: the run's own; at N = 1 one made for the check;
        # None when the C-ABI communicator could not be made at N > 1 (RCCL missing, or the dry run
        # with every rank on one GPU) -- peer stores then hand-shake through the host
        vcomm = comm if comm is not None else (tiles.Comm(ctx, rank, world) if world == 1 else None)
        mine_ok = None
        if args.exchange in ("auto", "push"):
            # the exchange as direct peer stores (csrc/comm.hip r1_comm_push_*): both planes of the
            # ring mapped on every rank, then the same tagged-tile check through the stores.  In
            # use only if EVERY rank mapped and verified; otherwise the RCCL p2p exchange below.
            def host_barrier():
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()

            def all_agree(ok):
                flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda")
                if world > 1:
                    torch.cuda.synchronize()
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return flag.item() == 1.0
            push_ok = None
            if world > 1 and not args.single_device and ctx.lib.r1_ipc_peer_access(ctx.h) == 0:
                push_why = "hipDeviceCanAccessPeer refuses a GPU of this node"   # do not map, do not store
            ring.append(Plane.from_numpy(host_ref, fw, fh, bd, 88, 88))
            # mapping is a collective per plane: every rank learns whether everybody mapped before
            # anyone enters the next one (a rank that failed alone would leave the others inside it)
            for pl in ring:
                if not all_agree(push_why is None):   # before the collective: nobody enters it alone
                    push_why = push_why or "another rank has no peer access"
                    break
                try:
                    ring_peers.append(tiles.PeerPlanes(ctx, pl, rank, world, comm=vcomm))
                except Exception as e:   # noqa: BLE001 -- keep the run alive on the other exchange
                    push_why = "%s" % (str(e)[:120],)
                if not all_agree(push_why is None):
                    push_why = push_why or "another rank could not map its peers' planes"
                    break
            try:
                if push_why is not None:
                    raise RuntimeError(push_why)
                push_ok = {"halo": True, "gather": True}
                for pl, pp in zip(ring, ring_peers):
                    v = tiles.verify_exchange(pl, vrects, rank, world, lambda: pp.push_halos(vrects),
                                              lambda: pp.push_tile(vrects), pre=host_barrier)
                    push_ok = {k: push_ok[k] and v[k] for k in push_ok}
                if not all(push_ok.values()):
                    push_why = "tagged tiles did not arrive: %s" % (push_ok,)
            except Exception as e:   # noqa: BLE001
                push_why = "%s" % (str(e)[:120],)
                push_ok = None
            if all_agree(push_why is None):
                mine_ok = push_ok
                through = "peer stores (r1_comm_push_*), both planes of the ring"
                ring_launches.append(build_launches(ring[1]))
                tile_ring[0] = tiles.TileRing(ring, ring_peers, vrects, rank, tiles.visible(ring[0]).clone())
                exch_note = None if world == 1 else (
                             "r1_comm_push_frame per step: halo stores (64 px) + tile stores into IPC-mapped planes "
                             "(two-plane ring) behind ONE hand-shake: %s" %
                             ("r1_comm_barrier, in stream order" if vcomm is not None else
                              "a host hand-shake (stream synchronize + torch.distributed barrier: no C-ABI communicator)"))
            else:
                push_why = push_why or "another rank could not use it"
                host_barrier()                   # nobody unmaps while a peer may still store
                for pp in ring_peers:
                    pp.close()
                del ring_peers[:], ring[1:]
        if (mine_ok is None or world == 1) and vcomm is not None:
            p2p_ok = tiles.verify_exchange(ref, vrects, rank, world,
                                           lambda: vcomm.exchange_tile_halos(ref, vrects),
                                           lambda: vcomm.allgather_tiles(ref, vrects))
            if mine_ok is None:
                mine_ok, through = p2p_ok, "RCCL p2p (r1_comm_exchange_halos + r1_comm_allgather_tiles)"
            else:   # world 1: both sets of entry points go through their (peerless) motions
                mine_ok = {k: mine_ok[k] and p2p_ok[k] for k in mine_ok}
                through += " and RCCL p2p"
        elif mine_ok is None:
            # the torch.distributed fallback moves row slabs of the allocation, no halo leg: a rank's
            # slab of the painted plane must arrive as it left (checked through the gathered buffer)
            rows, lo, hi = tiles.owned_rows(ref.data.shape[0], rank, world)
            send.fill_(rank + 1)
            tiles.exchange_rows(send, gathered)
            torch.cuda.synchronize()
            ok = all(bool((gathered[r * rows:(r + 1) * rows] == r + 1).all().item()) for r in range(world))
            send.zero_()
            send[: hi - lo] = ref.data[lo:hi]
            mine_ok = {"halo": None, "gather": ok}
            through = "torch.distributed row slabs"
        if world == 1:
            for pp in ring_peers:
                pp.close()
            del ring_peers[:], ring[1:], ring_launches[1:]
            vcomm.close()
        flags = torch.tensor([1.0 if mine_ok[k] in (True, None) else 0.0 for k in ("halo", "gather")],
                             dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        exchange_ok = {"halo": None if mine_ok["halo"] is None else bool(flags[0].item() == 1.0),
                       "gather": None if mine_ok["gather"] is None else bool(flags[1].item() == 1.0),
                       "checked_on": "every rank (MIN over ranks)" if world > 1 else "rank 0 (world 1: no peers)",
                       "through": through,
                       "peer_stores_not_used_because": push_why}
        fence()
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < args.prewarm_ms:
            for _ in range(8):
                step(False, exchange=False)    # ranks leave this loop at different counts: no collectives in it
            torch.cuda.synchronize()
            prewarm_steps += 8
        if world > 1:                      # every rank leaves the pre-warm before anyone warms up
            fence()
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    # ---- N > 1: is the exchange still right AFTER the timed loop?  Every rank recomputes what every tile of
    # the plane the last step left must hold -- ORIGINAL ^ ring_tag(exchange steps done), the peers' tiles
    # included -- on its own copy; MIN over ranks.  A store that raced the hand-shake, or arrived a ring cycle
    # late, leaves a stale tag that the single pre-run check cannot see.
    if world > 1 and exchange_ok is not None:
        if tile_ring[0] is not None:
            after = tile_ring[0].check() and tile_ring[0].t == xsteps[0]
        elif comm is not None:
            after = bool(torch.equal(tiles.visible(ref), torch.bitwise_xor(orig_vis, tiles.ring_tag(xsteps[0]))))
        else:
            after = None    # torch.distributed row slabs: no per-tile content to recompute (and no halo leg)
        flag = torch.tensor([1.0 if after in (True, None) else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        exchange_ok["after_run"] = None if after is None else bool(flag.item() == 1.0)
        exchange_ok["after_run_note"] = ("every rank's copy of every tile == original ^ ring_tag(%d exchange steps), checked "
                                         "after the timed loop, MIN over ranks" % xsteps[0]) if after is not None else \
            "torch.distributed row-slab fallback: no halo leg, exchange_ms not comparable with the other two paths"

    # ---- aggregate: max time over ranks, total pixels over ranks ----
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        p = torch.tensor([my_px], dtype=torch.float64, device="cuda")
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        total_px = float(p.item())
    else:
        total_px = float(my_px)
    split_ms = None
    if world > 1 and split_events:
        # mean over this rank's marked steps, then the slowest rank's figure and rank 0's own
        mine = [0.0, 0.0]
        if xev:
            mine = [sum(e[0].elapsed_time(e[1]) for e in xev) / len(xev),
                    sum(e[1].elapsed_time(e[2]) for e in xev) / len(xev)]
        t = torch.tensor(mine, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        split_ms = {"compute_ms": round(float(t[0].item()), 4), "exchange_ms": round(float(t[1].item()), 4),
                    "rank0_compute_ms": round(mine[0], 4), "rank0_exchange_ms": round(mine[1], 4),
                    "samples": len(xev),
                    "note": "HIP events on every %dth timed step (launches on one stream in those steps): "
                            "compute = the step's launches, exchange = stand-in reconstruction write + "
                            "the exchange config.exchange names; max over ranks" % EV_EVERY}

    if rank == 0:
        per = {}
        for s in cands:
            if ev[s]:
                ms = [a.elapsed_time(b) for a, b in ev[s]]
                per[s] = sum(ms) / len(ms)
        if per:
            dom = max(per, key=lambda s: per[s])
            n_dom = len(cands[dom])
            abytes = abytes_per_cand(dom) * n_dom
            kname = "k_rdo_cand<bd=%d,%dx%d%s>" % (bd, dom, dom, ",pixel" if pixel else (",quant" if full else ""))
            pmc, pmc_src = (None, None) if full else pmc_counters(bd, dom, fw, fh, args.k)
            wr = n_dom * ((8 + dom * dom * (2 if bpp == 1 else 4)) if not full else (18 if pixel else 26))
            roof = build_roofline(kname, abytes, per[dom], pmc, pmc_src,
                                  None if full else valu_issue_model(bd, dom),
                                  working_set=host_org.nbytes + host_ref.nbytes, write_bytes=wr)
        else:
            dom, per = None, {}
            abytes = sum(abytes_per_cand(s) * len(c) for s, c in cands.items())
            roof = build_roofline("k_rdo_cand (all sizes, step time)", abytes, dt / args.steps * 1e3,
                                  None, None, None)
            roof["avg_launch_ms"] = None
        res = {
            "metric": "RDO-candidate Mpixels/s (dist+fwd_tx+mc) at 4K speed-6" if not full else
                      ("full RDO-candidate Mpixels/s (mc+dist+fwd_tx+quantize+tx_dist+rate) at 4K speed-6"
                       if not pixel else "pixel-domain RDO-candidate Mpixels/s (mc+dist+fwd_tx+quantize+"
                       "inverse+cdef_dist) at 4K speed-6"),
            "value": round(total_px * args.steps / dt / 1e6, 2),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8" if bpp == 1 else "u16",
            "data": "synthetic",
            "config": {"workload": "%dx%d %d-bit luma, speed-6 ladder 64/32/16/8, K=%d fused "
                                   "candidates per block (put_8tap REGULAR -> SAD+SATD -> diff -> "
                                   "fwd DCT_DCT), MV +-32 px, random 1/16-pel fractions"
                                   % (fw, fh, bd, args.k),
                       "candidates_per_step": int(sum(len(c) for c in cands.values())) if world == 1
                       else None,
                       "tiles": world, "exchange": exch_note, "exchange_ok": exchange_ok, "launch_streams": args.streams,
                       "compute_ms": split_ms["compute_ms"] if split_ms else None,
                       "exchange_ms": split_ms["exchange_ms"] if split_ms else None,
                       "step_split": split_ms,
                       "parallelism": "tile-per-gpu x%d" % world if world > 1 else "single-gpu"},
            "roofline": roof,
            "prewarm_steps": prewarm_steps,
            "kernel_ms": {str(s): round(v, 4) for s, v in per.items()},
            "kernel_ms_note": "HIP events around each launch of every %dth timed step "
                              "(%d samples per size)" % (EV_EVERY, max([len(v) for v in ev.values()] + [0])),
        }
        if world == 1 and not fan and not args.no_extra:
            # The same K steps with the four launches of a step on four FREE-RUNNING streams (no join
            # between steps: the launches of consecutive steps overlap, as independent batches in
            # flight do -- the tiles of a frame are encoded independently), the 64x64 launch at high
            # queue priority.  The CUs and register files a draining launch leaves idle are filled by
            # the other sizes.  Reported beside `value`, which stays the serialized step: per-launch
            # durations (and with them `roofline`) are not separable once launches share the GPU.
            least, greatest = torch.cuda.Stream.priority_range()
            pst = {s_: torch.cuda.Stream(priority=greatest if s_ == max(W.LADDER) else least) for s_ in W.LADDER}

            def pipe_steps(n_):
                for _ in range(n_):
                    for s_ in sorted(W.LADDER, reverse=True):
                        if len(cands[s_]):
                            with torch.cuda.stream(pst[s_]):
                                launches[s_]()
            torch.cuda.synchronize()
            pipe_steps(args.warmup)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            pipe_steps(args.steps)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - tp
            res["pipelined"] = {"value": round(total_px * args.steps / dtp / 1e6, 2), "unit": "Mpixels/s",
                                "ms_per_step": round(dtp / args.steps * 1e3, 4), "steps": args.steps,
                                "plan": PIPELINE_PLAN,
                                "note": "the same steps, one free-running stream per block size ('/' separates streams, "
                                        "h / l = queue priority), no join between steps; outputs checked like the "
                                        "serialized steps' (the parity legs below run on what these launches left)"}
        if pixel and world == 1:
            # The same chain as rav1e's RDO runs it: rdo_tx_size_type -> encode_tx_block -> compute_distortion
            # (src/rdo.rs:1073, src/encoder.rs:1404-1661, src/rdo.rs:254-347) never takes SAD or SATD of a
            # candidate -- those belong to the motion search (compute_mv_rd, src/me.rs:1445-1463).  `value` keeps
            # them (the step the previous rounds timed: the headline candidate carried through the quantizer);
            # this is the step without them, NULL sad / satd outputs, same launches otherwise.
            lo = {}
            for s_, c_ in cands.items():
                n_ = len(c_)
                o_ = {"eob": outs[s_]["eob"], "dist": outs[s_]["dist"]}
                lo[s_] = (lambda s_=s_, n_=n_, o_=o_: ctx.rdo_pixel_cand_batch(
                    org, ref, s_, s_, dcands[s_], args.qindex, 3, scales=scales, n=n_, outs=o_, want_sad=False,
                    want_satd=False))
            for _ in range(args.warmup):
                for s_ in W.LADDER:
                    lo[s_]()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                for s_ in W.LADDER:
                    lo[s_]()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            res["rdo_only"] = {"value": round(total_px * args.steps / dt1 / 1e6, 2), "unit": "Mpixels/s",
                               "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                               "note": "the pixel-domain chain without SAD / SATD of the candidate (NULL outputs): what "
                                       "rdo_tx_size_type -> encode_tx_block -> compute_distortion compute per candidate; "
                                       "`value` above keeps both (the motion search's distortions ride along)"}
        bad = []
        if world == 1 and not full and not args.no_extra:
            res["extra_lines"] = extra_lines(ctx, args)
            if (fw, fh, bd) == (3840, 2160, 8):   # the default run: BASELINE configs 2-4 beside the headline
                res["extra_lines"] += config_lines(ctx, args)
            bad += ["extra:" + e["name"] for e in res["extra_lines"] if not e.get("parity_ok", True)]
        if world == 1 and args.cpu_seconds > 0 and not full:
            # the CPU legs evaluate strided samples of this very step at 4K: what they return is
            # compared with what the timed launches left in HBM (sad, satd and every coefficient)
            res["cpu_baseline"], parity = cpu_baseline(args, host_org, host_ref, cands, outs)
            bad += parity.pop("bad")
            res.update(parity)
            res["parity_ok"] = not bad
        emit(res, args.detail_out)
        if bad:
            print("PARITY FAILURE at block sizes %s" % bad, file=sys.stderr)
            sys.exit(3)
    if ring_peers:
        torch.cuda.synchronize()
        dist.barrier()                       # nobody unmaps a plane a peer may still store into
        for pp in ring_peers:
            pp.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
