#!/usr/bin/env python3
"""bench.py -- RDO-candidate Mpixels/s (dist + fwd_tx + mc) at 4K speed-6.

One "step" = one pass of the fused hot path (put_8tap -> SAD + SATD -> diff ->
forward DCT, one launch per block size) over every candidate of a synthetic 4K
frame: for each block of the speed-6 ladder 64/32/16/8 (rav1e_amd/workload.py)
K candidates with random motion vectors and 1/16-pel fractions.  Inputs are
resident in HBM before the timed region.  N > 1: one process per GPU, rank r
owns tile r of the frame (uniform tiling, src/tiling/tiler.rs:56) and the
ranks exchange their rows of the reconstructed/reference plane with one RCCL
all-gather per step, issued on a side stream.

Prints ONE JSON line on rank 0 (contract in the task statement), carrying
`roofline` for the dominant kernel and `cpu_baseline` (the CPU oracle timed on
the host cores, rank 0 / N=1 only).
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--streams", type=int, default=1,
                    help="> 1: the launches of a step (one per block size, independent of each other) "
                         "go to one HIP stream per size, so that a launch fills the CUs the previous "
                         "one is draining; the steps that carry timing events stay on one stream")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps for this long BEFORE the W warm-up steps, so that short runs "
                         "(small W and K) are measured at the clocks a long run settles at; 0 = off")
    ap.add_argument("--no-events", action="store_true",
                    help="skip per-kernel event timing (roofline.achieved falls back to step time)")
    return ap.parse_args()


def cpu_baseline(args, host_org, host_ref, cands):
    """Time the CPU oracle (port of the reference's Rust path, OpenMP over
    candidates) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = O.lib()
    cores = os.cpu_count() or 1
    L.r1o_set_threads(cores)
    ho = O.HostPlane(args.width, args.height, args.bit_depth)
    hr = O.HostPlane(args.width, args.height, args.bit_depth)
    ho.data, hr.data = host_org, host_ref
    pa, pb = ho.cstruct(), hr.cstruct()
    ct = np.int16 if ho.bpp == 1 else np.int32

    check = {}

    def run(frac_n):
        px = 0
        t0 = time.perf_counter()
        for s, c in cands.items():
            n = max(1, int(len(c) * frac_n))
            idx = np.arange(0, len(c), max(1, len(c) // n))[:n]
            sub = np.ascontiguousarray(c[idx])
            sad = np.zeros(len(sub), np.uint32)
            satd = np.zeros(len(sub), np.uint32)
            co = np.zeros((len(sub), s * s), ct)
            ts = {64: 4, 32: 3, 16: 2, 8: 1}[s]
            rc = L.r1o_rdo_cand_batch(C.byref(pa), C.byref(pb), s, s, ts, O.ptr(sub), len(sub),
                                      O.ptr(sad), O.ptr(satd), O.ptr(co), None)
            assert rc == 0
            px += len(sub) * s * s
            check[s] = (idx, sad, satd, co)      # what the oracle says about these candidates
        return px, time.perf_counter() - t0
    frac, px, dt = 0.002, 0, 0.0
    while True:                               # grow the sample until it is measurable
        px, dt = run(frac)
        if dt >= 1.5 or frac >= 1.0:
            break
        frac = min(1.0, frac * 4)
    if dt < 0.6 * args.cpu_seconds and frac < 1.0:
        frac = min(1.0, frac * args.cpu_seconds / dt)
        px, dt = run(frac)
    return {"value": round(px / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores,
            "kind": "port",
            "sample": "%.3f%% of the step's candidates (every ladder size, strided), %.1f s, "
                      "OpenMP over candidates" % (100 * frac, dt)}, check


def parity_check(check, outs):
    """Compare the buffers the timed GPU steps wrote with the oracle's results for the
    candidates of the CPU-baseline sample (same planes, same descriptors)."""
    import torch
    n_checked, bad = 0, []
    for s, (idx, sad, satd, co) in check.items():
        ix = torch.from_numpy(idx.astype(np.int64)).cuda()
        g_sad = outs[s]["sad"].index_select(0, ix).cpu().numpy().view(np.uint32)
        g_satd = outs[s]["satd"].index_select(0, ix).cpu().numpy().view(np.uint32)
        g_co = outs[s]["coeffs"].index_select(0, ix).cpu().numpy()
        ok = np.array_equal(g_sad, sad) and np.array_equal(g_satd, satd) and np.array_equal(g_co, co)
        if not ok:
            bad.append(s)
        n_checked += len(idx)
    return n_checked, bad


def pmc_traffic(bd, size, fw, fh, k):
    """HBM bytes per launch of the dominant kernel from the PMC counters
    (FETCH_SIZE / WRITE_SIZE), collected in separate `rocprofv3 --pmc` passes of
    this same workload (tools/gpu_pmc.sh) and summarised by tools/pmc_summary.py
    under profiles/.  A counter pass cannot run inside the timed bench, so the
    figure is read from the committed summary and only reported when it was
    taken on the default workload."""
    import glob
    if (fw, fh, k) != (3840, 2160, 16):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    lg = {64: 6, 32: 5, 16: 4, 8: 3}[size]
    base = "k_rdo_cand<%d,%d,%d,%s" % (bd, lg, lg, "short" if bd == 8 else "int")
    key = next((k for k in (base + ",0>", base + ">") if k in d), None)   # QM = 0: the headline variant
    if key is None or "hbm_traffic_bytes" not in d[key]:
        return None, None
    return int(d[key]["hbm_traffic_bytes"]), (
        "%s: 2*FETCH_SIZE + WRITE_SIZE KiB per dispatch (gfx950 FETCH_SIZE correction, "
        "MI355X_MICROARCH.md); below the algorithmic bytes because the K candidates of a "
        "block share window rows in L2/MALL" % os.path.basename(files[-1]))


def main():
    args = parse()
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
        try:
            comm = tiles.Comm(ctx, rank, world)
            exch_note = "r1_comm_exchange_halos (64 px) + r1_comm_allgather_tiles per step, in stream order"
        except Exception as e:   # keep the scaling run alive; the JSON says which path ran
            exch_note = "torch.distributed all_gather_into_tensor (C-ABI comm failed: %s)" % (str(e)[:80],)
            send, gathered = tiles.make_exchange_buffers(ref.data, rank, world)

    full = args.chain != "cand"
    pixel = args.chain == "pixel"
    if pixel:
        scales = torch.from_numpy(np.random.default_rng(9).integers(
            1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8)).astype(np.int32)).cuda()
        launches = {}
        for s, c in cands.items():
            n = len(c)
            if not n:
                continue
            del outs[s]["coeffs"]
            outs[s].update(eob=torch.empty(n, dtype=torch.int16, device="cuda"),
                           dist=torch.empty(n, dtype=torch.int64, device="cuda"))
            launches[s] = (lambda s=s, n=n: ctx.rdo_pixel_cand_batch(
                org, ref, s, s, dcands[s], args.qindex, 3, scales=scales, n=n, outs=outs[s]))
    elif full:
        for s, c in cands.items():
            n = len(c)
            del outs[s]["coeffs"]
            outs[s].update(eob=torch.empty(n, dtype=torch.int16, device="cuda"),
                           tx_dist=torch.empty(n, dtype=torch.int64, device="cuda"),
                           est_rate=torch.empty(n, dtype=torch.int64, device="cuda"))
        launches = {s: ctx.prepare_rdo_full_cand(org, ref, s, s, dcands[s], len(cands[s]),
                                                 args.qindex, outs[s])
                    for s in cands if len(cands[s])}
    else:
        launches = {s: ctx.prepare_rdo_cand(org, ref, s, s, dcands[s], len(cands[s]), outs[s])
                    for s in cands if len(cands[s])}

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

    # A timing-event pair costs ~0.5 % of a step (the event drains the stream), so every
    # EV_EVERY-th timed step carries them (around each size's launch), the others run bare.
    EV_EVERY = 4
    nstep = [0]

    fan = args.streams > 1
    size_streams = {s: torch.cuda.Stream() for s in W.LADDER} if fan else {}

    def step(timed, exchange=True):
        mark = timed and use_events and nstep[0] % EV_EVERY == 0
        if timed:
            nstep[0] += 1
        main = torch.cuda.current_stream()
        if fan and not mark:
            # independent launches, one stream per block size (each stream is in order with the
            # same size's launch of the previous step, which wrote the same output buffers)
            for s in W.LADDER:
                if len(cands[s]):
                    with torch.cuda.stream(size_streams[s]):
                        launches[s]()
        else:
            for st in size_streams.values():
                main.wait_stream(st)
            for s in W.LADDER:
                n = len(cands[s])
                if n == 0:
                    continue
                if mark:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                launches[s]()
                if mark:
                    e1.record()
                    ev[s].append((e0, e1))
            for st in size_streams.values():
                st.wait_stream(main)
        if world > 1 and exchange:
            for st in size_streams.values():
                main.wait_stream(st)
            if comm is not None:
                comm.exchange_tile_halos(ref, rects)
                comm.allgather_tiles(ref, rects)
            else:
                tiles.exchange_rows(send, gathered)
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
            achieved = abytes / (per[dom] * 1e-3) / 1e9
            kname = "k_rdo_cand<bd=%d,%dx%d%s>" % (bd, dom, dom, ",pixel" if pixel else (",quant" if full else ""))
            traffic, traffic_note = (None, None) if full else pmc_traffic(bd, dom, fw, fh, args.k)
        else:
            dom, per = None, {}
            abytes = sum(abytes_per_cand(s) * len(c) for s, c in cands.items())
            achieved = abytes / (dt / args.steps) / 1e9
            kname = "k_rdo_cand (all sizes, step time)"
            traffic, traffic_note = None, None
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
                       "tiles": world, "exchange": exch_note,
                       "parallelism": "tile-per-gpu x%d" % world if world > 1 else "single-gpu"},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": int(abytes),
                         "avg_launch_ms": round(per[dom], 4) if dom else None},
            "prewarm_steps": prewarm_steps,
            "kernel_ms": {str(s): round(v, 4) for s, v in per.items()},
            "kernel_ms_note": "HIP events around each launch of every %dth timed step "
                              "(%d samples per size)" % (EV_EVERY, max(len(v) for v in ev.values())),
        }
        bad = []
        if world == 1 and args.cpu_seconds > 0 and not full:
            res["cpu_baseline"], check = cpu_baseline(args, host_org, host_ref, cands)
            # the oracle just evaluated a strided sample of this very step at 4K: compare it
            # with what the timed launches left in HBM (sad, satd and every coefficient)
            n_checked, bad = parity_check(check, outs)
            res["parity_checked"] = n_checked
            res["parity_ok"] = not bad
        print(json.dumps(res))
        if bad:
            print("PARITY FAILURE at block sizes %s" % bad, file=sys.stderr)
            sys.exit(3)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
