#!/usr/bin/env python3
"""One coded 4K frame's device-resident work, end to end (the stages of tools/frame_stages.py, the same list and the
same timing scheme -- HIP events per stage inside whole passes -- as bench.py's `config4_frame_4k_10bit` line), then
the two-stream plan (the ME of the next frame beside the other stages of this one).  Prints one JSON line.

    python tools/frame_pipeline.py [--bit-depth 8] [--reps 20] [--verify]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sustain-ms", type=float, default=300.0,
                    help="untimed passes for this long before the timed ones (sustained clocks); 0 = off")
    ap.add_argument("--verify", action="store_true", help="the per-stage parity samples against the CPU oracle first")
    ap.add_argument("--stages", default="", help="A/B runs: time only the stages whose name contains one of these comma-separated "
                                                  "substrings (the ME still runs once first: later stages read its MEStats)")
    args = ap.parse_args()
    import torch
    import frame_stages
    from rav1e_amd import workload as W
    from rav1e_amd.api import Context
    ctx = Context(0)
    F = frame_stages.build(ctx, args.bit_depth)
    fw, fh, bd = F["frame"]
    parity = None
    if args.verify:
        parity = {n: {"checked": c, "ok": ok} for n, (c, ok) in F["verify"]().items()}
    else:
        for _, fn in F["stages"][:4]:      # the ME fills the MEStats the later stages read
            fn()
    sel = [s for s in args.stages.split(",") if s]
    timed_stages = [(n, f) for n, f in F["stages"] if not sel or any(s in n for s in sel)]
    per, wall = frame_stages.time_stages(timed_stages, args.reps, 5, args.sustain_ms)
    stages = {n: round(v, 4) for n, v in per.items() if not n.endswith("_untimed")}
    assert ctx.me_status(wait=True)[0], "a persistent tile-ME launch flagged a timed-out wait"
    total = round(sum(stages.values()), 3)
    if sel:
        print(json.dumps({"frame": "%dx%d %d-bit 4:2:0" % (fw, fh, bd), "stage_ms": stages}))
        ctx.close()
        return
    ov_fn = F["overlapped"]
    ov_fn()
    torch.cuda.synchronize()
    W.sustain_clocks(ov_fn, args.sustain_ms)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ov_fn()
    torch.cuda.synchronize()
    ov = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
    p4 = F["pipelined4"]
    p4()
    torch.cuda.synchronize()
    W.sustain_clocks(p4, args.sustain_ms)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        p4()
    torch.cuda.synchronize()
    ov4 = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
    assert ctx.me_status(wait=True)[0]
    out = {"frame": "%dx%d %d-bit 4:2:0" % (fw, fh, bd), "stage_ms": stages, "sum_ms": total, "wall_ms_per_pass": round(wall, 3),
           "four_stream_ms (ME + pre-screens | luma chain | chroma chain + type search | post-filter decisions)": ov4,
           "frames_per_s_four_streams": round(1e3 / ov4, 1),
           "frames_per_s_if_serial": round(1e3 / total, 1),
           "two_stream_ms (ME of the next frame beside the other stages)": ov,
           "frames_per_s_two_streams": round(1e3 / ov, 1)}
    out["loop_decision"] = F.get("loop_decision")
    if parity is not None:
        out["parity"] = parity
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
