#!/usr/bin/env python3
"""Convert the reference's plumbing clip (BASELINE.json configs[0]:
/root/reference/tests/small_input.y4m, 64x64 4:2:0 8-bit, 5 frames) into
tests/golden/small_input_frames.npz.  /root/reference does not travel to the
GPU box, the planes do.  Run in the build container:
    python tests/golden/gen_small_input.py
"""
import os

import numpy as np

SRC = "/root/reference/tests/small_input.y4m"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_input_frames.npz")


def parse_y4m(path):
    raw = open(path, "rb").read()
    nl = raw.index(b"\n")
    hdr = raw[:nl].split()
    assert hdr[0] == b"YUV4MPEG2"
    tags = {t[:1]: t[1:] for t in hdr[1:]}
    w, h = int(tags[b"W"]), int(tags[b"H"])
    assert tags[b"C"].startswith(b"420"), tags[b"C"]
    cw, ch = (w + 1) // 2, (h + 1) // 2
    pos, ys, us, vs = nl + 1, [], [], []
    while pos < len(raw):
        e = raw.index(b"\n", pos)
        assert raw[pos:e].startswith(b"FRAME")
        pos = e + 1
        for lst, (pw, ph) in ((ys, (w, h)), (us, (cw, ch)), (vs, (cw, ch))):
            lst.append(np.frombuffer(raw, np.uint8, pw * ph, pos).reshape(ph, pw).copy())
            pos += pw * ph
    return np.stack(ys), np.stack(us), np.stack(vs)


if __name__ == "__main__":
    y, u, v = parse_y4m(SRC)
    np.savez_compressed(OUT, y=y, u=u, v=v)
    print(OUT, y.shape, u.shape, v.shape)
