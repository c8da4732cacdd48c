"""Helpers shared by the gen_*_ref.py generators (TEST INFRASTRUCTURE).

The generators execute the reference's own Rust source text through
tools/rustlite (a Rust-subset -> Python transpiler) in the build container and
write the resulting vectors to tests/golden/*_ref.npz.  They import nothing
from oracle/ and nothing from tests/*_util.py: the numbers in the .npz files
come from the reference's statements alone.  The GPU box only reads the .npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from rustlite import runtime as R  # noqa: E402
from rustlite.transpile import Crate  # noqa: E402

# R1_REF_SRC: point the generators at a scratch COPY of the reference's src/ (mutation checks:
# change one constant in the copy, regenerate into a temp dir, see the fixture change)
REF_SRC = os.environ.get("R1_REF_SRC", "/root/reference/src")


def crate(*files, strict=False):
    if not os.path.isdir(REF_SRC):
        raise SystemExit("the reference tree is not present: run this in the build container")
    c = Crate(REF_SRC, strict=strict)
    for f in files:
        c.load(f)
    return c


# Types of the un-vendored v_frame 0.3.9 crate that reference code names (Cargo.lock:2075-2077);
# restated, not executed: variant order as in v_frame::pixel::ChromaSampling.
V_FRAME_TEXT = """
pub enum ChromaSampling { Cs420, Cs422, Cs444, Cs400 }
"""


def load_v_frame_types(c):
    c.load_text("<v_frame 0.3.9: ChromaSampling>", V_FRAME_TEXT)


def pixel_type(bd):
    return {"T": "u8" if bd == 8 else "u16"}


def np_dtype(bd):
    return np.uint8 if bd == 8 else np.uint16


def plane_from_array(a, bd, xpad=0, ypad=0, xdec=0, ydec=0, pad_value=None):
    """v_frame Plane holding `a` in its visible area; padding filled with pad_value
    (or left at the constructor's fill)."""
    h, w = a.shape
    p = R.Plane.new(w, h, xdec, ydec, xpad, ypad, 1 if bd == 8 else 2)
    cfg = p.cfg
    if pad_value is not None:
        p.data[:] = [int(pad_value)] * len(p.data)
    for y in range(h):
        base = (cfg.yorigin + y) * cfg.stride + cfg.xorigin
        p.data[base:base + w] = [int(v) for v in a[y]]
    return p


def plane_from_padded(a, bd, xpad, ypad, xdec=0, ydec=0):
    """`a` has shape (h + 2*ypad, w + 2*xpad): visible area plus real padding pixels."""
    H, W = a.shape
    h, w = H - 2 * ypad, W - 2 * xpad
    p = R.Plane.new(w, h, xdec, ydec, xpad, ypad, 1 if bd == 8 else 2)
    cfg = p.cfg
    for y in range(H):
        base = (cfg.yorigin - ypad + y) * cfg.stride + cfg.xorigin - xpad
        p.data[base:base + W] = [int(v) for v in a[y]]
    return p


def plane_to_array(p, dtype):
    cfg = p.cfg
    out = np.zeros((cfg.height, cfg.width), dtype)
    for y in range(cfg.height):
        base = (cfg.yorigin + y) * cfg.stride + cfg.xorigin
        out[y] = p.data[base:base + cfg.width]
    return out


def enum(c, ename, vname):
    if ename not in c.enums:
        c.autoload(ename)
    return c.G[c.enum_value(ename, vname)[0]]


def struct(c, name):
    if "S_" + name not in c.G:
        c.autoload(name)
    return c.G["S_" + name]


def save(name, out):
    path = os.path.join(os.environ.get("R1_GOLDEN_OUT", HERE), name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
