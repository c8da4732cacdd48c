"""The reference's OWN unit tests, executed from its source text by tools/rustlite -- the
transpiler's self-test.  Every vector in tests/golden/*_ref.npz rests on rustlite executing Rust
the way rustc would; here the code under execution carries its expected values with it
(`assert_eq!` against the tables the reference's authors wrote), so a transpiler that mis-executes
integer arithmetic, slices, iterators or the Plane layout fails them.

Runs only where the reference tree is present (the build container); the GPU box skips it.

  src/dist.rs      get_sad_same_u8 / u16, get_satd_same_u8 / u16   (22 block sizes each, :416-533)
  src/predict.rs   pred_matches_u8, pred_max                        (:1514-1693)
  src/rdo.rs       estimate_rate_test                               (:2749-2752)
  src/cdef.rs      check_max_element                                (:628-660)
  src/quantize/mod.rs  test_divu_pair, test_tx_log_scale, gen_divu_table  (:159-216)
  src/transform/mod.rs log_tx_ratios                                (:521-552)
  src/activity.rs  overflow_test (ssim_boost, :193-216);  src/partition.rs from_wh_matches_naive
  src/tiling/tiler.rs  test_tiling_info_from_tile_count, from_target_tiles_422, tile_log2_overflow (:277-860)
  src/transform/mod.rs roundtrips_u8 / roundtrips_u16               (:479-618; forward_transform ->
                       inverse_transform_add, 44 (size, type) pairs each, the authors' tolerances;
                       `rand::random::<u8>()` is a seeded generator, get_func as in gen_rdo_glue_ref.py)
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")

CASES = [
    ("dist.rs", "get_sad_same_u8"), ("dist.rs", "get_sad_same_u16"),
    ("dist.rs", "get_satd_same_u8"), ("dist.rs", "get_satd_same_u16"),
    ("predict.rs", "pred_matches_u8"), ("predict.rs", "pred_max"),
    ("rdo.rs", "estimate_rate_test"),
    ("cdef.rs", "check_max_element"),
    ("quantize/mod.rs", "test_divu_pair"), ("quantize/mod.rs+transform/mod.rs", "test_tx_log_scale"),
    ("quantize/mod.rs", "gen_divu_table"),
    ("transform/mod.rs", "log_tx_ratios"),
    ("activity.rs", "overflow_test"),
    ("partition.rs", "from_wh_matches_naive"),
    ("tiling/tiler.rs", "test_tiling_info_from_tile_count"), ("tiling/tiler.rs", "from_target_tiles_422"),
    ("tiling/tiler.rs", "tile_log2_overflow"),
]


def crate_with_tests(rel):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rustlite.transpile import Crate
    c = Crate(REF)
    rels = rel.split("+")          # "a.rs+b.rs": b.rs is loaded for the names a.rs's tests glob-import
    for extra in rels[1:]:
        c.load(extra)
    c.load(rels[0], tests=True)
    return c


@pytest.mark.parametrize("rel,name", CASES)
def test_reference_unit_test_passes_under_rustlite(rel, name):
    c = crate_with_tests(rel)
    # the predict.rs tests call generic kernels with u8 buffers: rustc infers T = u8 from the test
    # body's literals ([0u8; 16] / vec![0u16; ..]), the transpiler is told
    g = {"T": "u16" if name == "pred_max" else "u8"} if rel == "predict.rs" else {}
    c.get(name)(g)           # a failing assert! / assert_eq! raises rustlite.runtime.Panic


TRANSFORM_FILES = ["transform/forward.rs", "transform/forward_shared.rs", "transform/inverse.rs", "context/mod.rs",
                   "context/block_unit.rs", "tiling/plane_region.rs", "util/mod.rs", "util/uninit.rs", "frame/mod.rs"]


def transform_crate(seed, tamper=None):
    """transform/mod.rs with its test module, the 2-D drivers of forward.rs / inverse.rs as written;
    get_func (a macro body) = the 1-D networks of the same macro as translated from the text by
    tests/golden/gen_fwd_tx_golden.py; random() = a seeded stand-in for the rand crate."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_fwd_tx_golden as FT
    from rustlite.transpile import Crate
    c = Crate(REF)
    for f in TRANSFORM_FILES:
        c.load(f)
    c.load("transform/mod.rs", tests=True)
    ns, _ = FT.load_reference_1d()

    def get_func(_g, t):
        idx = t.disc if hasattr(t, "disc") else int(t)
        name, n = FT.TXFM[idx], FT.TXFM_LEN[idx]

        def run(coeffs):
            buf = FT.Buf(n)
            for i in range(n):
                buf[i] = FT.V(np.array([coeffs[i]], np.int32))
            ns[name](buf)
            for i in range(n):
                coeffs[i] = int(buf[i].v[0])
        return run
    c.define_py("get_func", get_func)
    rng = np.random.default_rng(seed)
    c.define_py("random", lambda _g: int(rng.integers(0, 256)))
    return c


@pytest.mark.parametrize("name", ["roundtrips_u8", "roundtrips_u16"])
def test_reference_transform_roundtrips(name):
    """forward_transform (forward.rs:71-161) -> inverse_transform_add (inverse.rs:1633-1705), both
    2-D drivers executed as written, meet the tolerances the reference's authors assert."""
    for seed in (1, 2):
        transform_crate(seed).get(name)({})


def test_roundtrip_harness_catches_a_broken_inverse_driver():
    """not vacuous: the same test against an inverse driver whose column-pass rounding shift is
    3 instead of 4 (inverse.rs:1699) panics on the authors' tolerance"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rustlite import runtime as R
    src = open(os.path.join(REF, "transform/inverse.rs")).read()
    a = src.index("pub fn inverse_transform_add<T: Pixel>(")
    b = src.index("/* From AV1 Spec.", a)
    body = src[a:b]
    assert "round_shift(*temp, 4)" in body
    c = transform_crate(3)
    c.load_text("<inverse.rs:1633-1705 with the final shift changed to 3>",
                "pub mod mutated { use super::*; " + body.replace("round_shift(*temp, 4)", "round_shift(*temp, 3)") + "}")
    # the mutated definition shadows the original for callers resolving `inverse_transform_add`
    infos = c.fns["inverse_transform_add"]
    infos.insert(0, infos.pop())
    with pytest.raises(R.Panic):
        c.get("roundtrips_u8")({})


def test_tile_rects_equal_the_executed_tiling_info():
    """rav1e_amd/workload.py::tile_rects (what shards the frame over the GPUs) against the
    reference's TilingInfo::from_target_tiles EXECUTED (src/tiling/tiler.rs:56-150) inside the tile-count
    loop of Sequence::new (src/encoder.rs:248-277, eight lines, restated here): same grid, same tile
    sizes in superblocks, for the frame sizes and tile counts of BASELINE.json and a sweep around them."""
    sys.path.insert(0, ROOT)
    from rav1e_amd import workload as W
    c = crate_with_tests("tiling/tiler.rs")
    ftt = c.get("from_target_tiles", owner="TilingInfo")

    def reference_tiling(w, h, tiles):
        rl = cl = 0
        while True:
            t = ftt({}, 6, w, h, 60.0, cl, rl, False)
            if t.rows * t.cols >= tiles:
                return t
            if not (rl < t.max_tile_rows_log2 or cl < t.max_tile_cols_log2):
                return t
            if (t.tile_height_sb >= t.tile_width_sb and t.tile_rows_log2 < t.max_tile_rows_log2) or \
                    cl >= t.max_tile_cols_log2:
                rl += 1
            else:
                cl += 1
    n = 0
    for (w, h) in ((3840, 2160), (1920, 1080), (1280, 720), (4096, 2304), (640, 360), (3840, 2176), (1000, 600)):
        for tiles in (1, 2, 4, 8, 16):
            t = reference_tiling(w, h, tiles)
            try:
                rects = W.tile_rects(tiles, w, h)
            except ValueError:
                assert t.rows * t.cols != tiles, (w, h, tiles)      # the reference lands on another count too
                continue
            assert t.rows * t.cols == tiles, (w, h, tiles, t.cols, t.rows)
            want = []
            for r in range(t.rows):
                for cc in range(t.cols):
                    x0, y0 = cc * t.tile_width_sb * 64, r * t.tile_height_sb * 64
                    want.append((x0, y0, min(x0 + t.tile_width_sb * 64, w), min(y0 + t.tile_height_sb * 64, h)))
            assert rects == want, (w, h, tiles, rects[:3], want[:3])
            n += 1
    assert n >= 25, n


def test_a_wrong_expectation_is_caught():
    """the harness is not vacuous: the same SATD test with one expected value changed fails"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rustlite import runtime as R
    from rustlite.transpile import Crate
    src = open(os.path.join(REF, "dist.rs")).read()
    a = src.index("fn get_satd_same_inner")
    body = src[a:src.index("#[test]", a)]
    assert "(4, 4, 1408)" in body
    c = Crate(REF)
    c.load("dist.rs", tests=True)
    c.load_text("<dist.rs get_satd_same_inner with one expectation off by one>",
                "pub mod mutated { use super::*; " +
                body.replace("fn get_satd_same_inner", "fn mutated_satd_inner").replace("(4, 4, 1408)", "(4, 4, 1409)") + "}")
    with pytest.raises(R.Panic):
        c.get("mutated_satd_inner")({"T": "u8"})


def test_lrf_search_fixture_follows_the_reference_text(tmp_path):
    """lrf_search_ref.npz is what the reference's text computes, not what the generator assumes: the
    generator run on a scratch copy of src/ with (a) `err * fi.dist_scale[pli]` of rdo_loop_plane_error
    (rdo.rs:2092) reduced to `err` changes every error and no weight; (b) the second weight's clamp
    bound of sgrproj_solve (lrf.rs:1091) moved changes weights.  One small case each."""
    import shutil
    import subprocess
    src = tmp_path / "src"
    shutil.copytree(REF, src)
    gen = os.path.join(ROOT, "tests", "golden", "gen_lrf_search_ref.py")
    base = np.load(os.path.join(ROOT, "tests", "golden", "lrf_search_ref.npz"))

    def run(rel, old, new, tag):
        text = open(os.path.join(REF, rel)).read()
        assert text.count(old) == 1, (rel, old)
        (src / rel).write_text(text.replace(old, new))
        out = tmp_path / tag
        out.mkdir()
        env = dict(os.environ, R1_REF_SRC=str(src), R1_GOLDEN_OUT=str(out), R1_LRF_SEARCH_CASES="2")
        subprocess.run([sys.executable, gen], check=True, env=env, cwd=os.path.dirname(gen),
                       stdout=subprocess.DEVNULL, timeout=600)
        (src / rel).write_text(text)
        return np.load(out / "lrf_search_ref.npz")
    m = run("rdo.rs", "  err * fi.dist_scale[pli]\n}", "  err\n}", "a")
    assert np.array_equal(m["s2_rows"], base["s2_rows"])                     # same units, sets, weights
    assert (m["s2_err"] != base["s2_err"]).mean() > 0.9                      # (nearly) every error moved
    m = run("lrf.rs", "      SGRPROJ_XQD_MAX[1] as i32,\n    );\n    (xqd0 as i8, xqd1 as i8)",
            "      SGRPROJ_XQD_MAX[1] as i32 - 60,\n    );\n    (xqd0 as i8, xqd1 as i8)", "b")
    assert not np.array_equal(m["s2_rows"], base["s2_rows"])                 # weights moved



def test_loop_decision_fixture_follows_the_reference_text(tmp_path):
    """loop_decision_ref.npz is what rdo_loop_decision's OWN text does, loop and all: the generator run on a scratch
    copy of src/ with (a) the visible width of a restoration unit (rdo.rs:2645-2649) shortened by 8 changes the units'
    rows and every error behind them; (b) the CDEF leg's trial loop cut to index 0 (rdo.rs:2394) leaves no error
    for any other index and moves the picks.  One small case each."""
    import shutil
    import subprocess
    src = tmp_path / "src"
    shutil.copytree(REF, src)
    gen = os.path.join(ROOT, "tests", "golden", "gen_loop_decision_ref.py")
    base = np.load(os.path.join(ROOT, "tests", "golden", "loop_decision_ref.npz"))

    def run(old, new, tag, case):
        text = open(os.path.join(REF, "rdo.rs")).read()
        assert text.count(old) == 1, old
        (src / "rdo.rs").write_text(text.replace(old, new))
        out = tmp_path / tag
        out.mkdir()
        env = dict(os.environ, R1_REF_SRC=str(src), R1_GOLDEN_OUT=str(out), R1_LOOP_DECISION_CASES=case)
        subprocess.run([sys.executable, gen], check=True, env=env, cwd=os.path.dirname(gen), stdout=subprocess.DEVNULL,
                       timeout=900)
        (src / "rdo.rs").write_text(text)
        return np.load(out / "loop_decision_ref.npz")
    m = run("              let vis_width = unit_size.min(\n                (crop_w >> xdec)\n",
            "              let vis_width = unit_size.min(\n                (crop_w >> xdec) - 8\n", "a", "ldl2")
    assert m["ldl2_rows"].shape == base["ldl2_rows"].shape
    assert (m["ldl2_rows"][:, 3] != base["ldl2_rows"][:, 3]).any()          # widths of the last unit column moved
    assert (m["ldl2_err"] != base["ldl2_err"]).mean() > 0.3
    m = run("          for cdef_index in 0..(1 << fi.cdef_bits) {", "          for cdef_index in 0..1 {", "b", "ldc2")
    assert np.array_equal(m["ldc2_err"][:, :, 0], base["ldc2_err"][:, :, 0])
    assert not m["ldc2_err"][:, :, 1:].any() and base["ldc2_err"][:, :, 1].any()
    assert not np.array_equal(m["ldc2_best"], base["ldc2_best"])
