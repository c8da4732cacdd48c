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
"""
import os
import sys

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
