"""rustlite: a Rust-subset -> Python transpiler (TEST INFRASTRUCTURE).

The build container has no Rust toolchain, so the reference (xiph/rav1e)
cannot be compiled.  To pin the CPU oracle to what the reference itself
computes, the golden-vector generators under tests/golden/ read the
reference's .rs files WHERE THEY LIE under /root/reference, translate the
functions they need to Python with this package, and execute the result.
The vectors are therefore produced by the reference's own statements,
constants and tables -- not by a restatement of ours.

Nothing here is product code: it is imported only by tests/golden/gen_*.py
(run in the build container; the GPU box never needs it).

  lexer.py      tokens
  parser.py     Rust subset -> AST (items eagerly, function bodies lazily)
  transpile.py  AST -> Python source, with the crate-level item registry
  runtime.py    slices / raw pointers / iterators / Option / int helpers,
                and the hand-written stand-ins for the few third-party types
                (v_frame::Plane, PlaneSlice) the reference code touches
"""
try:
    from .transpile import Crate  # noqa: F401
except ImportError:  # during bring-up
    pass
