"""The per-table-entry dispatch symbols (include/rav1e_amd_dispatch.h, 711 of them, generated
by tools/gen_dispatch.py) through a compiled C program -- dlsym + the reference's fn-pointer
types, not ctypes (tests/c/test_dispatch.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "rav1e_amd", "librav1e_hip.so")


def build(tmp_path):
    import oracle_lib
    oracle_lib.build()
    exe = str(tmp_path / "test_dispatch")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-o", exe, os.path.join(ROOT, "tests", "c", "test_dispatch.c"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lr1oracle", "-ldl",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return exe


def test_every_dispatch_symbol_is_exported(tmp_path):
    """no GPU needed: the library loads and dlsym finds all 711 entries"""
    out = subprocess.run([build(tmp_path), SO, "list"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("711 symbols"), out.stdout


def test_generated_files_are_current(tmp_path):
    """the generator writes into a scratch directory; the tracked files are only read"""
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "gen_dispatch.py"), str(tmp_path)],
                          stdout=subprocess.DEVNULL)
    for f in ("include/rav1e_amd_dispatch.h", "rav1e_amd/csrc/dispatch_gen.inc", "tests/c/dispatch_list.h"):
        assert open(os.path.join(ROOT, f)).read() == open(os.path.join(str(tmp_path), f)).read(), \
            f + " is stale: run tools/gen_dispatch.py"


@pytest.mark.gpu
def test_dispatch_entries_match_the_oracle(tmp_path):
    out = subprocess.run([build(tmp_path), SO], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "0 failures" in out.stdout
