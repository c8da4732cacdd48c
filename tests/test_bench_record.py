"""bench.py's stdout contract: ONE compact JSON line the driver can parse (the 27 KB
line of round 5 left BENCH_r05.json.parsed = null).  Runs the record builder on the committed long
document of round 5 and on a padded worst case; no GPU, nothing measured."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _long():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_v8_bench.json")))


def test_compact_record_round_trips_and_fits():
    res = _long()
    s = bench.compact_record(res, "gpurun_out/bench_detail.json")
    assert "\n" not in s and len(s) < bench.COMPACT_LIMIT == 4096
    d = json.loads(s)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == res["value"] and d["ms_per_step"] == res["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert d["roofline"][k] == res["roofline"][k]
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"]
    assert d["cpu_baseline"]["kind"] in ("port", "reference")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert [e["name"] for e in d["extra_lines"]] == [e["name"] for e in res["extra_lines"]]
    assert all(isinstance(e["value"], float) and e["parity_ok"] is True for e in d["extra_lines"])
    # numbers, no prose: no string in an extra line longer than a name
    for e in d["extra_lines"]:
        assert all(len(v) < 40 for v in e.values() if isinstance(v, str))


def test_compact_record_sheds_weight_instead_of_growing():
    res = _long()
    pad = copy.deepcopy(res["extra_lines"][-1])
    for i in range(40):     # 50 extra lines with 60 stages each: still one parseable line under the limit
        e = copy.deepcopy(pad)
        e["name"] = "padding_line_%02d" % i
        e["stage_ms"] = {"stage_with_a_long_name_%02d" % j: 0.123456 for j in range(60)}
        res["extra_lines"].append(e)
    s = bench.compact_record(res)
    assert len(s) < bench.COMPACT_LIMIT
    d = json.loads(s)
    for k in CONTRACT:
        assert k in d, k


def test_emit_prints_one_stdout_line(capsys, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_long())
    out, err = capsys.readouterr()
    lines = out.strip().split("\n")
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["detail"] == "gpurun_out/bench_detail.json"
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))["value"] == _long()["value"]
    assert "extra_line " in err and "detail " in err


def test_bench_functions_that_use_torch_import_it():
    """bench.py keeps torch out of its module scope (the record builder is imported without it); a function that names
    `torch` must import it itself -- a missing import only shows on the GPU box, at the end of a 30 s run"""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and any(a.name.split(".")[0] == "torch" for a in n.names)
                   for n in tree.body)
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.Name) and n.id == "torch" for n in ast.walk(fn))
        imports = any(isinstance(n, ast.Import) and any(a.name == "torch" for a in n.names) for n in ast.walk(fn))
        assert not uses or imports, fn.name
