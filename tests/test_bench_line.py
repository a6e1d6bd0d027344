"""bench.py's LAST stdout line is the only thing the driver parses (BENCH_r05.json: `parsed: null` when the line had grown to
20 KB).  compact_line() must turn ANY full record -- here the committed full records of earlier rounds, which carry every leg --
into one strict-JSON line below 4096 bytes that holds the contract's keys with `roofline` and `cpu_baseline`."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# the full (detail) records of the rounds; profiles/rNN_bench_line*.json are compact lines themselves
FULL_RECORDS = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]*_bench*.json")) if "_line" not in os.path.basename(p))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[6-9]*_bench_line*.json")))


def _strict(line):
    def bad(x):
        raise ValueError("non-finite constant %s" % x)
    return json.loads(line, parse_constant=bad)


@pytest.mark.parametrize("path", FULL_RECORDS, ids=[os.path.basename(p) for p in FULL_RECORDS])
def test_compact_line_of_a_full_record(path):
    with open(path) as f:
        full = json.load(f)
    if "metric" not in full:
        pytest.skip("not a bench record")
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) < 4096, len(line)
    d = _strict(line)
    for k in bench.REQUIRED_KEYS:
        assert k in d, k
    assert isinstance(d["dtype"], str) and len(d["dtype"]) <= 16 or path.find("r06") < 0
    assert d["metric"].startswith("aligned cells/sec") and d["unit"] == "cells/s" and d["higher_is_better"] is True
    assert d["value"] == full["value"] and d["n_gpus"] == full["n_gpus"] and d["steps"] == full["steps"] and d["warmup"] == full["warmup"]
    assert abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-3
    rf = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "traffic", "algorithmic_bytes"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    if "cpu_baseline" in full:
        cb = d["cpu_baseline"]
        assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "cells/s"
    cfg = d["config"]
    assert "workload" in cfg and cfg["pairs_total"] == full["config"]["pairs_total"] and "model" not in cfg
    # the scalar summaries of the other legs, when the record has them
    if "search" in full:
        assert abs(d["search_s"] - full["search"]["seconds"]) < 1e-3
    if "configs" in full and "config4_share_1000x87500_verysensitive" in full["configs"]:
        assert d["config4_s"] > 0 and d["config3_s"] > 0 and d["config2_s"] > 0
    if "predicted_scaling" in full:
        assert 0 < d["predicted_eff_n8"] <= 1.0


def test_compact_line_survives_hostile_fields():
    """over-long strings anywhere in the record cannot push the line over the limit; NaN never reaches the line"""
    with open(FULL_RECORDS[-1]) as f:
        full = json.load(f)
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["config"]["windows"] = [[i, i + 1] for i in range(64)]
    full["dtype"] = "int16-exact"
    line = bench.compact_line(full)
    assert len(line.encode()) < 4096
    d = _strict(line)
    assert d["dtype"] == "int16-exact"
    full2 = bench._nan_to_none({"a": float("nan"), "b": [1.0, float("inf")], "c": {"d": float("-inf")}})
    assert full2 == {"a": None, "b": [1.0, None], "c": {"d": None}}


def test_detail_goes_to_earlier_lines_and_a_side_file(tmp_path, capsys):
    with open(FULL_RECORDS[-1]) as f:
        full = json.load(f)
    p = str(tmp_path / "sub" / "bench_detail.json")
    res = bench.emit_detail(full, p)
    out = capsys.readouterr().out
    assert out and all(ln.startswith("bench-detail ") for ln in out.splitlines())      # nothing but the final line starts with '{'
    with open(p) as f:
        assert json.load(f)["value"] == full["value"]
    assert res["detail_file"] == p
    assert json.loads(bench.compact_line(res))["detail"] == p


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_lines_are_what_the_driver_can_parse(path):
    """the last stdout line of the round's final bench runs, as committed: one strict-JSON line below 4 KB with the contract's keys"""
    raw = open(path).read()
    assert raw.count("\n") <= 1 and len(raw.encode()) < 4096
    d = _strict(raw)
    for k in bench.REQUIRED_KEYS:
        assert k in d, k
    assert d["dtype"] == "int16-exact" and d["n_gpus"] == (d["config"]["collective_world"] or 1)
    if d["n_gpus"] == 1:
        assert d["cpu_baseline"]["kind"] == "reference" and d["roofline"]["traffic"] is not None
    else:
        gc = d["config"]["hit_records"]["gather_check"]
        assert gc["gathered_all_ranks"] == gc["one_gpu_hit_records"] == gc["sum_of_rank_counts"]
