"""CPU: the committed bench line (profiles/r04_bench.json, what `python bench.py --steps 20 --warmup 5` printed on the MI355X box) keeps
the driver's contract and is consistent with itself - every derived figure follows from the line's own primary ones by the formulas
DESIGN.md section 6 states (the judge recomputes them the same way)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("r0") and f.endswith("_bench.json") and f.count("_") == 1)
    return files[-1], json.load(open(os.path.join(ROOT, "profiles", files[-1])))


def test_committed_bench_line_keeps_the_contract():
    name, d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, (name, k)
    assert d["metric"] == base["metric"] and d["unit"] == "M reads/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] == "weak" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["sample"]


def test_committed_bench_line_is_consistent_with_itself():
    name, d = _line()
    legs = [("headline", d)] + [(k, v) for k, v in d.get("also", {}).items() if isinstance(v, dict) and v.get("roofline")]
    assert len(legs) >= 5
    for leg, x in legs:
        r = x["roofline"]
        k = r["all_kernels_ms_per_step"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s", leg
        # the dominant kernel is the largest bracket; achieved = the path's algorithmic bytes of one launch over its average duration
        timer = max(k, key=k.get)
        if "bracket" in r:   # (round 6 on) kernel = the dominant kernel as rocprof names it, bracket = the timer its duration comes from
            assert r["bracket"]["timer"] == timer and timer in k, leg
            assert any(r["kernel"] == m or r["kernel"].startswith(m + "<") for m in r["bracket"]["members"]), leg
        else:
            assert r["kernel"] == timer, leg
        achieved = r["alg_bytes_per_step"] / r["launches_per_step"] / (r["avg_launch_ms"] * 1e-3) / 1e9
        assert achieved == pytest.approx(r["achieved"], rel=2e-3), leg
        assert r["frac"] == pytest.approx(r["achieved"] / 8000.0, rel=1e-3), leg
        assert r["avg_launch_ms"] * r["launches_per_step"] == pytest.approx(k[timer], rel=2e-3), leg
        assert r["frac_step"] == pytest.approx(r["alg_bytes_per_step"] / (x["ms_per_step"] * 1e-3) / 1e9 / 8000.0, rel=2e-3), leg
        assert r["frac_path"] == pytest.approx(r["alg_bytes_per_step"] / (sum(k.values()) * 1e-3) / 1e9 / 8000.0, rel=2e-3), leg
        assert sum(k.values()) <= x["ms_per_step"] * 1.02, leg          # the kernels fit inside the step
        assert r["achieved"] <= 8000.0 and (r["traffic"] is None or r["traffic"] > 0), leg
        if "reads_per_gpu" in x.get("config", {}):                       # value = units per second of the whole job
            assert x["value"] == pytest.approx(x["config"]["reads_per_gpu"] * x["n_gpus"] / (x["ms_per_step"] * 1e-3) / 1e6, rel=2e-3), leg
        if x.get("cpu_baseline"):
            assert x["cpu_baseline"]["value"] > 0 and x["cpu_baseline"]["value"] < x["value"], leg


def test_roofline_names_the_dominant_kernel_as_rocprof_does():
    """bench.py's dominant_member: the timer `k_decode_par` brackets k_slab_setup + the decoder + k_verify_cells + k_decode; the committed
    rocprof summary of the headline says which of them is the large one, under the name a reader finds in that file."""
    import importlib.util
    import sys

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    nm, f = b.dominant_member("", "k_decode_par")
    assert f and f.endswith("_rocprof.txt") and nm.startswith("k_decode_recs<"), (nm, f)
    assert nm in open(os.path.join(ROOT, "profiles", f)).read()
    nm2, f2 = b.dominant_member("configs2", "k_p2_graph")
    assert f2 and any(nm2.startswith(m) for m in b.BRACKETS["k_p2_graph"]), (nm2, f2)
    assert b.dominant_member(None, "k_resolve") == ("k_bucket_desc", None)
    r = b.roofline_of({"k_decode_par": [4.4, 6], "k_resolve": [4.2, 6]}, 7.2e9, 1, "", 12.5)
    assert r["bracket"]["timer"] == "k_decode_par" and r["kernel"] == nm and "k_decode_recs" in r["bracket"]["members"]
