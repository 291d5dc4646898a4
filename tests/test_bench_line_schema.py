"""The contract of bench.py's JSON line, checked on the line committed with the round's evidence (profiles/r6_bench_default.json --
produced by `python bench.py` on an MI355X; this test needs no GPU): every key the driver and the judge read is there, the roofline
object is self-consistent, the CPU baseline says what it timed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r6_bench_default.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_required_keys_and_types():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "views/s" and "1080p" in d["metric"] and "300k" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["data"] == "synthetic" and d["dtype"].startswith("f32")
    assert abs(d["value"] - d["n_gpus"] * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]          # whole-job throughput of the timed steps
    assert isinstance(d.get("preroll_steps"), int)


def test_roofline_object_is_self_consistent():
    r = _line()["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-2 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes"]
    assert "forward" in r and "compositing_valu_frac" in r and "bound_note" in r


def test_cpu_baseline_object():
    c = _line()["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and len(c["sample"]) > 20


def test_round6_keys():
    """the precision price, two views in flight (checked against the serial run in the same run),
    the whole-iteration graph replay and the spread of the window's own steps are ON the line."""
    d = _line()
    f = d["fp32_variant"]
    assert f["views_per_s"] > 0 and 0.3 < f["vs_headline"] < 1.2 and set(f["max_map_diff_view0"]) == {"image", "features", "depth"}
    assert max(f["max_map_diff_view0"].values()) < 1e-4
    v = d["views_in_flight_2"]
    assert v["views_differing_from_serial"] == 0 and v["views_checked"] >= 16 and v["views_per_s"] > 0
    s = d["step_ms"]
    assert s["min"] <= s["median"] <= s["max"] and abs(s["median"] - d["ms_per_step"]) < 0.2 * d["ms_per_step"]
    g = d["iteration_ms"]["whole_iteration_graph_replay"]
    assert g["gaussian"] > 0 and g["feature"] > 0 and "error" not in g
