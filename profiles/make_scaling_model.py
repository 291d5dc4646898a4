#!/usr/bin/env python
"""Communication-INCLUSIVE model of both multi-GPU axes: bytes per step and predicted step time at the
S4 (300k Gaussians, 1080p) and S5 (2.5 M Gaussians, 1280x960) sizes for a ring and for a direct (all seven xGMI links of a
GPU at once) gradient exchange.  Inputs: single-GPU measurements of this round (bench.py default line, bench.py
--strip-table) and the link figures of SURVEY.md section 5 as encoded in trase_amd/dp.py.  NOTHING here is measured on
multi-GPU hardware; the point of the table is to say what each axis costs once the exchange is counted and which axis
BASELINE config 5 should use.

    python profiles/make_scaling_model.py > profiles/r4_scaling_model.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.dp import COLLECTIVE_LATENCY_MS, XGMI_LINK_EFFICIENCY, XGMI_LINK_GBS_PER_DIR, exchange_model_ms, recommended_chunks  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
strip = json.load(open(os.path.join(HERE, "r4_strip_scaling.json")))
B_PER_GAUSSIAN = {"all parameters": 364, "GAUSSIAN state": 236, "FEATURE state": 128}
SIZES = {"S4": dict(n=300_000, view_ms=float(os.environ.get("S4_VIEW_MS", "1.12")), tail_ms=0.18, rgb_bytes=3 * 1920 * 1080 * 4),
         "S5": dict(n=2_500_000, view_ms=strip["world"]["1"]["max_ms"], tail_ms=0.75, rgb_bytes=3 * 1280 * 960 * 4)}
out = {"note": "MODEL, not a measurement: single-GPU kernel times measured on one MI355X; exchange times from "
               f"{XGMI_LINK_GBS_PER_DIR} GB/s per direction per xGMI link x {XGMI_LINK_EFFICIENCY} efficiency, "
               f"{COLLECTIVE_LATENCY_MS} ms per collective phase (trase_amd/dp.py exchange_model_ms).  ring = 2(W-1) steps over ONE link "
               "per GPU (RCCL ring all-reduce, or ring reduce-scatter + all-gather); direct = two phases over all W-1 links "
               "(FlatGradBucket exchange='direct').  Unmeasured on multi-GPU hardware.",
       "axis_1_view_parallel": {}, "axis_2_tile_rows_of_one_view": {}}
for name, sz in SIZES.items():
    rows = {}
    for state, bpg in B_PER_GAUSSIAN.items():
        nbytes = sz["n"] * bpg
        per_w = {}
        for w in (2, 4, 8):
            rec = {}
            for algo in ("ring", "direct"):
                ex = exchange_model_ms(nbytes, w, algo)
                k = recommended_chunks(nbytes, w, algo, tail_ms=sz["tail_ms"])
                hidden = min(ex, sz["tail_ms"]) * (k - 1) / k - 0.027 * (k - 1) if k > 1 else 0.0
                step = sz["view_ms"] + ex - hidden
                rec[algo] = {"exchange_ms": round(ex, 3), "overlap_ranges": k, "hidden_ms": round(hidden, 3), "step_ms": round(step, 3),
                             "views_per_s": round(w / step * 1e3, 1), "weak_scaling_efficiency": round(sz["view_ms"] / step, 3)}
            per_w[str(w)] = rec
        rows[state] = {"bucket_bytes": nbytes, "world": per_w}
    out["axis_1_view_parallel"][name] = {"view_ms_one_gpu": sz["view_ms"], "backward_tail_ms_available_for_overlap": sz["tail_ms"], "buckets": rows}
# axis 2: one view per step for the whole job; measured strip times (slowest strip) + gradient exchange of the WHOLE bucket
# (every rank holds partial sums for every Gaussian it touched; the optimizer needs the full sum on every replica) + the
# all-gather of the RGB strips (full-frame losses)
n5 = SIZES["S5"]["n"]
for state, bpg in B_PER_GAUSSIAN.items():
    nbytes = n5 * bpg
    per_w = {}
    for w in (2, 4, 8):
        strip_ms = strip["world"][str(w)]["max_ms"]
        rec = {"slowest_strip_ms_measured_one_gpu": strip_ms}
        for algo in ("ring", "direct"):
            ex = exchange_model_ms(nbytes, w, algo)
            ag = exchange_model_ms(SIZES["S5"]["rgb_bytes"], w, algo) / 2          # one phase: all-gather of the RGB strips
            step = strip_ms + ex + ag
            rec[algo] = {"exchange_ms": round(ex, 3), "rgb_allgather_ms": round(ag, 3), "step_ms": round(step, 3),
                         "speedup_vs_one_gpu": round(SIZES["S5"]["view_ms"] / step, 3)}
        per_w[str(w)] = rec
    out["axis_2_tile_rows_of_one_view"][state] = {"bucket_bytes": nbytes, "one_gpu_view_ms": SIZES["S5"]["view_ms"], "world": per_w}
v8 = out["axis_1_view_parallel"]["S5"]["buckets"]["all parameters"]["world"]["8"]["direct"]
t8 = dict(out["axis_2_tile_rows_of_one_view"]["all parameters"]["world"]["8"]["direct"],
          slowest_strip_ms_measured_one_gpu=out["axis_2_tile_rows_of_one_view"]["all parameters"]["world"]["8"]["slowest_strip_ms_measured_one_gpu"])
out["which_axis_for_config_5"] = (
    f"View parallelism.  At S5 the gradient exchange moves the same {n5 * 364 / 1e6:.0f} MB per step on either axis (every replica needs "
    f"every Gaussian's summed gradient); the view axis amortises it over 8 views per step ({v8['views_per_s']} views/s modelled, direct "
    f"exchange), the tile-row axis over ONE ({1e3 / t8['step_ms']:.0f} views/s; the exchange alone, {t8['exchange_ms']} ms, is longer than a "
    f"rank's strip, {t8['slowest_strip_ms_measured_one_gpu']} ms).  Tile rows are the axis for what has no gradient exchange: rendering one "
    "frame at low latency (inference, GUI, evaluation) -- there the step is the slowest strip plus the all-gather of the strips -- or "
    "for scenes whose per-view working set does not fit one GPU.")
print(json.dumps(out, indent=1))
