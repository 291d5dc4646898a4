#!/usr/bin/env python
"""Communication-INCLUSIVE model of both multi-GPU axes: bytes per step and predicted step time at the
S4 (300k Gaussians, 1080p) and S5 (2.5 M Gaussians, 1280x960) sizes for a ring and for a direct (all seven xGMI links of a
GPU at once) gradient exchange.  Inputs: single-GPU measurements of this round (bench.py default line, bench.py
--strip-table) and the link figures of SURVEY.md section 5 as encoded in trase_amd/dp.py.  NOTHING here is measured on
multi-GPU hardware; the point of the table is to say what each axis costs once the exchange is counted and which axis
BASELINE config 5 should use.

    python profiles/make_scaling_model.py > profiles/r6_scaling_model.json"""
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
# ---- round 5: the phased exchange (trase_amd.dp.FlatGradBucket.allreduce_phased) against WHOLE iterations ---------------------
# GAUSSIAN state: phase A = xyz (12 B / Gaussian) + the MLP's 2.03 MB, waited on before Adam(xyz, MLP); phase B = the rest
# (f_dc 12, opacity 4, scaling 12, rotation 16 + 12 B per ACTIVE f_rest coefficient row, 15 at full degree) runs underneath the
# next iteration's MLP training forward + the optimizer step of phase A.  FEATURE state: the 128 B / Gaussian feature bucket
# underneath the no_grad MLP forward.  Single-GPU figures of this round: profiles/r5_iteration_breakdown.json.
ITER = {"gaussian_iteration_ms": float(os.environ.get("S4_GAUSSIAN_ITER_MS", "2.05")), "mlp_train_forward_ms": 0.41, "adam_first_ms": 0.02,
        "feature_iteration_ms": float(os.environ.get("S4_FEATURE_ITER_MS", "2.31")), "mlp_inference_forward_ms": 0.31}
MLP_BYTES = 506_378 * 4
n4 = SIZES["S4"]["n"]
ph = {"inputs": ITER, "note": "exposed = phase A + what of phase B outlasts its hiding window; step = iteration + exposed; the un-phased row "
      "is one exchange of the whole state bucket after the backward (round 4's schedule).  MODEL: nothing here ran on two GPUs.", "gaussian_state": {},
      "feature_state": {}}
for deg in (0, 1, 2, 3):
    k = (deg + 1) ** 2 - 1
    a_bytes = n4 * 12 + MLP_BYTES
    b_bytes = n4 * (12 + 4 + 12 + 16 + 12 * k)
    full = n4 * 236 + MLP_BYTES
    rows = {}
    for w in (2, 4, 8):
        rec = {}
        for algo in ("ring", "direct"):
            ta, tb, tf = exchange_model_ms(a_bytes, w, algo), exchange_model_ms(b_bytes, w, algo), exchange_model_ms(full, w, algo)
            window = ITER["mlp_train_forward_ms"] + ITER["adam_first_ms"]
            exposed = ta + max(0.0, tb - window)
            it = ITER["gaussian_iteration_ms"]
            rec[algo] = {"phase_a_ms": round(ta, 3), "phase_b_ms": round(tb, 3), "hidden_ms": round(min(tb, window), 3), "exposed_ms": round(exposed, 3),
                         "efficiency_phased": round(it / (it + exposed), 3), "efficiency_unphased_full_bucket": round(it / (it + tf), 3)}
        rows[str(w)] = rec
    ph["gaussian_state"][f"active_sh_degree_{deg}"] = {"phase_a_bytes": a_bytes, "phase_b_bytes": b_bytes, "unphased_bytes": full, "world": rows}
rows = {}
for w in (2, 4, 8):
    rec = {}
    for algo in ("ring", "direct"):
        tb = exchange_model_ms(n4 * 128, w, algo)
        window = ITER["mlp_inference_forward_ms"]
        it = ITER["feature_iteration_ms"]
        rec[algo] = {"phase_b_ms": round(tb, 3), "hidden_ms": round(min(tb, window), 3), "exposed_ms": round(max(0.0, tb - window), 3),
                     "efficiency_phased": round(it / (it + max(0.0, tb - window)), 3), "efficiency_unphased": round(it / (it + tb), 3)}
    rows[str(w)] = rec
ph["feature_state"] = {"bucket_bytes": n4 * 128, "world": rows}
out["axis_1_phased_exchange_S4_whole_iterations"] = ph
# ---- round 6: the visible-set exchange (trase_amd.dp.FlatGradBucket.allreduce_visible) -----------------------------------------------
# a rank contributes only the rows its view touched (radii > 0) and receives back only the rows ANY rank touched.  Both phases use
# all W-1 links like the direct exchange; per link a rank sends v x (its share of an owner's shard) in the reduce phase and u x (its own
# shard) in the gather phase, v / u = measured visible / union fractions of the bench's camera sets (profiles/r6_visibility.json,
# profiles/measure_visibility.py); + one more collective latency for the all-gather of the bit masks (P / 8 bytes).
vis_path = os.path.join(HERE, "r6_visibility.json")
if os.path.exists(vis_path):
    vis = json.load(open(vis_path))
    link = XGMI_LINK_GBS_PER_DIR * XGMI_LINK_EFFICIENCY * 1e9
    vs = {"inputs": vis, "note": "MODEL (unmeasured on multi-GPU hardware): time = (v + u) x shard_bytes / link + 3 collective latencies; dense direct = "
          "2 x shard_bytes / link + 2 latencies.  The synthetic orbit scenes keep 75 % (S4) / 90 % (S5) of the Gaussians in view: the saving is "
          "what culling leaves -- a scene seen from inside, or a camera rig that splits the scene, has more to give."}
    for name, sz in SIZES.items():
        rows = {}
        for state, bpg in (("GAUSSIAN state", 236), ("FEATURE state", 128)):
            per_w = {}
            for w in (2, 4, 8):
                v, u = vis[name][str(w)]["visible_fraction_per_rank"], vis[name][str(w)]["union_fraction"]
                shard = sz["n"] * bpg / w
                t_vis = (v + u) * shard / link * 1e3 + 3 * COLLECTIVE_LATENCY_MS
                t_dense = exchange_model_ms(sz["n"] * bpg, w, "direct")
                per_w[str(w)] = {"bytes_sent_per_rank_dense": int(2 * shard * (w - 1)), "bytes_sent_per_rank_visible": int((v + u) * shard * (w - 1) + sz["n"] / 8 * (w - 1)),
                                 "exchange_ms_dense_direct": round(t_dense, 3), "exchange_ms_visible": round(t_vis, 3), "ratio": round(t_vis / t_dense, 3),
                                 "weak_scaling_efficiency_dense": round(sz["view_ms"] / (sz["view_ms"] + t_dense), 3),
                                 "weak_scaling_efficiency_visible": round(sz["view_ms"] / (sz["view_ms"] + t_vis), 3)}
            rows[state] = per_w
        vs[name] = rows
    out["axis_1_visible_set_exchange"] = vs
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
