#!/usr/bin/env python
"""Lane utilisation of the compositing backward: what share of the evaluated (pixel, Gaussian)
lane slots pass the gates.  Runs the S4 workload (or --gaussians/--width/--height) once with variant 0x8000 (the
counting instantiation of render_bwd_hw_kernel) and prints the header counters as JSON."""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trase_amd import rasterizer as R  # noqa: E402
from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera  # noqa: E402
from gaussian_renderer import render  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=300_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--scale-mult", type=float, default=0.27)
ap.add_argument("--views", type=int, default=4)
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
pc = SynthGaussianModel(make_scene(a.gaussians, feat_dim=32, seed=0, scale_mult=a.scale_mult).to(dev))
bg = torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(1234)
P = a.width * a.height
g_img = (torch.randn(3, a.height, a.width, generator=g) / P).to(dev)
g_feat = (torch.randn(32, a.height, a.width, generator=g) / P).to(dev)
R.set_sync(True)
R.set_variant(0x8000)     # TRASE_VARIANT_AB_COUNT: needs a `make -C trase_amd/csrc AB=1` build of the library
names = ["entries", "chunks", "steps_run", "steps_skipped", "slots_real", "slots_pass_exponent_gates", "slots_blended",
         "walked_pairs_nobody_blended"]
tot = dict.fromkeys(names, 0)
pairs = 0
for k in range(a.views):
    cam = orbit_camera(a.width, a.height, angle=2 * math.pi * k / 16, fid=k / 16).to(dev)
    for p_ in pc.parameters():
        p_.grad = None
    out = render(cam, pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
    torch.autograd.backward([out["render"], out["render_gaussian_features"]], [g_img, g_feat])
    torch.cuda.synchronize()
    hdr = R._Policy.last_geom[:256].view(torch.int32).cpu().tolist()
    for i, n in enumerate(names):
        tot[n] += hdr[40 + i] & 0xffffffff
    pairs += R.last_status()[2]
res = {k: v / a.views for k, v in tot.items()}
res["subtile_pairs"] = pairs / a.views
res["slots_evaluated_incl_tail_lanes"] = res["steps_run"] * 128
res["share_real_of_evaluated"] = res["slots_real"] / max(res["slots_evaluated_incl_tail_lanes"], 1)
res["share_pass_gates_of_real"] = res["slots_pass_exponent_gates"] / max(res["slots_real"], 1)
res["share_blended_of_real"] = res["slots_blended"] / max(res["slots_real"], 1)
res["lane_utilisation"] = res["slots_blended"] / max(res["slots_evaluated_incl_tail_lanes"], 1)
res["steps_skipped_share"] = res["steps_skipped"] / max(res["steps_run"] + res["steps_skipped"], 1)
res["entries_walked_over_pairs"] = res["entries"] / max(res["subtile_pairs"], 1)
res["rows_not_written_share_of_walked"] = res["walked_pairs_nobody_blended"] / max(res["entries"], 1)
res["workload"] = f"{a.gaussians} Gaussians {a.width}x{a.height} F=32 scale_mult {a.scale_mult}, {a.views} views"
js = json.dumps(res, indent=1)
print(js)
if a.out:
    open(a.out, "w").write(js)
