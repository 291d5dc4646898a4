#!/usr/bin/env python
"""Where one GAUSSIAN-state iteration (trase_amd.bench_iterations.make_gaussian_iteration, S4 size) spends its time:
host-clock ms per iteration, the library's per-kernel HIP-event times (ProfScope), and their sum -- the difference is
torch glue + launch gaps.  python profiles/iteration_breakdown.py [image|all|feature]   (feature: one FEATURE-state iteration)"""
import sys, os, math, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd import rasterizer as R
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel
from trase_amd.bench_iterations import make_feature_iteration, make_gaussian_iteration, time_iterations

scope = sys.argv[1] if len(sys.argv) > 1 else "image"
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 8, fid=k / 8).to(dev) for k in range(8)]
if scope == "feature":
    it, _restore = make_feature_iteration(pc, cams, W, H, dev)
else:
    it = make_gaussian_iteration(pc, cams, W, H, dev, image_scope=(scope == "image"))
R.set_sync(True)
caps = []
for i in range(8):
    it(i); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
ms = time_iterations(it, iters=16, warm=40)
R.profile_enable(1)
for i in range(8):
    it(i)
torch.cuda.synchronize()
prof = R.profile_report(); R.profile_enable(0)
ks = {k: round(v["ms"], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
print(json.dumps({"scope": scope, "iteration_ms": round(ms, 3), "library_kernels_ms_sum": round(sum(ks.values()), 3), "kernels_ms": ks}))
