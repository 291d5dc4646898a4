#!/usr/bin/env python
"""Inputs of the visible-set exchange model (trase_amd.dp.FlatGradBucket.allreduce_visible): for the bench's camera sets (rank r of a
W-rank job renders orbit view k + 0.37 r of 16 at step k, bench.py), the fraction of Gaussians a rank's view touches (radii > 0: what
it SENDS in the reduce phase) and the fraction touched by ANY rank of the step (the union: what comes back in the gather phase), at
S4 (300k, 1920x1080) and S5 (2.5 M, 1280x960).  python profiles/measure_visibility.py > profiles/r6_visibility.json"""
import json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd import rasterizer as R
from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
from gaussian_renderer import render
dev = torch.device("cuda", 0)
out = {}
for name, (N, W, H) in {"S4": (300_000, 1920, 1080), "S5": (2_500_000, 1280, 960)}.items():
    pc = SynthGaussianModel(make_scene(N, feat_dim=32, seed=0, scale_mult=0.27).to(dev), requires_grad=False)
    pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
    R.set_sync(True)
    rec = {}
    for world in (2, 4, 8):
        vs, us = [], []
        for k in range(0, 16, 4):
            masks = []
            for r in range(world):
                cam = orbit_camera(W, H, angle=2 * math.pi * (k + r * 0.37) / 16, fid=k / 16).to(dev)
                with torch.no_grad():
                    masks.append(render(cam, pc, pipe, bg, 0.0, 0.0, 0.0)["radii"] > 0)
            m = torch.stack(masks)
            vs.append(float(m.float().mean()))
            us.append(float(m.any(dim=0).float().mean()))
        rec[str(world)] = {"visible_fraction_per_rank": round(sum(vs) / len(vs), 4), "union_fraction": round(sum(us) / len(us), 4)}
    out[name] = rec
    del pc
print(json.dumps(out))
