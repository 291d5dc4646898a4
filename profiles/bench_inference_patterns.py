#!/usr/bin/env python
"""Forward-only render() at S4 (300k Gaussians, 1920x1080, F = 32) for the inference call patterns of the reference
(render.py:188-397, gui.py:927-1128): plain, override_color, mask, override_color + mask, is_6dof -- the fused path (round 6)
against the operator-level composition these calls took until round 5 (`_fusable` forced to False).  ms per call, host clock
around 20 calls after 10 warm-up calls, device synchronised on both sides; under torch.no_grad() as the reference's render scripts.
python profiles/bench_inference_patterns.py > profiles/r6_inference_patterns.json"""
import json, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd import rasterizer as R, renderer
from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera

N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev), requires_grad=False)
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 8, fid=k / 8).to(dev) for k in range(8)]
pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(3)
oc = torch.rand(N, 3, generator=g).to(dev)
mask = (torch.rand(N, generator=g) < 0.7).to(dev)
d = [(0.003 * torch.randn(N, c, generator=g)).to(dev) for c in (3, 4, 3)]
T44 = (torch.eye(4).repeat(N, 1, 1) + 0.002 * torch.randn(N, 4, 4, generator=g)).to(dev)
T44[:, 3, :3] = 0
PAT = {"plain": dict(), "override_color": dict(override_color=oc), "mask": dict(mask=mask),
       "override_color+mask": dict(override_color=oc, mask=mask), "is_6dof": dict(is_6dof=True)}
real = renderer._fusable


def run(kw, i):
    dx = T44 if kw.get("is_6dof") else d[0]
    with torch.no_grad():
        return renderer.render(cams[i % 8], pc, pipe, bg, dx, d[1], d[2], **kw)


def ms(kw, n=20, warm=10):
    for i in range(warm):
        run(kw, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        run(kw, i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {"workload": f"{N} Gaussians, {W}x{H}, F={F}, forward only under no_grad, sync capacity policy (exact pair count read per call, as the reference)",
       "fused_ms": {}, "operator_composition_ms": {}}
R.set_sync(True)
for name, kw in PAT.items():
    renderer._fusable = real
    a = run(kw, 0)
    out["fused_ms"][name] = round(ms(kw), 4)
    renderer._fusable = lambda *a_, **k_: False
    b = run(kw, 0)
    out["operator_composition_ms"][name] = round(ms(kw), 4)
    out.setdefault("max_abs_image_diff", {})[name] = float((a["render"] - b["render"]).abs().max())
renderer._fusable = real
# the sync-free policy (capacity known beforehand), fused: what a render loop over a trained scene gets
caps = []
for i in range(8):
    run({}, i); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
out["fused_sync_free_ms"] = {name: round(ms(kw), 4) for name, kw in PAT.items() if "mask" not in name}
R.set_sync(True)
print(json.dumps(out))
