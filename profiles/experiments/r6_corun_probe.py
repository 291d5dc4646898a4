"""Premise test: does the MLP training forward (matrix / LDS bound) hide under the render + loss + render backward of a view (VALU / HBM
bound) when they run on two streams of one process?  Serial vs concurrent, same work."""
import sys, os, json, time, math, torch
sys.path.insert(0, os.getcwd())
from trase_amd import rasterizer as R
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe, SynthDeformNetwork
from trase_amd.deform import deform_forward
from trase_amd.losses import photometric_loss
from trase_amd.renderer import render, set_forward_scope
R.set_stream_ordering(False)
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda", 0); torch.manual_seed(0)
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
net = SynthDeformNetwork().to(dev)
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 8, fid=k / 8).to(dev) for k in range(8)]
bg = torch.zeros(3, device=dev); gt = torch.rand(3, H, W, device=dev); pipe = SynthPipe()
x = pc.get_xyz.detach(); t = torch.tensor([[0.4]], device=dev).expand(N, -1)
d0 = [torch.zeros(N, 3, device=dev), torch.zeros(N, 4, device=dev), torch.zeros(N, 3, device=dev)]
live = dict(net.named_parameters())
def render_step(i):
    for p in pc.parameters(): p.grad = None
    set_forward_scope("image")
    try: out = render(cams[i % 8], pc, pipe, bg, d0[0], d0[1], d0[2])
    finally: set_forward_scope("all")
    photometric_loss(out["render"], gt, 0.2).backward()
keep = []
def mlp_step():
    keep.clear(); keep.append(deform_forward(live, x, t))          # training forward: saves the state
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(mode, reps=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        if mode == "render": render_step(i)
        elif mode == "mlp": mlp_step()
        elif mode == "serial": render_step(i); mlp_step()
        else:
            with torch.cuda.stream(sa): render_step(i)
            with torch.cuda.stream(sb): mlp_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for m in ("render", "mlp", "serial", "two"): run(m, 10)
res = {}
for rep in range(2):
    for m in ("render", "mlp", "serial", "two"): res.setdefault(m, []).append(round(run(m), 4))
print(json.dumps(res))
