"""Round-4 experiment (see profiles/r4_two_streams.md): the per-Gaussian colours written by the preprocess kernel of a view on one
HIP stream, checked against a torch evaluation of the same SH formula, while ANOTHER stream of the same GPU runs torch kernels or a
forward of this library.  DBG_VARIANT=0x2000 selects the VALU forward.  python profiles/experiments/r4_two_streams_colour_check.py"""
import sys, os, math, time, torch
sys.path.insert(0, os.getcwd())
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe
from trase_amd import rasterizer as R
from trase_amd.sh import sh_colors_python
from gaussian_renderer import render
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda")
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
pipe = SynthPipe()
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev) for k in range(16)]
bg = torch.zeros(3, device=dev)
P = N
def step(i):
    with torch.no_grad():
        o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)
    g = R._Policy.last_geom
    return g[256 + 24 * P: 256 + 40 * P].clone().view(torch.float32).view(P, 4), o["radii"].clone()
R.set_sync(True)
caps = []
for i in range(16):
    step(i); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
def check(tag, i, rg, radii):
    want = sh_colors_python(pc, cams[i % 16].camera_center).float()     # torch fp32 evaluation of the same formula
    vis = radii > 0
    err = (rg[:, :3] - want).abs().max(1).values * vis
    badrows = (err > 1e-5).nonzero().flatten()
    print(tag, "view", i, "rows off the torch colour by > 1e-5:", badrows.numel(), "lanes", (badrows % 64).unique().tolist()[:20], "first", badrows[:6].tolist(), "max", float(err.max().detach()), flush=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
A = torch.randn(4096, 4096, device=dev); Ab = A.bfloat16(); B = torch.randn(64 << 20, device=dev); C = torch.empty_like(B)
S = torch.randn(1 << 22, device=dev)
def noise(kind):
    if kind == "matmul_f32": torch.mm(A, A)
    elif kind == "matmul_bf16": torch.mm(Ab, Ab)
    elif kind == "exp": torch.exp(B, out=C)
    elif kind == "copy": C.copy_(B)
    elif kind == "sort": torch.sort(S)
    elif kind == "cumsum": torch.cumsum(B, 0, out=C)
    elif kind == "forward": step(3)
if os.environ.get("DBG_VARIANT"): R.set_variant(int(os.environ["DBG_VARIANT"], 0))
for kind in ("none", "matmul_bf16", "exp", "sort", "forward"):
    bad = 0
    for i in range(48):
        with torch.cuda.stream(sb):
            if kind != "none":
                for _ in range(2): noise(kind)
        with torch.cuda.stream(sa):
            rg, radii = step(i)
        with torch.cuda.stream(sb):
            if kind != "none": noise(kind)
        sa.synchronize()
        want = sh_colors_python(pc, cams[i % 16].camera_center).float()
        err = ((rg[:, :3] - want).abs().max(1).values * (radii > 0)).detach()
        bad += int(err.max() > 1e-5)
    torch.cuda.synchronize()
    print("concurrent", kind, ": views with wrong colours", bad, "of 48", flush=True)
