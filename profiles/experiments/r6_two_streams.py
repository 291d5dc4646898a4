"""Round-6 bounded experiments on the co-residency corruption (history: profiles/r4_two_streams.md,
r5_two_streams.md).  Victim = the per-Gaussian colours written by `preprocess_fwd_raw` for a view (checked against a torch
evaluation of the same SH formula); aggressor = forward + backward of this library (render_fwd_mf / render_bwd_hw: transposing LDS
reads feeding MFMAs) looping meanwhile.

  mode "streams"  : aggressor on stream A, victim on stream B of ONE process (TRASE_UNORDERED_STREAMS=1: the wrapper's
                    cross-stream wait off).  Run once per library variant (TRASE_RAST_LIB=...): default, noslab, noslp, prio.
  mode "victim"   : the victim alone on the default stream of THIS process, 48 views, while ANOTHER process ("aggressor" mode) loops
                    forward+backward on the same GPU -- the arrangement of tests/test_gpu_bench_ranks.py (ranks sharing a GPU).
  mode "aggressor": loops forward+backward for SECONDS.
Prints one JSON line per run."""
import json, math, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "streams"
if mode == "streams":
    os.environ.setdefault("TRASE_UNORDERED_STREAMS", "1")
import torch
sys.path.insert(0, os.getcwd())
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe
from trase_amd import rasterizer as R, _lib
from trase_amd.sh import sh_colors_python
from gaussian_renderer import render

N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda")
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev) for k in range(16)]
gi, gf = torch.randn(3, H, W, device=dev), torch.randn(F, H, W, device=dev)


def victim(i):
    o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)      # (grad mode on: a no_grad forward does not store the colour array read below)
    g = R._Policy.last_geom
    return g[256 + 24 * N: 256 + 40 * N].clone().view(torch.float32).view(N, 4), o["radii"].clone()


def aggressor(i):
    for p in pc.parameters():
        p.grad = None
    o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)
    torch.autograd.backward([o["render"], o["render_gaussian_features"]], [gi, gf])


R.set_sync(True)
caps = []
for i in range(16):
    victim(i); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
want = [sh_colors_python(pc, cams[i].camera_center).float() for i in range(16)]


def wrong(i, rg, radii):
    err = ((rg[:, :3] - want[i % 16]).abs().max(1).values * (radii > 0)).detach()
    bad = (err > 1e-5).nonzero().flatten()
    return int(bad.numel()), sorted(set((bad % 64).tolist()))


lib = os.path.basename(_lib.LIB_PATH)
if os.environ.get("TRASE_AGG_VARIANT") and mode == "aggressor":      # e.g. 0x2040: the packed-FP32 compositing kernels (no MFMA, no transposing reads)
    R.set_variant(int(os.environ["TRASE_AGG_VARIANT"], 0))
if mode == "aggressor":
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        aggressor(n); n += 1
    torch.cuda.synchronize()
    print(json.dumps({"mode": mode, "iterations": n, "seconds": round(time.time() - t0, 1)}), flush=True)
elif mode == "victim":
    # wait until the other process is running (a flag file), then 48 views on the default stream
    flag = sys.argv[2] if len(sys.argv) > 2 else None
    t0 = time.time()
    while flag and not os.path.exists(flag) and time.time() - t0 < 120:
        time.sleep(0.2)
    views, words, lanes = 0, 0, set()
    t1 = time.time()
    for i in range(48):
        rg, radii = victim(i)
        torch.cuda.synchronize()
        n, l = wrong(i, rg, radii)
        views += int(n > 0); words += n; lanes |= set(l)
    print(json.dumps({"mode": "two processes, victim on its default stream", "lib": lib, "views_wrong": views, "of": 48, "rows_wrong": words,
                      "lanes": sorted(lanes), "seconds": round(time.time() - t1, 2)}), flush=True)
else:
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    for kind in ("none", "fwd+bwd"):
        views, words, lanes = 0, 0, set()
        for i in range(48):
            with torch.cuda.stream(sb):
                if kind != "none":
                    for _ in range(2):
                        aggressor(3)
            with torch.cuda.stream(sa):
                rg, radii = victim(i)
            with torch.cuda.stream(sb):
                if kind != "none":
                    aggressor(3)
            sa.synchronize()
            n, l = wrong(i, rg, radii)
            views += int(n > 0); words += n; lanes |= set(l)
        torch.cuda.synchronize()
        res[kind] = {"views_wrong": views, "of": 48, "rows_wrong": words, "lanes": sorted(lanes)}
    print(json.dumps({"mode": "two streams, one process", "lib": lib, **res}), flush=True)
