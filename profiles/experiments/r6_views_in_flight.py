"""Round 6: with the library built without packed-FP32 VALU (the cure of the co-residency corruption, profiles/r6_two_streams.md),
are two views in flight on two streams of ONE process bit-identical to the serial product path -- and what do they buy?
S4 views, forward + backward, gradients through torch.autograd.grad (no shared .grad accumulation across the streams); every map
and gradient of every view is digested on its own stream and compared with the serial run.  TRASE_UNORDERED_STREAMS=1 (the
wrapper's cross-stream wait off).  python profiles/experiments/r6_views_in_flight.py [reps]"""
import json, math, os, sys, time
os.environ.setdefault("TRASE_UNORDERED_STREAMS", "1")
import torch
sys.path.insert(0, os.getcwd())
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe
from trase_amd import rasterizer as R, _lib
from gaussian_renderer import render
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda")
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
pipe, bg, params = SynthPipe(), torch.zeros(3, device=dev), pc.parameters()
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev) for k in range(16)]
g = torch.Generator().manual_seed(1234)
gi = (torch.randn(3, H, W, generator=g) / (W * H)).to(dev); gf = (torch.randn(F, H, W, generator=g) / (W * H)).to(dev)


def step(i, digest=True):
    o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)
    gr = torch.autograd.grad([o["render"], o["render_gaussian_features"]], params + [o["viewspace_points"]], [gi, gf], allow_unused=True)
    if not digest:
        return None
    ts = [o["render"], o["render_gaussian_features"], o["depth"], o["radii"]] + [t for t in gr if t is not None]
    return torch.stack([t.contiguous().view(torch.int32).to(torch.int64).sum() for t in ts])


R.set_sync(True)
caps = []
for i in range(16):
    step(i, False); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
V = 48
ref = [step(i).cpu() for i in range(V)]
assert all(torch.equal(a, step(i).cpu()) for i, a in enumerate(ref)), "serial runs are not reproducible"


def flight(ns, n, digest):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    out = []
    for i in range(n):
        with torch.cuda.stream(streams[i % ns]):
            out.append(step(i, digest))
    torch.cuda.synchronize()
    return out


res = {"lib": os.path.basename(_lib.LIB_PATH)}
for ns in (2, 3):
    bad = 0
    for r in range(reps):
        got = flight(ns, V, True)
        bad += sum(int(not torch.equal(a, b.cpu())) for a, b in zip(ref, got))
    res[f"streams_{ns}_views_differing"] = f"{bad} of {reps * V}"


def rate(fn, n=64):
    fn(8); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return round(n / (time.perf_counter() - t0), 1)


for r in range(2):
    res.setdefault("views_per_s", []).append({"serial": rate(lambda n: [step(i, False) for i in range(n)]),
                                               "2 streams": rate(lambda n: flight(2, n, False)),
                                               "3 streams": rate(lambda n: flight(3, n, False))})
print(json.dumps(res))
