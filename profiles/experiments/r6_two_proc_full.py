"""Two PROCESSES sharing one GPU, whole views compared (round 6): the victim process renders 48 views forward +
backward alone (baseline digests of every map and gradient), then again while an aggressor process loops forward + backward on the
same GPU, and counts the views whose digests differ.  Each process uses its own default stream -- the arrangement of
tests/test_gpu_bench_ranks.py (ranks sharing a GPU).  Run once per library (TRASE_RAST_LIB) to compare builds.
  python r6_two_proc_full.py victim   |   python r6_two_proc_full.py aggressor SECONDS"""
import json, math, os, sys, time
mode = sys.argv[1]
import torch
sys.path.insert(0, os.getcwd())
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe
from trase_amd import rasterizer as R, _lib
from gaussian_renderer import render
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda")
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev) for k in range(16)]
g = torch.Generator().manual_seed(5)
gi, gf = (torch.randn(3, H, W, generator=g) / (W * H)).to(dev), (torch.randn(F, H, W, generator=g) / (W * H)).to(dev)


def step(i):
    for p in pc.parameters():
        p.grad = None
    o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)
    torch.autograd.backward([o["render"], o["render_gaussian_features"]], [gi, gf])
    ts = [o["render"], o["render_gaussian_features"], o["depth"], o["radii"], o["viewspace_points"].grad] + [p.grad for p in pc.parameters()]
    return [int(t.contiguous().view(torch.int32).to(torch.int64).sum()) for t in ts]


R.set_sync(True)
caps = []
for i in range(16):
    step(i); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
lib = os.path.basename(_lib.LIB_PATH)
if mode == "aggressor":
    while not os.path.exists("/tmp/start_agg"):
        time.sleep(0.1)
    for i in range(50):
        step(i)
    torch.cuda.synchronize()
    open("/tmp/agg_up", "w").close()
    t0, n = time.time(), 0
    while time.time() - t0 < float(sys.argv[2]):
        step(n); n += 1
    torch.cuda.synchronize()
else:
    base = [step(i) for i in range(48)]
    again = [step(i) for i in range(48)]
    assert base == again, "the victim is not reproducible on an idle GPU"
    open("/tmp/start_agg", "w").close()
    t0 = time.time()
    while not os.path.exists("/tmp/agg_up") and time.time() - t0 < 120:
        time.sleep(0.1)
    bad, first = 0, None
    t1 = time.time()
    for rep in range(2):
        for i in range(48):
            d = step(i)
            if d != base[i]:
                bad += 1
                first = first or [k for k, (a, b) in enumerate(zip(d, base[i])) if a != b]
    print(json.dumps({"lib": lib, "views": 96, "views_differing_from_the_solo_run": bad, "first_differing_tensors": first,
                      "aggressor_seen": os.path.exists("/tmp/agg_up"), "seconds": round(time.time() - t1, 2)}), flush=True)
