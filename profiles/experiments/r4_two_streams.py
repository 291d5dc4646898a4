"""Round-4 experiment (see profiles/r4_two_streams.md).  python profiles/experiments/r4_two_streams.py [steps] -- S4 view fwd+bwd: serial vs round-robin over 2 / 3 / 4 HIP streams (views in flight)."""
import sys, os, math, time, torch
sys.path.insert(0, os.getcwd())
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe
from trase_amd import rasterizer as R
from gaussian_renderer import render
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda")
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
pipe = SynthPipe(); params = pc.parameters()
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev) for k in range(16)]
bg = torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(1234)
gi = (torch.randn(3, H, W, generator=g) / (W * H)).to(dev); gf = (torch.randn(F, H, W, generator=g) / (W * H)).to(dev)
def step(i):
    for p in params: p.grad = None
    o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)
    torch.autograd.backward([o["render"], o["render_gaussian_features"]], [gi, gf])
R.set_sync(True)
caps = []
for i in range(16):
    step(i); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
def run(ns, keep=None):
    streams = [torch.cuda.Stream() for _ in range(ns)] if ns > 0 else None
    torch.cuda.synchronize()
    def go(k):
        for i in range(k):
            if streams is None: step(i)
            else:
                with torch.cuda.stream(streams[i % ns]): step(i)
                if keep is not None and i == keep:
                    pass
    go(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(steps); torch.cuda.synchronize(); t = time.perf_counter() - t0
    return steps / t
def run_lock(ns):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    main = torch.cuda.current_stream()
    def go(k):
        for b0 in range(0, k, ns):
            for v in range(ns):
                streams[v].wait_stream(main)
                with torch.cuda.stream(streams[v]): step(b0 + v)
            for v in range(ns): main.wait_stream(streams[v])
    torch.cuda.synchronize(); go(2 * ns); torch.cuda.synchronize()
    k = (steps // ns) * ns
    t0 = time.perf_counter(); go(k); torch.cuda.synchronize(); t = time.perf_counter() - t0
    return k / t
for rep in range(2):
    for ns in (2, 3, 4):
        print("lockstep batches of", ns, "%.1f views/s" % run_lock(ns), flush=True)
for rep in range(2):
    for ns in (0, 1, 2, 3, 4):
        print("streams", ns, "%.1f views/s" % run(ns), flush=True)
# gradients of one view on a side stream with another view in flight vs serial
def grads(i):
    return [p.grad.clone() for p in params if p.grad is not None]
step(5); torch.cuda.synchronize(); ref = grads(5)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(s1): step(4)
with torch.cuda.stream(s2):
    step(5); got = grads(5)
torch.cuda.synchronize()
print("bit-identical under overlap:", all(torch.equal(a, b) for a, b in zip(ref, got)))
