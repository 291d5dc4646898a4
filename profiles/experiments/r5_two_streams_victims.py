"""Round-5 experiment: are STOCK torch kernels victims of the two-stream corruption that
profiles/r4_two_streams.md found in this library's own preprocess kernel?

Aggressors (stream A, looping): a forward of this library (`render_fwd_mf`: transposing LDS reads feeding MFMAs), a
forward+backward (`render_bwd_hw` too), the deformation MLP's inference kernel (`mlp_fwd_kernel_blk`).
Victims (stream B, known answers = the same call made with nothing else on the GPU, compared BITWISE):
  poly3    a degree-3 polynomial over 10^7 floats (elementwise VALU, literals in the FMAs)
  cumsum   torch.cumsum over 2^24 floats
  matmul   bf16 2048^3 matmul (hipBLASLt MFMA kernel)
  allreduce  one-rank RCCL all_reduce of 2^22 floats followed by *2 (checks the payload)
  masked   poly3 under a data-dependent mask through torch.where on a strided view (partial results kept)
  preprocess  POSITIVE CONTROL: this library's preprocess colours vs the torch SH evaluation (the known victim)
Counts wrong words over REPS repetitions per (aggressor, victim) pair.  TRASE_UNORDERED_STREAMS=1 must be set (the
library otherwise orders its own launches across streams).
  TRASE_UNORDERED_STREAMS=1 python profiles/experiments/r5_two_streams_victims.py [reps]"""
import sys, os, math, json, time
os.environ.setdefault("TRASE_UNORDERED_STREAMS", "1")
import torch
sys.path.insert(0, os.getcwd())
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe, SynthDeformNetwork
from trase_amd import rasterizer as R
from trase_amd.sh import sh_colors_python
from trase_amd.deform import deform_forward
from gaussian_renderer import render

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda")
torch.manual_seed(0)
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
pipe = SynthPipe()
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev) for k in range(16)]
bg = torch.zeros(3, device=dev)
net = SynthDeformNetwork().to(dev)
mlp_params = dict(net.state_dict())
xs = (torch.rand(N, 3, device=dev) * 2 - 1) * 1.3
ts = torch.tensor([[0.4]], device=dev).expand(N, -1)


def fwd(i):
    with torch.no_grad():
        return render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)


def fwd_bwd(i):
    for p in pc.parameters():
        p.grad = None
    o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)
    torch.autograd.backward([o["render"], o["render_gaussian_features"]], [gi, gf])


R.set_sync(True)
caps = []
for i in range(16):
    fwd(i); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
gi = torch.randn(3, H, W, device=dev); gf = torch.randn(F, H, W, device=dev)

try:
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
    have_pg = True
except Exception as e:                                    # noqa: BLE001
    print("no process group:", e, flush=True)
    have_pg = False

X = torch.randn(10_000_000, device=dev)
Xc = torch.randn(1 << 24, device=dev)
Am = torch.randn(2048, 2048, device=dev).bfloat16(); Bm = torch.randn(2048, 2048, device=dev).bfloat16()
Ar = torch.randn(1 << 22, device=dev)
Xs = torch.randn(5_000_000, 2, device=dev)


def v_poly3():
    return ((0.37 * X + 1.19) * X - 0.73) * X + 0.11


def v_cumsum():
    return torch.cumsum(Xc, 0)


def v_matmul():
    return Am @ Bm


def v_allreduce():
    t = Ar.clone()
    dist.all_reduce(t)
    return t * 2.0


def v_masked():
    x = Xs[:, 0]
    return torch.where(x > 0.25, ((0.37 * x + 1.19) * x - 0.73) * x + 0.11, x)


def v_preprocess():
    # this library's own preprocess (the known victim): the colours it wrote for view 5, against torch
    o = fwd(5)
    rgb = R.last_geom_view(N)["rgb_depth"][:, :3]
    return torch.where((o["radii"] > 0)[:, None], rgb, torch.zeros_like(rgb)).contiguous()   # culled rows are undefined


victims = {"poly3": v_poly3, "cumsum": v_cumsum, "matmul_bf16": v_matmul, "masked_poly": v_masked}
if have_pg:
    victims["allreduce_1rank"] = v_allreduce
victims["preprocess(control)"] = v_preprocess
aggressors = {
    "none": lambda i: None,
    "forward(render_fwd_mf)": fwd,
    "fwd+bwd(render_bwd_hw)": fwd_bwd,
    "mlp_fwd_kernel_blk": lambda i: deform_forward(mlp_params, xs, ts),
}

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
want, tol = {}, {}
for k, f in victims.items():
    a = f(); torch.cuda.synchronize()
    dev_max = 0.0
    for _ in range(6):
        b = f(); torch.cuda.synchronize()
        dev_max = max(dev_max, float((a.float() - b.float()).abs().max()))
    want[k] = a
    # bitwise where the kernel is reproducible serially; otherwise (torch.cumsum: look-back order varies) a word counts as
    # wrong when it is further from the first serial result than 4 x the largest serial-to-serial deviation seen
    tol[k] = 4.0 * dev_max
    print(f"victim {k}: serial run-to-run max deviation {dev_max:.3e} -> {'bitwise' if dev_max == 0 else 'tolerance %.3e' % tol[k]}", flush=True)
if "preprocess(control)" in want:
    with torch.no_grad():
        ref = sh_colors_python(pc, cams[5].camera_center).float()
        vis = (want["preprocess(control)"].abs().sum(1) > 0)[:, None]
        print("control: serial preprocess colours vs torch SH (visible rows), max abs",
              float(((want["preprocess(control)"] - ref) * vis).abs().max()), flush=True)

table = {}
t_start = time.time()
for an, ag in aggressors.items():
    for vn, vf in victims.items():
        reps = REPS if vn != "preprocess(control)" else min(REPS, 48)
        wrong_runs, wrong_words = 0, 0
        for i in range(reps):
            with torch.cuda.stream(sa):
                for _ in range(2):
                    ag(i)
            with torch.cuda.stream(sb):
                out = vf()
            with torch.cuda.stream(sa):
                ag(i)
            sb.synchronize()
            d = (out != want[vn]) if tol[vn] == 0.0 else ((out.float() - want[vn].float()).abs() > tol[vn])
            n = int(d.sum())
            wrong_words += n; wrong_runs += int(n > 0)
        torch.cuda.synchronize()
        table[f"{an} | {vn}"] = {"reps": reps, "runs_with_wrong_words": wrong_runs, "wrong_words": wrong_words}
        print(f"aggressor {an:28s} victim {vn:22s} runs wrong {wrong_runs:4d} / {reps}  words {wrong_words}", flush=True)
        if time.time() - t_start > float(os.environ.get("BUDGET_S", "1500")):
            print("time budget reached", flush=True)
            break
print(json.dumps(table))
