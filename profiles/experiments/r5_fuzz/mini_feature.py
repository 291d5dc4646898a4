import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_renderer import render
from trase_amd import rasterizer as R
from trase_amd.deform import DeformNetworkHIP
from trase_amd.feature_head import contrastive_head, get_sample_pixel_and_mask, mask_stats
from trase_amd.optim import FusedAdam
from trase_amd.synthetic import SynthDeformNetwork, SynthGaussianModel, SynthPipe, make_scene, orbit_camera
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W, H, N = 480, 270, 40000
cams = [orbit_camera(W, H, angle=0.5 * k).to(dev) for k in range(4)]
for k, c in enumerate(cams):
    c.fid = torch.tensor([0.1 * (k + 1)], device=dev)
pc = SynthGaussianModel(make_scene(N, feat_dim=32, seed=2, scale_mult=0.6).to(dev))
for p in pc.parameters():
    p.requires_grad_(p is pc._gaussian_features)
pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(0)
# masks: vertical stripes (consistent segments across views is not the point; the head's machinery is)
sams = []
for c in cams:
    sam = torch.zeros(24, H, W, dtype=torch.bool, device=dev)
    for n in range(24):
        x0 = int(torch.randint(0, W - 60, (1,), generator=g)); y0 = int(torch.randint(0, H - 60, (1,), generator=g))
        sam[n, y0:y0 + int(torch.randint(30, 120, (1,), generator=g)), x0:x0 + int(torch.randint(30, 160, (1,), generator=g))] = True
    sams.append(sam)
hip_net = DeformNetworkHIP(SynthDeformNetwork().to(dev))
opt = FusedAdam([{"params": [pc._gaussian_features], "lr": 2.5e-3, "name": "gaussian_feats"}], eps=1e-15)
R.set_sync(True)
caps = []
with torch.no_grad():
    for c in cams:
        render(c, pc, pipe, bg, 0.0, 0.0, 0.0, norm_gaussian_features=True, is_smooth_gaussian_features=True, smooth_K=16); caps.append(R.last_status()[2])
R.set_sync(False, capacity=int(max(caps) * 2.0) + 4096)
R.set_graph("auto")
hist = []
for it in range(80):
    k = it % len(cams)
    cam, sam = cams[k], sams[k]
    with torch.no_grad():
        t = cam.fid.reshape(1, 1).expand(N, -1)
        d = hip_net(pc.get_xyz.detach(), t)
    d = [0.02 * x for x in d]
    out = render(cam, pc, pipe, bg, *d, norm_gaussian_features=True, is_smooth_gaussian_features=True, smooth_K=16)
    cover, size = mask_stats(sam)
    sp, sm = get_sample_pixel_and_mask(sam, 2000, 20, cover_count=cover, rng="cuda")
    lp, ln, ps, ns, reg = contrastive_head(out["render_gaussian_features"], sam, sp, sm, "soft", 0.75, 0.5, mask_size=size, with_norm_reg=True)
    loss = lp + ln + 1.0 * reg
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    if it % 10 == 0 or it == 79:
        R.check_overflow()
        hist.append((float(loss.detach()), float(ps), float(ns)))
        print(it, "loss %.4f pos_sim %.3f neg_sim %.3f graph hits %d" % (hist[-1] + (R.graph_stats()["hits"],)), flush=True)
torch.cuda.synchronize(); R.set_sync(True)
assert all(math.isfinite(h[0]) for h in hist)
print("done", hist[0], "->", hist[-1])
