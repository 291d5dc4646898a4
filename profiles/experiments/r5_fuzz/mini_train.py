"""A miniature GAUSSIAN-state training loop over the whole stack with everything that changes state over time: deformation MLP (Morton
row order cache), fused render() under the sync-free policy with launch-graph replay and the guarded optimizer, photometric loss,
densification statistics, densify / prune every few iterations (the Gaussian count changes: workspaces, graph records, row-order cache,
optimizer state), opacity reset.  Checks: nothing raises, the loss stays finite and goes down, P follows densification."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_renderer import render
from trase_amd import rasterizer as R
from trase_amd.deform import DeformNetworkHIP
from trase_amd.densify import add_densification_stats, densify_and_prune
from trase_amd.losses import photometric_loss
from trase_amd.optim import FusedAdam
from trase_amd.synthetic import SynthDeformNetwork, SynthGaussianModel, SynthPipe, make_scene, orbit_camera

dev = torch.device("cuda", 0)
torch.manual_seed(0)
W, H = 320, 192
N0 = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 80
cams = [orbit_camera(W, H, angle=0.5 * k).to(dev) for k in range(6)]
for k, c in enumerate(cams):
    c.fid = torch.tensor([0.1 * (k + 1)], device=dev)
pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
# ground truth: another scene's renders
gt_pc = SynthGaussianModel(make_scene(N0, feat_dim=32, seed=1, scale_mult=0.9).to(dev), requires_grad=False)
with torch.no_grad():
    gts = [render(c, gt_pc, pipe, bg, 0.0, 0.0, 0.0)["render"].clone() for c in cams]
pc = SynthGaussianModel(make_scene(N0, feat_dim=32, seed=2, scale_mult=0.9).to(dev))
NAMES = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation",
         "gaussian_feats": "_gaussian_features"}
for a in NAMES.values():
    setattr(pc, a, torch.nn.Parameter(getattr(pc, a).detach().clone()))
lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3, "gaussian_feats": 2.5e-3}
grp = lambda names: [{"params": [getattr(pc, NAMES[n])], "lr": lrs[n], "name": n} for n in names]
pc.optimizer = {"GAUSSIAN": FusedAdam(grp(list(NAMES)[:6]), lr=0.0, eps=1e-15), "FEATURE": FusedAdam(grp(list(NAMES)[6:]), lr=0.0, eps=1e-15)}
pc.percent_dense, pc.feature_smooth_map, pc.mode = 0.01, None, "from_scratch"
def reset_stats():
    P = pc._xyz.shape[0]
    pc.xyz_gradient_accum = torch.zeros(P, 1, device=dev); pc.denom = torch.zeros(P, 1, device=dev); pc.max_radii2D = torch.zeros(P, device=dev)
reset_stats()
net = SynthDeformNetwork().to(dev)
with torch.no_grad():
    for m in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling):
        m.weight.mul_(0.01); m.bias.zero_()
hip_net = DeformNetworkHIP(net)
opt_net = FusedAdam(list(net.parameters()), lr=8e-4, eps=1e-15)

def size_capacity():
    R.set_sync(True)
    caps = []
    with torch.no_grad():
        for c in cams:
            render(c, pc, pipe, bg, 0.0, 0.0, 0.0); caps.append(R.last_status()[2])
    R.set_sync(False, capacity=int(max(caps) * 1.6) + 4096)

size_capacity()
R.set_graph("auto")
losses, Ps = [], []
skipped = 0
for it in range(iters):
    cam = cams[it % len(cams)]
    P = pc._xyz.shape[0]
    t = cam.fid.reshape(1, 1).expand(P, -1)
    d_xyz, d_rot, d_scale = hip_net(pc.get_xyz.detach(), t) if it >= 10 else (0.0, 0.0, 0.0)      # warm-up without deformation (train.py:189)
    try:
        out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
    except R.IterationSkipped as e:
        skipped += 1
        print(f"   it {it}: {str(e)[:90]} ...", flush=True)
        out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
    loss = photometric_loss(out["render"], gts[it % len(cams)], 0.2)
    loss.backward()
    add_densification_stats(pc, out["viewspace_points"], out["radii"])
    if it > 0 and it % 20 == 0:
        extent = 5.0
        nc, ns = densify_and_prune(pc, 0.00002, 0.005, extent, 20 if it > 40 else None)
        reset_stats()
        size_capacity()
        print(f"   it {it}: densify clone {nc} split {ns} -> P {pc._xyz.shape[0]}", flush=True)
    for o in pc.optimizer.values():
        o.step()
        o.zero_grad(set_to_none=True)
    if it >= 10:
        opt_net.step(); opt_net.zero_grad(set_to_none=True)
    if it % 10 == 0 or it == iters - 1:
        try:
            R.check_overflow()
        except R.IterationSkipped as e:
            skipped += 1
            print(f"   it {it}: (check) {str(e)[:90]} ...", flush=True)
        losses.append(float(loss)); Ps.append(pc._xyz.shape[0])
        print(f"it {it} loss {losses[-1]:.5f} P {Ps[-1]} graph {R.graph_stats()['hits']}", flush=True)
try:
    R.check_overflow()
except R.IterationSkipped:
    skipped += 1
torch.cuda.synchronize()
R.set_sync(True)
assert all(math.isfinite(x) for x in losses), losses
assert losses[-1] < losses[0], (losses[0], losses[-1])
print("skipped iterations reported:", skipped)
print("done", losses[0], "->", losses[-1], "P", Ps[0], "->", Ps[-1])
