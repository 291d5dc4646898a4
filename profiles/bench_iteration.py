#!/usr/bin/env python
"""Secondary measurement (not the headline metric): one GAUSSIAN-state training iteration of train.py:196-299 without
the optimizer step -- deformation MLP with gradients, render(), L1 + SSIM loss, loss.backward() -- at the S4 size
(300k Gaussians, 1080p, F = 32), one view per iteration:
  all_hip  : DeformNetworkHIP (fused bf16-MFMA MLP, training pair) + fused render() + fused L1/SSIM
  ref_comp : the reference's own composition around the HIP rasterizer operator: fp32 PyTorch MLP, PyTorch prep
             ops (activations / concat / normalise), PyTorch l1_loss + ssim (utils/loss_utils.py:30-86 restated)"""
import sys, os, time, json, math
import torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd import rasterizer as R
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe, SynthDeformNetwork
from trase_amd.deform import DeformNetworkHIP
from trase_amd.losses import photometric_loss
from trase_amd.optim import FusedAdam
from trase_amd.densify import add_densification_stats
from trase_amd.renderer import render
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def ref_ssim(x, y, win):
    conv = lambda t: Fn.conv2d(t, win, padding=5, groups=x.shape[0])
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()


def main():
    N, W, H, F = 300_000, 1920, 1080, 32
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
    net = SynthDeformNetwork().to(dev)
    with torch.no_grad():                       # small deformations, as after the warm-up of the reference
        for m in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling):
            m.weight.mul_(0.01); m.bias.zero_()
    hip_net = DeformNetworkHIP(net)
    params = pc.parameters() + list(net.parameters())
    cams = [orbit_camera(W, H, angle=2 * math.pi * k / 8, fid=k / 8).to(dev) for k in range(8)]
    bg = torch.zeros(3, device=dev)
    gts = [torch.rand(3, H, W, device=dev) for _ in range(2)]
    g = torch.tensor([math.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    win = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous().to(dev)
    pipe = SynthPipe()

    def zero():
        for p in params:
            p.grad = None

    # the rest of the iteration (train.py:361-389): densification statistics + the two optimizer steps
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3, 2.5e-3]
    # (learning rates 0: the update arithmetic is the same, but the scene -- fitted to random targets otherwise -- stays the
    # one that was sized; with the real rates the pair count drifts with the number of warm-up iterations)
    groups = lambda: [{"params": [p], "lr": 0.0 * lr, "name": str(k)} for k, (p, lr) in enumerate(zip(pc.parameters(), lrs))]
    opt_hip = [FusedAdam(groups(), lr=0.0, eps=1e-15), FusedAdam(list(net.parameters()), lr=0.0, eps=1e-15)]
    opt_ref = [torch.optim.Adam(groups(), lr=0.0, eps=1e-15), torch.optim.Adam(list(net.parameters()), lr=0.0, eps=1e-15)]
    from types import SimpleNamespace
    stats = SimpleNamespace(xyz_gradient_accum=torch.zeros(N, 1, device=dev), denom=torch.zeros(N, 1, device=dev),
                            max_radii2D=torch.zeros(N, device=dev))

    def all_hip(i):
        zero()
        cam = cams[i % 8]
        t = cam.fid.reshape(1, 1).expand(N, -1)            # train.py:186-196: a stride-0 view of the camera's device-resident fid
        d_xyz, d_rot, d_scale = hip_net(pc.get_xyz.detach(), t)
        out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
        photometric_loss(out["render"], gts[i % 2], 0.2).backward()
        return out["viewspace_points"], out["radii"]

    def all_hip_full(i):
        vp, radii = all_hip(i)
        add_densification_stats(stats, vp, radii)
        for o in opt_hip:
            o.step()

    def ref_comp(i):
        zero()
        cam = cams[i % 8]
        t = cam.fid.reshape(1, 1).expand(N, -1)            # train.py:186-196: a stride-0 view of the camera's device-resident fid
        d_xyz, d_rot, d_scale = net(pc.get_xyz.detach(), t.contiguous())
        st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                           tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
                                           viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                           sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
        m2d = torch.zeros_like(pc.get_xyz, requires_grad=True)
        gf = pc.get_gaussian_features
        sh_objs = gf / (gf.norm(dim=2, keepdim=True) + 1e-9)
        img, radii, feats, depth = GaussianRasterizer(raster_settings=st)(
            means3D=pc.get_xyz + d_xyz, means2D=m2d, shs=pc.get_features, sh_objs=sh_objs, colors_precomp=None,
            opacities=pc.get_opacity, scales=pc.get_scaling + d_scale, rotations=pc.get_rotation + d_rot, cov3D_precomp=None)
        gt = gts[i % 2]
        (0.8 * (img - gt).abs().mean() + 0.2 * (1.0 - ref_ssim(img, gt, win))).backward()
        return m2d, radii

    def ref_comp_full(i):
        vp, radii = ref_comp(i)
        vis = radii > 0                                                                   # train.py:362-365
        stats.max_radii2D[vis] = torch.max(stats.max_radii2D[vis], radii[vis])
        stats.xyz_gradient_accum[vis] += torch.norm(vp.grad[vis, :2], dim=-1, keepdim=True)
        stats.denom[vis] += 1
        for o in opt_ref:
            o.step()

    # capacity for sync-free steps
    R.set_sync(True)
    all_hip(0)
    cap = int(R.last_status()[2] * 1.3) + 1024
    R.set_sync(False, capacity=cap)

    def timed(fn, iters=16):
        for i in range(30):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    t_hip = timed(all_hip)
    from trase_amd.renderer import set_forward_scope
    set_forward_scope("image")           # opt-in: the GAUSSIAN state never reads the feature map (train.py:211)
    t_hip_img = timed(all_hip)
    set_forward_scope("all")
    t_ref = timed(ref_comp)
    t_hip_full = timed(all_hip_full)
    t_ref_full = timed(ref_comp_full)
    print(json.dumps({"workload": "GAUSSIAN-state iteration, 300k Gaussians, 1920x1080, F=32",
                      "all_hip_ms": round(t_hip, 3), "all_hip_image_scope_ms": round(t_hip_img, 3),
                      "ref_composition_around_hip_rasterizer_ms": round(t_ref, 3),
                      "iterations_per_s_all_hip": round(1e3 / t_hip, 1),
                      "with_stats_and_optimizer_steps": {"all_hip_ms": round(t_hip_full, 3), "ref_composition_ms": round(t_ref_full, 3),
                                                         "iterations_per_s_all_hip": round(1e3 / t_hip_full, 1)}}))


if __name__ == "__main__":
    main()
