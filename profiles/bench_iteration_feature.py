#!/usr/bin/env python
"""Secondary measurement: one FEATURE-state training iteration of train.py:189-299 (after the warm-up) without the
optimizer step, at the S4 size (300k Gaussians, 1080p, F = 32, 100 SAM masks at image resolution, 5000 sampled pixels,
50 sampled masks, smooth_K = 16, contrastive_mode = 'soft'):
  deformation MLP under no_grad -> render(norm_gaussian_features, smoothed features) -> regulariser + sampling +
  C / C_F / weights + pair losses + similarities -> loss.backward()
  all_hip  : DeformNetworkHIP forward, fused render() with fused KNN smoothing, trase_amd.feature_head (no S x S matrix)
  ref_comp : the reference's own composition around the HIP rasterizer operator: fp32 PyTorch MLP, PyTorch smoothing
             gather (scene/gaussian_model.py:79-104 restated; the KNN indices are cached as in the reference), PyTorch
             prep ops, and the helpers of utils/feature_utils.py + the 'soft' pair losses of utils/loss_utils.py restated"""
import sys, os, time, json, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd import rasterizer as R
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe, SynthDeformNetwork
from trase_amd.deform import DeformNetworkHIP
from trase_amd.renderer import render
from trase_amd.feature_head import contrastive_head, get_sample_pixel_and_mask, mask_stats
import pytorch3d.ops
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def ref_soft(C, C_F, pth, nth, w):                                     # utils/loss_utils.py:304-349
    diag = torch.eye(C_F.shape[0], dtype=bool, device=C_F.device)
    out = []
    for neg in (False, True):
        cond = torch.logical_and(C_F > nth, C == 0) if neg else torch.logical_and(C_F < pth, C == 1)
        m = torch.triu(torch.logical_and(torch.any(cond, dim=0), ~diag), diagonal=0)
        npair = torch.nonzero(m).shape[0]
        m = torch.logical_and(m, C == (0 if neg else 1)).bool()
        if m.sum() == 0:
            out.append(0.0)
        elif neg:
            out.append((w[m] * torch.relu(C_F[m])).sum() / npair)
        else:
            out.append((-w[m] * C_F[m]).sum() / npair)
    return out


def ref_head(feats, sam_masks, nsp, nsm):                              # train.py:251-296 + utils/feature_utils.py:17-57
    sampled_mask = torch.rand(sam_masks.shape[0]).cuda() < nsm / sam_masks.shape[0]
    sampled_pixel = torch.rand(sam_masks.shape[-2], sam_masks.shape[-1]).cuda() < nsp / (sam_masks.shape[-1] * sam_masks.shape[-2])
    sampled_pixel = torch.logical_and(sampled_pixel, ~(sam_masks.sum(dim=0) == 0))
    v = sam_masks[:, sampled_pixel][sampled_mask, :]
    C = torch.einsum("nh,nj->hj", v.float(), v.float())
    C[C != 0] = 1
    reg = (1 - feats.norm(dim=0, p=2).mean()) ** 2
    feats = torch.nn.functional.interpolate(feats.unsqueeze(0), sam_masks.shape[-2:], mode="bilinear").squeeze(0)
    f = torch.nn.functional.normalize(feats[:, sampled_pixel].permute([1, 0]), dim=-1, p=2)
    C_F = torch.einsum("hc,jc->hj", f, f)
    size = sam_masks * sam_masks.sum(-1).sum(-1)[:, None, None]
    m = (size.sum(dim=0) / (sam_masks.sum(dim=0) + 1e-9))[sampled_pixel]
    pp = m.unsqueeze(0) * m.unsqueeze(1)
    mx = pp.max()
    pp[pp == 0] = 1e10
    w = torch.clamp(mx / pp, 1.0, None)
    w = (w - w.min()) / (w.max() - w.min()) * 9. + 1.
    lp, ln = ref_soft(C, C_F, 0.75, 0.5, w)
    with torch.no_grad():
        C_F[C == 1].mean(); C_F[C == 0].mean()
    return lp + ln + 1.0 * reg


def main():
    N, W, H, F, NM = 300_000, 1920, 1080, 32, 100
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
    for p in pc.parameters():                       # change_optimization_target('FEATURE'), scene/gaussian_model.py:303-315
        p.requires_grad_(p is pc._gaussian_features)
    net = SynthDeformNetwork().to(dev)
    with torch.no_grad():
        for m in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling):
            m.weight.mul_(0.01); m.bias.zero_()
    hip_net = DeformNetworkHIP(net)
    cams = [orbit_camera(W, H, angle=2 * math.pi * k / 8, fid=k / 8).to(dev) for k in range(8)]
    bg = torch.zeros(3, device=dev)
    g = torch.Generator().manual_seed(0)
    sam = torch.zeros(NM, H, W, dtype=torch.bool, device=dev)
    for n in range(NM):
        y0, x0 = int(torch.randint(0, H - 50, (1,), generator=g)), int(torch.randint(0, W - 50, (1,), generator=g))
        h, w = int(torch.randint(40, 500, (1,), generator=g)), int(torch.randint(40, 700, (1,), generator=g))
        sam[n, y0:y0 + h, x0:x0 + w] = True
    pipe = SynthPipe()
    knn_idx = pytorch3d.ops.knn_points(pc.get_xyz.detach().unsqueeze(0), pc.get_xyz.detach().unsqueeze(0), K=16).idx.squeeze()   # cached until densification

    def all_hip(i, rng):
        pc._gaussian_features.grad = None
        cam = cams[i % 8]
        with torch.no_grad():
            t = cam.fid.reshape(1, 1).expand(N, -1)            # train.py:186-196: a stride-0 view of the camera's device-resident fid
            d_xyz, d_rot, d_scale = hip_net(pc.get_xyz.detach(), t)
        out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale, norm_gaussian_features=True, is_smooth_gaussian_features=True, smooth_K=16)
        cover, size = mask_stats(sam)
        sp, sm = get_sample_pixel_and_mask(sam, 5000, 50, cover_count=cover, rng=rng)
        lp, ln, ps, ns, reg = contrastive_head(out["render_gaussian_features"], sam, sp, sm, "soft", 0.75, 0.5, mask_size=size,
                                               with_norm_reg=True)
        (lp + ln + 1.0 * reg).backward()

    def ref_comp(i):
        pc._gaussian_features.grad = None
        cam = cams[i % 8]
        with torch.no_grad():
            t = cam.fid.reshape(1, 1).expand(N, -1)            # train.py:186-196: a stride-0 view of the camera's device-resident fid
            d_xyz, d_rot, d_scale = net(pc.get_xyz.detach(), t.contiguous())
        st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                           tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
                                           viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                           sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
        m2d = torch.zeros_like(pc.get_xyz, requires_grad=True)
        normed = torch.nn.functional.normalize(pc.get_gaussian_features, dim=-1, p=2)   # scene/gaussian_model.py:95-101
        select_idx = knn_idx[:, torch.randperm(16)[:8]]
        sm = normed[select_idx, 0, :].mean(dim=1).unsqueeze(1)
        sh_objs = sm / (sm.norm(dim=2, keepdim=True) + 1e-9)                            # gaussian_renderer/__init__.py:119-121
        img, radii, feats, depth = GaussianRasterizer(raster_settings=st)(
            means3D=pc.get_xyz + d_xyz, means2D=m2d, shs=pc.get_features, sh_objs=sh_objs, colors_precomp=None,
            opacities=pc.get_opacity, scales=pc.get_scaling + d_scale, rotations=pc.get_rotation + d_rot, cov3D_precomp=None)
        ref_head(feats, sam, 5000, 50).backward()

    R.set_sync(True)
    all_hip(0, "cuda")
    cap = int(R.last_status()[2] * 1.3) + 1024
    R.set_sync(False, capacity=cap)

    def timed(fn, iters=12):
        for i in range(30):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    from trase_amd.renderer import set_backward_scope
    t_hip = timed(lambda i: all_hip(i, "cuda"))
    set_backward_scope("features")            # after densify_until_iter: only the features need a gradient
    t_hip_scope = timed(lambda i: all_hip(i, "cuda"))
    set_backward_scope("all")
    t_hip_cpu = timed(lambda i: all_hip(i, "cpu"))
    t_ref = timed(ref_comp)
    print(json.dumps({"workload": "FEATURE-state iteration without optimizer step, 300k Gaussians, 1920x1080, F=32, 100 masks, "
                                  "5000 sampled pixels, smooth_K=16, contrastive 'soft'",
                      "all_hip_ms": round(t_hip, 3), "all_hip_feature_only_backward_ms": round(t_hip_scope, 3),
                      "all_hip_reference_cpu_sampling_ms": round(t_hip_cpu, 3),
                      "ref_composition_around_hip_rasterizer_ms": round(t_ref, 3),
                      "iterations_per_s_all_hip": round(1e3 / t_hip, 1)}))


if __name__ == "__main__":
    main()
