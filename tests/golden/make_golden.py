"""Generates the golden fixtures in this directory from the *imported reference* (Python parts of
yunjinli/TRASE at /root/reference).  Runs only in the build container; the reference never
travels to the GPU box -- only the .npz files written here do.

    python tests/golden/make_golden.py

Fixtures:
  G1 sh_eval.npz        utils/sh_utils.py:57-112 eval_sh, + the colour rule gaussian_renderer/__init__.py:105-108
  G2 cov3d.npz          utils/general_utils.py:122-154 build_scaling_rotation + strip_symmetric (scene/gaussian_model.py:37-41)
  G3 camera.npz         utils/graphics_utils.py:45-77 getWorld2View2/getProjectionMatrix + scene/cameras.py:76-79
  G4 deform_mlp.npz     utils/time_utils.py:60-131 DeformNetwork forward/backward, fixed state_dict
     deform_mlp_blender.npz   the same for is_blender=True (t_multires 6 + timenet)
     deform_mlp_6dof.npz      forward of is_6dof=True (screw-axis heads + exp_se3)
  G5 losses.npz         utils/loss_utils.py:30-86 l1_loss, ssim
  G6 contrastive.npz    utils/loss_utils.py:275-406 pixel-pair losses, modes soft / all / hard
  G8 feature_head.npz   train.py:251-296 FEATURE-state head: utils/feature_utils.py:17-57 (sampler, C, C_F, weights) +
                        utils/loss_utils.py:275-406 pair losses (3 modes) + similarities + feature-norm regulariser
  G7 densify.npz        scene/gaussian_model.py:617-635 GaussianModel.densify_and_prune (clone, split, prune, both Adam
                        optimizers' state) on a CPU instance of the reference class, with and without max_screen_size
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    sys.path.insert(0, REF)
    for name in ("plyfile", "pytorch3d", "pytorch3d.ops", "simple_knn", "simple_knn._C",
                 "diff_gaussian_rasterization", "torchvision", "torchvision.models", "torchvision.utils",
                 "cv2", "imageio"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.modules["pytorch3d.ops"].knn_points = None
    sys.modules["pytorch3d"].ops = sys.modules["pytorch3d.ops"]
    sys.modules["torchvision.utils"].save_image = None
    sys.modules["pytorch3d.ops"].ball_query = None
    sys.modules["plyfile"].PlyData = None
    sys.modules["plyfile"].PlyElement = None
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizationSettings = None
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizer = None


class no_cuda_kwarg:
    """Drops device='cuda' from torch factory calls while reference code runs on the CPU."""

    def __enter__(self):
        self.saved = {}
        for fn in ("zeros", "ones", "empty", "tensor", "zeros_like", "rand"):
            orig = getattr(torch, fn)
            self.saved[fn] = orig

            def wrap(*a, __orig=orig, **k):
                if str(k.get("device", "")).startswith("cuda"):
                    k.pop("device")
                return __orig(*a, **k)
            setattr(torch, fn, wrap)
        return self

    def __exit__(self, *exc):
        for fn, orig in self.saved.items():
            setattr(torch, fn, orig)


def main():
    import_reference()
    torch.manual_seed(0)
    np.random.seed(0)
    from utils.sh_utils import eval_sh
    from utils.general_utils import build_scaling_rotation, strip_symmetric
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix

    # ---- G1
    n = 257
    shs = torch.randn(n, 16, 3) * 0.4            # (N,16,3) as stored by GaussianModel.get_features
    xyz = torch.randn(n, 3) * 2.0
    campos = torch.tensor([0.3, -0.2, 4.0])
    out = {}
    for deg in range(4):
        shs_view = shs.transpose(1, 2).view(-1, 3, 16)     # gaussian_renderer/__init__.py:104
        dir_pp = xyz - campos.repeat(n, 1)
        dir_n = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = eval_sh(deg, shs_view, dir_n)
        out[f"rgb_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()
        out[f"raw_deg{deg}"] = sh2rgb.numpy()
    np.savez_compressed(os.path.join(HERE, "sh_eval.npz"), shs=shs.numpy(), xyz=xyz.numpy(), campos=campos.numpy(), **out)

    # ---- G2
    s = torch.rand(n, 3) * 0.2 + 0.01
    q = torch.nn.functional.normalize(torch.randn(n, 4))
    with no_cuda_kwarg():
        L = build_scaling_rotation(1.3 * s, q)
        cov = strip_symmetric(L @ L.transpose(1, 2))
    np.savez_compressed(os.path.join(HERE, "cov3d.npz"), scales=s.numpy(), rotations=q.numpy(), modifier=np.float32(1.3),
                        cov6=cov.numpy())

    # ---- G3
    ang = 0.7
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]], dtype=np.float64)
    T = np.array([0.1, -0.3, 4.2])
    fovx, fovy = 0.9, 0.6
    w2v = getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)
    wvt = torch.tensor(w2v).transpose(0, 1)
    proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    np.savez_compressed(os.path.join(HERE, "camera.npz"), R=R, T=T, fovx=fovx, fovy=fovy, world_view_transform=wvt.numpy(),
                        projection_matrix=proj.numpy(), full_proj_transform=full.numpy(), camera_center=center.numpy())

    # ---- G4
    from utils.time_utils import DeformNetwork
    torch.manual_seed(1)
    net = DeformNetwork(D=8, W=256, multires=10, is_blender=False, is_6dof=False)
    m = 96
    x = (torch.rand(m, 3) * 2 - 1).requires_grad_(False)
    t = torch.full((m, 1), 0.37)
    d_xyz, d_rot, d_scale = net(x, t)
    gx, gr, gs = torch.randn_like(d_xyz), torch.randn_like(d_rot), torch.randn_like(d_scale)
    loss = (d_xyz * gx).sum() + (d_rot * gr).sum() + (d_scale * gs).sum()
    loss.backward()
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    grads = {"grad_" + k: p.grad.detach().numpy() for k, p in net.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "deform_mlp.npz"), x=x.numpy(), t=t.numpy(), d_xyz=d_xyz.detach().numpy(),
                        d_rotation=d_rot.detach().numpy(), d_scaling=d_scale.detach().numpy(), gx=gx.numpy(),
                        gr=gr.numpy(), gs=gs.numpy(), **{"w_" + k: v for k, v in sd.items()}, **grads)

    # ---- G4b: the is_blender variant (D-NeRF: t_multires = 6 + timenet, utils/time_utils.py:74-86), same cotangents
    rng_state = torch.get_rng_state()                      # G5 continues G4's random stream
    torch.manual_seed(11)
    netb = DeformNetwork(D=8, W=256, multires=10, is_blender=True, is_6dof=False)
    tb = torch.full((m, 1), 0.37)
    bx, br, bs = netb(x, tb)
    ((bx * gx).sum() + (br * gr).sum() + (bs * gs).sum()).backward()
    sdb = {k: v.detach().numpy() for k, v in netb.state_dict().items()}
    gradsb = {"grad_" + k: p.grad.detach().numpy() for k, p in netb.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "deform_mlp_blender.npz"), x=x.numpy(), t=tb.numpy(), d_xyz=bx.detach().numpy(),
                        d_rotation=br.detach().numpy(), d_scaling=bs.detach().numpy(), gx=gx.numpy(), gr=gr.numpy(),
                        gs=gs.numpy(), **{"w_" + k: v for k, v in sdb.items()}, **gradsb)
    # ---- G4c: is_6dof (screw-axis heads + exp_se3, utils/time_utils.py:100-118, utils/rigid_utils.py:43-86); outputs only
    torch.manual_seed(12)
    net6 = DeformNetwork(D=8, W=256, multires=10, is_blender=False, is_6dof=True)
    with torch.no_grad():
        sx, sr, ss = net6(x, t)
    np.savez_compressed(os.path.join(HERE, "deform_mlp_6dof.npz"), x=x.numpy(), t=t.numpy(), d_xyz=sx.numpy(),
                        d_rotation=sr.numpy(), d_scaling=ss.numpy(),
                        **{"w_" + k: v.detach().numpy() for k, v in net6.state_dict().items()})
    torch.set_rng_state(rng_state)

    # ---- G5
    from utils.loss_utils import l1_loss, ssim
    a = torch.rand(3, 48, 64)
    b = (a + 0.1 * torch.randn(3, 48, 64)).clamp(0, 1)
    a.requires_grad_(True)
    l1 = l1_loss(a, b)
    ss = ssim(a, b)
    total = (1.0 - 0.2) * l1 + 0.2 * (1.0 - ss)            # train.py:235-238
    total.backward()
    np.savez_compressed(os.path.join(HERE, "losses.npz"), a=a.detach().numpy(), b=b.numpy(), l1=l1.item(), ssim=ss.item(),
                        total=total.item(), grad_a=a.grad.numpy())
    # ---- G6: contrastive pixel-pair losses, 'soft' mode (arguments/__init__.py:131), utils/loss_utils.py:304-349
    from utils.loss_utils import positive_pixel_pair_loss, negative_pixel_pair_loss
    torch.manual_seed(6)
    S, nm = 192, 9
    memb = (torch.rand(nm, S) < 0.25).float()              # pixel-mask correspondence vectors (utils/feature_utils.py:46-55)
    C = torch.einsum("nh,nj->hj", memb, memb)
    C[C != 0] = 1
    f = torch.nn.functional.normalize(torch.randn(S, 32) + 1.5 * memb.t() @ torch.randn(nm, 32), dim=-1)
    f.requires_grad_(True)
    CF = torch.einsum("hc,jc->hj", f, f)
    CF.retain_grad()
    wts = 1.0 + 9.0 * torch.rand(S, S)
    lp = positive_pixel_pair_loss["soft"](C=C, C_F=CF, positive_th=0.75, weights=wts)
    ln = negative_pixel_pair_loss["soft"](C=C, C_F=CF, negative_th=0.5, weights=wts)
    (lp + ln).backward()
    extra = {}
    for mode in ("all", "hard"):                           # utils/loss_utils.py:275-302 and :351-394
        CFm = CF.detach().clone().requires_grad_(True)
        mp = positive_pixel_pair_loss[mode](C=C, C_F=CFm, positive_th=0.75, weights=wts)
        mn = negative_pixel_pair_loss[mode](C=C, C_F=CFm, negative_th=0.5, weights=wts)
        (mp + mn).backward()
        extra.update({f"loss_pos_{mode}": float(mp), f"loss_neg_{mode}": float(mn), f"grad_CF_{mode}": CFm.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "contrastive.npz"), C=C.numpy(), CF=CF.detach().numpy(), weights=wts.numpy(),
                        loss_pos=float(lp), loss_neg=float(ln), grad_CF=CF.grad.numpy(), **extra)
    # ---- G7: densify_and_prune of the reference's own GaussianModel (scene/gaussian_model.py:617-635), run on the CPU.
    # The split samples are torch.normal(mean=0, std=stds) = standard normals * stds (checked below): the standard
    # normals are recorded as an input, so the fixture pins everything except the generator itself.
    from types import SimpleNamespace
    from scene.gaussian_model import GaussianModel
    targs = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                            position_lr_delay_mult=0.01, position_lr_max_steps=30_000, feature_lr=0.0025, opacity_lr=0.05,
                            scaling_lr=0.005, rotation_lr=0.001)                        # arguments/__init__.py:100-113
    extent = 3.7
    g7 = {"extent": extent, "percent_dense": targs.percent_dense, "max_grad": 0.0002, "min_opacity": 0.005}
    for tag, size_threshold in (("a", 20), ("b", None)):                                # train.py:369
        torch.manual_seed(70 + (tag == "b"))
        n7 = 1201
        with no_cuda_kwarg():
            gm = GaussianModel(3)
            gm.spatial_lr_scale = 1.0
            gm._xyz = torch.nn.Parameter(torch.randn(n7, 3) * 1.5)
            gm._features_dc = torch.nn.Parameter(torch.randn(n7, 1, 3))
            gm._features_rest = torch.nn.Parameter(torch.randn(n7, 15, 3) * 0.1)
            # activated scales around percent_dense * extent = 0.037, a tail above 0.1 * extent = 0.37
            gm._scaling = torch.nn.Parameter(torch.log(torch.tensor(0.037)) + torch.randn(n7, 3) * 1.2)
            gm._rotation = torch.nn.Parameter(torch.randn(n7, 4))
            gm._opacity = torch.nn.Parameter(torch.randn(n7, 1) * 3.0 - 2.0)           # some below sigmoid^-1(0.005) = -5.3
            gm._gaussian_features = torch.nn.Parameter(torch.randn(n7, 1, 32))
            gm.max_radii2D = torch.rand(n7) * 40.0
            gm.training_setup(targs)
            for _ in range(2):                                                          # non-trivial Adam moments
                for mode in ("GAUSSIAN", "FEATURE"):
                    for grp in gm.optimizer[mode].param_groups:
                        grp["params"][0].grad = torch.randn_like(grp["params"][0]) * 0.01
                    gm.optimizer[mode].step()
            gm.denom = torch.randint(0, 4, (n7, 1)).float()                             # zeros -> nan -> 0 path
            gm.xyz_gradient_accum = torch.rand(n7, 1) * 0.0006 * gm.denom
            inp = {}
            for mode in ("GAUSSIAN", "FEATURE"):
                for grp in gm.optimizer[mode].param_groups:
                    prm = grp["params"][0]
                    st = gm.optimizer[mode].state[prm]
                    inp[f"in_{grp['name']}"] = prm.detach().clone().numpy()
                    inp[f"in_{grp['name']}_m"] = st["exp_avg"].clone().numpy()
                    inp[f"in_{grp['name']}_v"] = st["exp_avg_sq"].clone().numpy()
            inp["in_accum"], inp["in_denom"] = gm.xyz_gradient_accum.clone().numpy(), gm.denom.clone().numpy()
            inp["in_max_radii2D"] = gm.max_radii2D.clone().numpy()
            rec = {}
            real_normal = torch.normal

            def spy_normal(*a, **k):
                state = torch.get_rng_state()
                out_ = real_normal(*a, **k)
                after = torch.get_rng_state()
                torch.set_rng_state(state)
                zz = torch.randn(out_.shape)
                torch.set_rng_state(after)
                assert torch.equal(zz * k["std"] + k["mean"], out_), "torch.normal(mean, std) != randn * std + mean"
                rec["z"] = zz
                return out_
            torch.normal = spy_normal
            real_empty_cache = torch.cuda.empty_cache
            torch.cuda.empty_cache = lambda: None
            try:
                torch.manual_seed(700)
                num_clone, num_split = gm.densify_and_prune(g7["max_grad"], g7["min_opacity"], extent, size_threshold)
            finally:
                torch.normal = real_normal
                torch.cuda.empty_cache = real_empty_cache
            out7 = {"z": rec["z"].numpy(), "num_clone": int(num_clone), "num_split": int(num_split)}
            for mode in ("GAUSSIAN", "FEATURE"):
                for grp in gm.optimizer[mode].param_groups:
                    prm = grp["params"][0]
                    st = gm.optimizer[mode].state[prm]
                    out7[f"out_{grp['name']}"] = prm.detach().numpy()
                    out7[f"out_{grp['name']}_m"] = st["exp_avg"].numpy()
                    out7[f"out_{grp['name']}_v"] = st["exp_avg_sq"].numpy()
            assert gm._xyz is gm.optimizer["GAUSSIAN"].param_groups[0]["params"][0]
            assert float(gm.xyz_gradient_accum.abs().sum() + gm.denom.abs().sum() + gm.max_radii2D.abs().sum()) == 0.0
            print("G7", tag, "rows", n7, "->", gm._xyz.shape[0], "clone", int(num_clone), "split", int(num_split))
        g7.update({f"{tag}_{k}": v for k, v in {**inp, **out7}.items()})
    np.savez_compressed(os.path.join(HERE, "densify.npz"), **g7)
    # ---- G8: the FEATURE-state head as train.py:251-296 composes it from the reference's helpers
    from utils.feature_utils import (get_sample_pixel_and_mask, get_pixel_weights, get_pixel_mask_correspondence_matrix,
                                     get_features_correspondence_matrix)
    torch.manual_seed(8)
    n8, h8, w8 = 14, 40, 56
    yy, xx = torch.meshgrid(torch.arange(h8), torch.arange(w8), indexing="ij")
    sam = torch.zeros(n8, h8, w8, dtype=torch.bool)
    for k in range(n8):                                     # overlapping boxes and discs, some pixels uncovered
        cy, cx = int(torch.randint(0, h8, (1,))), int(torch.randint(0, w8, (1,)))
        ry, rx = int(torch.randint(3, 14, (1,))), int(torch.randint(3, 18, (1,)))
        if k % 2:
            sam[k] = ((yy - cy).abs() <= ry) & ((xx - cx).abs() <= rx)
        else:
            sam[k] = ((yy - cy).float() / ry) ** 2 + ((xx - cx).float() / rx) ** 2 <= 1.0
    base = torch.randn(n8, 32)
    feat = (sam.float().permute(1, 2, 0) @ base).permute(2, 0, 1) * 0.7 + 0.6 * torch.randn(32, h8, w8)
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self         # the sampler moves its CPU draws to the GPU; stay on the CPU
    try:
        torch.manual_seed(80)
        sampled_pixel, sampled_mask = get_sample_pixel_and_mask(sam, 300, 7)
    finally:
        torch.Tensor.cuda = saved_cuda
    g8 = {"sam_masks": sam.numpy(), "features": feat.numpy(), "sampled_pixel": sampled_pixel.numpy(),
          "sampled_mask": sampled_mask.numpy(), "sampler_seed": 80, "num_sampled_pixels": 300, "num_sampled_masks": 7,
          "positive_th": 0.75, "negative_th": 0.5}
    Cm = get_pixel_mask_correspondence_matrix(sam, sampled_pixel, sampled_mask)
    Wm = get_pixel_weights(sam, sampled_pixel)
    g8["C"], g8["weights"] = Cm.numpy(), Wm.numpy()
    for mode, use_w in (("soft", True), ("all", True), ("hard", True), ("soft", False)):
        fr = feat.clone().requires_grad_(True)
        CFm = get_features_correspondence_matrix(fr, sampled_pixel)
        lp = positive_pixel_pair_loss[mode](C=Cm, C_F=CFm, positive_th=0.75, weights=Wm if use_w else None)
        ln = negative_pixel_pair_loss[mode](C=Cm, C_F=CFm, negative_th=0.5, weights=Wm if use_w else None)
        (lp + ln).backward()
        tag = mode + ("" if use_w else "_noweights")
        g8[f"{tag}_loss_pos"], g8[f"{tag}_loss_neg"], g8[f"{tag}_grad"] = float(lp), float(ln), fr.grad.numpy()
    with torch.no_grad():
        g8["C_F"] = CFm.detach().numpy()
        g8["pos_similarity"] = float(CFm[Cm == 1].mean())   # train.py:295-296
        g8["neg_similarity"] = float(CFm[Cm == 0].mean())
    fr = feat.clone().requires_grad_(True)
    reg = (1 - fr.norm(dim=0, p=2).mean()) ** 2             # train.py:281-282
    reg.backward()
    g8["reg"], g8["reg_grad"] = float(reg), fr.grad.numpy()
    print("G8 S =", int(sampled_pixel.sum()), "sampled masks =", int(sampled_mask.sum()),
          {k: round(v, 5) for k, v in g8.items() if isinstance(v, float)})
    np.savez_compressed(os.path.join(HERE, "feature_head.npz"), **g8)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
