"""Self-checks of the float64 oracle: semantics of the blend gates, autograd vs finite
differences, and the S1 plumbing case (1k Gaussians, 128x128, RGB only)."""
import numpy as np
import torch

from oracle import raster_oracle as ro
from tests.util import settings_for, small_case


def _render(act, st, **kw):
    return ro.rasterize(st, act["means3D"], None, shs=act["shs"], sh_objs=act.get("sh_objs"),
                        opacities=act["opacities"], scales=act["scales"], rotations=act["rotations"], **kw)


def test_s1_plumbing_case_runs_and_is_sane():
    act, cam = small_case(n=1000, w=128, h=128, feat=0, seed=0, scale_mult=0.6)
    st = settings_for(cam, bg=(0.2, 0.3, 0.4))
    out = _render(act, st)
    assert out.image.shape == (3, 128, 128) and out.depth.shape == (1, 128, 128) and out.feats.shape == (0, 128, 128)
    assert out.num_rendered > 1000
    assert torch.isfinite(out.image).all()
    # transmittance bookkeeping: image = C + T*bg  =>  where nothing was blended the pixel shows bg
    empty = out.n_contrib == 0
    if empty.any():
        np.testing.assert_allclose(out.image[:, empty].numpy(), np.tile(np.array([[0.2], [0.3], [0.4]]), (1, int(empty.sum()))))
    assert (out.final_T <= 1).all() and (out.final_T >= 1e-4 * (1 - 0.99) - 1e-12).all()


def test_single_gaussian_closed_form():
    """One isotropic Gaussian in front of the camera: alpha = min(0.99, o*exp(-r^2/(2 s^2)))."""
    from trase_amd.synthetic import orbit_camera
    cam = orbit_camera(64, 64, angle=0.0, elevation=0.0)
    st = settings_for(cam, sh_degree=0, bg=(0, 0, 0))
    sigma_w = 0.05
    means = torch.zeros(1, 3)
    shs = torch.zeros(1, 16, 3); shs[:, 0] = (torch.tensor([0.8, 0.5, 0.2]) - 0.5) / ro.SH_C0
    out = ro.rasterize(st, means, None, shs=shs, sh_objs=torch.ones(1, 1, 4), opacities=torch.full((1, 1), 0.7),
                       scales=torch.full((1, 3), sigma_w), rotations=torch.tensor([[1.0, 0, 0, 0]]))
    focal = 1.2 * 64
    var = (sigma_w * focal / 4.0) ** 2 + 0.3
    y, x = 31, 33
    r2 = (x - 31.5) ** 2 + (y - 31.5) ** 2
    alpha = min(0.99, 0.7 * np.exp(-0.5 * r2 / var))
    np.testing.assert_allclose(out.image[:, y, x].numpy(), alpha * np.array([0.8, 0.5, 0.2]), rtol=1e-6)
    np.testing.assert_allclose(out.feats[:, y, x].numpy(), alpha * np.ones(4), rtol=1e-6)
    np.testing.assert_allclose(out.depth[0, y, x].item(), alpha * 4.0, rtol=1e-6)
    assert int(out.radii[0]) == int(np.ceil(3 * np.sqrt(var)))


def test_autograd_matches_finite_differences():
    act, cam = small_case(n=40, w=48, h=32, feat=4, seed=3, scale_mult=2.5)
    st = settings_for(cam, bg=(0.1, 0.2, 0.3))
    gen = torch.Generator().manual_seed(0)
    gi = torch.randn(3, 32, 48, generator=gen, dtype=torch.float64)
    gf = torch.randn(4, 32, 48, generator=gen, dtype=torch.float64)
    leaves = {k: act[k].double().clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "shs", "sh_objs")}

    def loss_of(lv):
        o = ro.rasterize(st, lv["means3D"], None, shs=lv["shs"], sh_objs=lv["sh_objs"], opacities=lv["opacities"],
                         scales=lv["scales"], rotations=lv["rotations"], opt=ro.OracleOptions(lineage_grads=False))
        return (o.image * gi).sum() + (o.feats * gf).sum(), o

    loss, o = loss_of(leaves)
    loss.backward()
    rng = np.random.default_rng(0)
    for name in ("means3D", "opacities", "scales", "rotations", "sh_objs"):
        t = leaves[name]
        flat = t.detach().reshape(-1)
        checked = 0
        for idx in rng.permutation(flat.numel())[:12]:
            eps = 1e-6
            vals = []
            for sgn in (+1, -1):
                lv = {k: v.detach().clone() for k, v in leaves.items()}
                lv[name].reshape(-1)[idx] += sgn * eps
                vals.append(loss_of(lv)[0].item())
            fd = (vals[0] - vals[1]) / (2 * eps)
            an = t.grad.reshape(-1)[idx].item()
            if abs(fd - an) > 1e-4 * max(1.0, abs(fd)):   # a gate flipped inside the FD stencil: skip
                continue
            checked += 1
        assert checked >= 8, name


def test_lineage_switches_of_the_oracle():
    """OracleOptions.feats_bg / depth_normalised (SURVEY.md Appendix A switches; the HIP side's variant bits 0x10000 /
    0x20000 are compared against exactly this in tests/test_gpu_lineage.py): closed-form relations to the default render,
    and autograd through both (finite differences)."""
    act, cam = small_case(n=120, w=48, h=32, feat=4, seed=5, scale_mult=1.5)
    st = settings_for(cam, bg=(0.1, 0.2, 0.3))
    base = _render(act, st)
    fb = _render(act, st, opt=ro.OracleOptions(feats_bg=True, feat_bg_value=0.35))
    dn = _render(act, st, opt=ro.OracleOptions(depth_normalised=True))
    np.testing.assert_allclose(fb.feats.numpy(), (base.feats + 0.35 * base.final_T[None]).numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(fb.image.numpy(), base.image.numpy(), rtol=0, atol=0)
    A = (1.0 - base.final_T)[None]
    want = torch.where(A > 1e-10, base.depth / A.clamp_min(1e-10), torch.zeros_like(base.depth))
    np.testing.assert_allclose(dn.depth.numpy(), want.numpy(), rtol=0, atol=1e-12)
    covered = base.final_T < 0.5
    assert bool(covered.any()) and float(dn.depth[0][covered].min()) > 1.5     # a normalised depth is a depth of the scene
    # gradients: d/d opacity of sum(feats) and sum(depth) through the switches against finite differences
    op = act["opacities"].double().clone().requires_grad_(True)
    opt = ro.OracleOptions(feats_bg=True, feat_bg_value=0.35, depth_normalised=True, lineage_grads=False)

    def loss_of(o_):
        r = ro.rasterize(st, act["means3D"], None, shs=act["shs"], sh_objs=act["sh_objs"], opacities=o_, scales=act["scales"],
                         rotations=act["rotations"], opt=opt)
        return r.feats.sum() + r.depth.sum()
    loss_of(op).backward()
    rng = np.random.default_rng(1)
    ok = 0
    for idx in rng.permutation(op.numel())[:10]:
        eps = 1e-6
        v = []
        for sgn in (+1, -1):
            o2 = op.detach().clone()
            o2.reshape(-1)[idx] += sgn * eps
            v.append(loss_of(o2).item())
        fd = (v[0] - v[1]) / (2 * eps)
        if abs(fd - op.grad.reshape(-1)[idx].item()) <= 1e-4 * max(1.0, abs(fd)):
            ok += 1
    assert ok >= 7
