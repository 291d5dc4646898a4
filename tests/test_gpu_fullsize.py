"""Size-independent properties at BASELINE.json's full sizes (the oracle is far too slow there):
energy conservation of the blend weights, linearity in the colours, bit-reproducibility (forward AND the
atomic-free backward), invariance to the order in which Gaussians are stored, gradient sum rules."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n, w, h, seed=0, scale_mult=0.27, feat=32):
    from trase_amd.synthetic import make_scene, orbit_camera
    from tests.util import settings_for
    dev = torch.device("cuda", 0)
    scene = make_scene(n, feat_dim=feat, seed=seed, scale_mult=scale_mult).to(dev)
    cam = orbit_camera(w, h, angle=0.3)
    return scene.activated(), cam, dev, settings_for


def _render(act, st, colors=None, need_grad=False, perm=None):
    from diff_gaussian_rasterization import GaussianRasterizer
    a = {k: (v if perm is None else v[perm]).clone().requires_grad_(need_grad) for k, v in act.items()}
    m2d = torch.zeros_like(a["means3D"], requires_grad=need_grad)
    kw = dict(means3D=a["means3D"], means2D=m2d, sh_objs=a["sh_objs"], opacities=a["opacities"], scales=a["scales"],
              rotations=a["rotations"])
    if colors is None:
        kw["shs"] = a["shs"]
    else:
        kw["colors_precomp"] = (colors if perm is None else colors[perm])
    out = GaussianRasterizer(raster_settings=st)(**kw)
    return out, a, m2d


@pytest.mark.parametrize("n,w,h", [(300_000, 1920, 1080), (150_000, 480, 270)])
def test_weights_sum_to_one_minus_transmittance(n, w, h):
    """colours == 1, features == 1, background == 1  =>  image = sum(w) + T_final = 1 exactly (telescoping),
    and the feature map (no background term) equals 1 - T_final."""
    act, cam, dev, settings_for = _setup(n, w, h)
    st = settings_for(cam, bg=(1.0, 1.0, 1.0), device=dev)
    act = dict(act)
    act["sh_objs"] = torch.ones_like(act["sh_objs"])
    (img, radii, feats, depth), _, _ = _render(act, st, colors=torch.ones(n, 3, device=dev))
    assert (img - 1.0).abs().max().item() < 2e-5
    assert (feats - feats[0:1]).abs().max().item() == 0.0          # all 32 channels identical
    assert feats.min().item() >= 0.0 and feats.max().item() <= 1.0 + 1e-5
    assert (img[0] - (feats[0] + (1.0 - feats[0]))).abs().max().item() < 2e-5
    assert int((radii > 0).sum()) > 0.3 * n
    assert depth.min().item() >= 0.0


def test_linearity_in_colours_and_determinism_full_size():
    n, w, h = 300_000, 1920, 1080
    act, cam, dev, settings_for = _setup(n, w, h)
    st = settings_for(cam, bg=(0.0, 0.0, 0.0), device=dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    c1, c2 = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    (i1, r1, f1, d1), _, _ = _render(act, st, colors=c1)
    (i2, _, _, _), _, _ = _render(act, st, colors=c2)
    (i12, _, f12, d12), _, _ = _render(act, st, colors=0.25 * c1 + 0.75 * c2)
    # geometry-only weights, black bg; the forward accumulates the channels as bf16-split MFMA products (~2^-16 relative
    # per term), so linearity holds to a few 1e-5 of the colour scale -- well inside the 1e-4 parity bar
    assert (i12 - (0.25 * i1 + 0.75 * i2)).abs().max().item() < 4e-5
    assert torch.equal(f1, f12) and torch.equal(d1, d12)                   # colours do not touch feats/depth
    (i1b, r1b, f1b, d1b), _, _ = _render(act, st, colors=c1)
    assert torch.equal(i1, i1b) and torch.equal(f1, f1b) and torch.equal(d1, d1b) and torch.equal(r1, r1b)


def test_backward_is_bit_reproducible_and_obeys_sum_rules():
    n, w, h = 300_000, 1920, 1080
    act, cam, dev, settings_for = _setup(n, w, h)
    st = settings_for(cam, bg=(0.2, 0.4, 0.6), device=dev)
    g = torch.Generator(device="cpu").manual_seed(7)
    gi = (torch.randn(3, h, w, generator=g) / (w * h)).to(dev)
    gf = (torch.randn(32, h, w, generator=g) / (w * h)).to(dev)
    grads = []
    for _ in range(2):
        (img, radii, feats, depth), a, m2d = _render(act, st, need_grad=True)
        torch.autograd.backward([img, feats], [gi, gf])
        grads.append({k: v.grad.clone() for k, v in a.items()} | {"means2D": m2d.grad.clone()})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), f"{k}: backward is not bit-reproducible"
        assert torch.isfinite(grads[0][k]).all(), k
    # culled Gaussians receive exactly zero gradient; means2D.grad has z == 0
    culled = radii == 0
    assert float(grads[0]["means3D"][culled].abs().max()) == 0.0
    assert float(grads[0]["sh_objs"][culled].abs().max()) == 0.0
    assert float(grads[0]["means2D"][:, 2].abs().max()) == 0.0
    # d(loss)/d(feature c of Gaussian i) = sum_p w_pi * gf[c,p]; summing over Gaussians with unit features
    # in place of gf gives sum_p sum_i w_pi = sum_p (1 - T_p): check through a second render
    ones = torch.ones(32, h, w, device=dev) / (w * h)
    (img, radii, feats, depth), a, _ = _render(act, st, need_grad=True)
    feats.backward(ones)
    act1 = dict(act); act1["sh_objs"] = torch.ones_like(act["sh_objs"])
    (_, _, f_one, _), _, _ = _render(act1, st)
    lhs = a["sh_objs"].grad[:, 0, 0].double().sum().item()
    rhs = (f_one[0].double().sum() / (w * h)).item()
    assert abs(lhs - rhs) < 2e-4 * abs(rhs)


def test_storage_order_invariance():
    """The result may not depend on where a Gaussian sits in the input arrays (ties in depth aside):
    render a shuffled copy and un-shuffle."""
    n, w, h = 150_000, 480, 270
    act, cam, dev, settings_for = _setup(n, w, h, seed=3)
    st = settings_for(cam, bg=(0.1, 0.1, 0.1), device=dev)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0)).to(dev)
    (i0, r0, f0, d0), _, _ = _render(act, st)
    (i1, r1, f1, d1), _, _ = _render(act, st, perm=perm)
    assert torch.equal(r0[perm], r1)
    # exact depth ties between different Gaussians are broken by index, which the shuffle changes; that can
    # only move the result where two equal-depth Gaussians overlap -- allow a vanishing fraction of pixels
    diff = ((i0 - i1).abs().amax(0) > 1e-6) | ((f0 - f1).abs().amax(0) > 1e-6)
    assert diff.float().mean().item() < 1e-3


def test_binning_count_and_emit_agree_at_s3_size(monkeypatch):
    """Regression: the live-sub-tile count (preprocess) and its re-evaluation (emit) once disagreed for one pair
    in ~8 M (different FMA contraction of two inlined copies), leaving a hole in the pair list.  BASELINE config 3
    (1 M Gaussians, 1352x1014, ragged bottom sub-tile row) with poisoned workspaces; trase_rast_status raises if
    the binning guards trip; exact- and over-sized capacity must give identical images."""
    from trase_amd import rasterizer as R
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    from gaussian_renderer import render
    monkeypatch.setattr(R, "_POISON", True)
    dev = torch.device("cuda", 0)
    n, w, h = 1_000_000, 1352, 1014
    pc = SynthGaussianModel(make_scene(n, feat_dim=32, seed=0, scale_mult=0.27).to(dev), requires_grad=False)
    bg = torch.zeros(3, device=dev)
    try:
        with torch.no_grad():
            for k in (0, 5):
                cam = orbit_camera(w, h, angle=2 * math.pi * k / 16).to(dev)
                R.set_sync(True)
                a = render(cam, pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
                st = R.last_status()
                R.set_sync(False, capacity=int(st[2] * 1.25) + 1024)
                b = render(cam, pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
                assert R.last_status() == st
                assert torch.equal(a["render"], b["render"]) and torch.equal(a["render_gaussian_features"], b["render_gaussian_features"])
                assert torch.isfinite(a["render"]).all()
    finally:
        R.set_sync(True)


def test_mfma_backward_matches_fp32_valu_backward_full_size(monkeypatch):
    """The default F = 32 backward runs the channel contractions as bf16-split MFMA GEMMs (render_bwd_hw.hip, three
    products per term).  At the headline size (300k Gaussians, 1080p) every gradient must agree with the packed-FP32
    formulation of the same algorithm (render_bwd_gs.hip, `variant` bit 0x40) to 5e-5 of the gradient's scale -- the
    split is exact to ~2^-17 per operand -- and stay bit-reproducible."""
    from trase_amd import rasterizer as R
    act, cam, dev, settings_for = _setup(300_000, 1920, 1080)
    st = settings_for(cam, device=dev)
    torch.manual_seed(5)
    g_img = torch.randn(3, 1080, 1920, device=dev)
    g_feat = torch.randn(32, 1080, 1920, device=dev)
    grads = {}
    try:
        for name, var in (("valu", 0x40), ("mfma", 0), ("mfma2", 0)):
            R.set_variant(var)
            (img, radii, feats, depth), a, m2d = _render(act, st, need_grad=True)
            torch.autograd.backward([img, feats], [g_img, g_feat])
            grads[name] = {k: v.grad.clone() for k, v in a.items() if v.grad is not None}
            grads[name]["means2D"] = m2d.grad.clone()
    finally:
        R.set_variant(0)
    for k, ref in grads["valu"].items():
        scale = float(ref.abs().max())
        err = float((grads["mfma"][k] - ref).abs().max())
        assert err <= 5e-5 * scale, f"{k}: max abs diff {err:.3e} vs scale {scale:.3e}"
        assert torch.equal(grads["mfma"][k], grads["mfma2"][k]), f"{k}: MFMA backward is not bit-reproducible"


def test_image_only_mfma_backward_matches_fp32_valu_backward_full_size():
    """A GAUSSIAN-state iteration back-propagates an image-only loss (train.py:235-243, :299): no feature cotangent.  That
    call takes the image-only scope of the MFMA backward (render_bwd_hw.hip: GEMM 1 over channels 32..35 only, 12-column
    gradient rows); at the headline size every gradient must agree with the packed-FP32 kernel (TRASE_VARIANT_VALU_BACKWARD)
    to 5e-5 of its scale, be bit-reproducible, and the feature gradient must be exactly zero."""
    from trase_amd import rasterizer as R
    act, cam, dev, settings_for = _setup(300_000, 1920, 1080)
    st = settings_for(cam, bg=(0.2, 0.4, 0.6), device=dev)
    torch.manual_seed(6)
    g_img = torch.randn(3, 1080, 1920, device=dev)
    grads = {}
    try:
        for name, var in (("valu", 0x40), ("mfma", 0), ("mfma2", 0)):
            R.set_variant(var)
            (img, radii, feats, depth), a, m2d = _render(act, st, need_grad=True)
            torch.autograd.backward([img], [g_img])
            grads[name] = {k: v.grad.clone() for k, v in a.items() if v.grad is not None}
            grads[name]["means2D"] = m2d.grad.clone()
    finally:
        R.set_variant(0)
    assert float(grads["mfma"]["sh_objs"].abs().max()) == 0.0
    for k, ref in grads["valu"].items():
        scale = float(ref.abs().max())
        err = float((grads["mfma"][k] - ref).abs().max())
        assert err <= 5e-5 * scale + 0.0, f"{k}: max abs diff {err:.3e} vs scale {scale:.3e}"
        assert torch.equal(grads["mfma"][k], grads["mfma2"][k]), f"{k}: image-only MFMA backward is not bit-reproducible"


@pytest.mark.parametrize("name,n,w,h,with_mlp,loss", [
    ("config 3: Neu3D size, deform MLP, RGB + features", 1_000_000, 1352, 1014, True, "l1ssim+feat"),
    ("config 4: HyperNeRF size, features + contrastive loss", 300_000, 536, 960, False, "contrastive"),
    ("config 5 (one rank's share): Immersive size", 2_500_000, 1280, 960, False, "l1ssim+feat"),
])
def test_baseline_configs_run_end_to_end_and_reproducibly(name, n, w, h, with_mlp, loss):
    """BASELINE.json's other configurations as end-to-end iterations on one GPU (the oracle is far too slow at these
    sizes): deformation MLP (training pair) -> fused render() -> fused loss head -> backward, sync-free with a measured
    capacity.  Checks: no overflow and no binning guard, every gradient finite and non-zero, and the whole iteration --
    image, loss and every gradient -- bit-identical when repeated."""
    from trase_amd import rasterizer as R
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, SynthDeformNetwork, make_scene, orbit_camera
    from trase_amd.deform import DeformNetworkHIP
    from trase_amd.losses import l1_ssim, pixel_mask_correspondence_loss_soft_hard_positive as soft_pos, \
        pixel_mask_correspondence_loss_soft_negative as soft_neg
    from gaussian_renderer import render
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pc = SynthGaussianModel(make_scene(n, feat_dim=32, seed=2, scale_mult=0.27).to(dev))
    net = SynthDeformNetwork().to(dev)
    with torch.no_grad():
        for m in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling):
            m.weight.mul_(0.01); m.bias.zero_()
    hip_net = DeformNetworkHIP(net)
    cam = orbit_camera(w, h, angle=0.9).to(dev)
    bg = torch.zeros(3, device=dev)
    gt = torch.rand(3, h, w, device=dev)
    g_feat = torch.randn(32, h, w, device=dev) / (h * w)
    S = 1500
    pix = torch.randperm(h * w, device=dev)[:S]
    memb = (torch.rand(12, S, device=dev) < 0.2).float()
    Cm = (memb.t() @ memb != 0).float()
    params = pc.parameters() + (list(net.parameters()) if with_mlp else [])

    def iteration():
        for p in params:
            p.grad = None
        d = (0.0, 0.0, 0.0)
        if with_mlp:
            t = torch.tensor([[0.4]], device=dev).expand(n, -1)
            d = hip_net(pc.get_xyz.detach(), t)
        out = render(cam, pc, SynthPipe(), bg, *d)
        img, feats = out["render"], out["render_gaussian_features"]
        if loss == "contrastive":
            f = torch.nn.functional.normalize(feats.reshape(32, -1)[:, pix].t(), dim=-1)     # utils/feature_utils.py:57-63
            cf = f @ f.t()
            total = soft_pos(Cm, cf, 0.75) + soft_neg(Cm, cf, 0.5)
        else:
            l1, ss = l1_ssim(img, gt)
            total = 0.8 * l1 + 0.2 * (1.0 - ss) + (feats * g_feat).sum()
        total.backward()
        return img.detach().clone(), total.detach().clone(), [p.grad.clone() for p in params]

    try:
        R.set_sync(True)
        iteration()
        st = R.last_status()
        R.set_sync(False, capacity=int(st[2] * 1.25) + 1024)
        img_a, tot_a, g_a = iteration()
        st2 = R.last_status()                 # raises if a binning guard tripped
        assert st2[1] == 0 and st2[2] == st[2], name
        img_b, tot_b, g_b = iteration()
    finally:
        R.set_sync(True)
    assert torch.isfinite(img_a).all() and torch.isfinite(tot_a)
    assert torch.equal(img_a, img_b) and torch.equal(tot_a, tot_b), name
    colour = {id(pc._features_dc), id(pc._features_rest)}     # receive no gradient from a feature-only loss
    for p, ga, gb in zip(params, g_a, g_b):
        assert torch.isfinite(ga).all(), name
        assert float(ga.abs().max()) > 0 or (loss == "contrastive" and id(p) in colour), name
        assert torch.equal(ga, gb), name


def test_fifty_million_gaussians_index_arithmetic_beyond_2_to_31():
    """288 GB of HBM hold scenes whose tensors pass 2^31 ELEMENTS: at 50 M Gaussians `_features_rest` has 2.25e9 floats.  The
    per-Gaussian forward state (pixel centre, conic + opacity, colour + depth, radii) of the LAST thousand Gaussians must be
    bit-identical to the state the same thousand get as a scene of their own (every input tensor indexed with 64-bit arithmetic),
    and the backward leaves finite gradients with the same non-zero rows in all seven tensors' tails.  ~60 GB of device memory."""
    from gaussian_renderer import render
    from trase_amd import rasterizer as R
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, SynthScene, make_scene, orbit_camera
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(0).total_memory < 150e9:
        pytest.skip("needs an MI355X-sized device")
    import psutil
    if psutil.virtual_memory().available < 120e9:
        pytest.skip("needs ~40 GB of host memory for the 50 M-Gaussian scene (two copies while it is built)")
    N, K, W, H = 50_000_000, 1000, 960, 540
    scene = make_scene(N, feat_dim=32, seed=0, scale_mult=0.27)
    cam = orbit_camera(W, H, angle=0.3).to(dev)
    pc = SynthGaussianModel(scene.to(dev))
    bg = torch.zeros(3, device=dev)
    out = render(cam, pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
    gv = R.last_geom_view(N)
    big = {k: gv[k][N - K:].clone() for k in ("xy", "conic_opacity", "rgb_depth")}
    radii = out["radii"][N - K:].clone()
    g = torch.Generator().manual_seed(0)
    gi, gf = torch.randn(3, H, W, generator=g).to(dev), torch.randn(32, H, W, generator=g).to(dev)
    torch.autograd.backward([out["render"], out["render_gaussian_features"]], [gi, gf])
    rows = None
    for p in pc.parameters():
        assert torch.isfinite(p.grad[N - 100_000:]).all()
        nz = p.grad[N - K:].reshape(K, -1).abs().sum(1) > 0
        rows = nz if rows is None else rows
        assert torch.equal(nz, rows), "the tail rows that received a gradient differ between the parameter tensors"
    assert int(rows.sum()) > 0 and bool((radii[rows] > 0).all())
    del out, pc, gv
    torch.cuda.empty_cache()
    sub = SynthScene(*[t[N - K:].clone() for t in (scene.xyz, scene.features_dc, scene.features_rest, scene.scaling, scene.rotation,
                                                     scene.opacity, scene.gaussian_features)])
    out2 = render(cam, SynthGaussianModel(sub.to(dev)), SynthPipe(), bg, 0.0, 0.0, 0.0)
    gv2 = R.last_geom_view(K)
    vis = radii > 0
    assert int(vis.sum()) > 100 and torch.equal(radii, out2["radii"])
    for k in ("xy", "conic_opacity", "rgb_depth"):
        assert torch.equal(big[k][vis], gv2[k][vis]), k
