"""render()'s remaining call patterns on the FUSED path: override_color (gaussian_renderer/__init__.py:112-113;
render.py:240,296,344), mask (:123-135), is_6dof (:75-80), pipe.convert_SHs_python (:103-108), pipe.compute_cov3D_python (:93-94).
Until round 6 these fell back to the reference's composition of ~10 PyTorch kernels + gathers around the operator.

(a) the fused result equals the operator-level composition (the branch `_fusable` used to send these calls to; pinned against the
    imported reference by tests/test_gpu_render_prep.py) -- maps, radii (the SUBSET under a mask), and every gradient, including the
    ones only these patterns have: dL/doverride_color, dL/d(the (N,4,4) transforms), and means2D.grad at full size under a mask;
(b) a forward under torch.no_grad() (which skips the stores only a backward reads) gives bit-identical maps;
(c) every one of the eleven reference-generated call patterns of render_prep.npz takes the fused path."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

N, W, H = 3000, 160, 96


def _setup(dev, seed=5):
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    pc = SynthGaussianModel(make_scene(N, feat_dim=32, seed=seed, scale_mult=0.8).to(dev))
    cam = orbit_camera(W, H, angle=0.7, fid=0.4).to(dev)
    g = torch.Generator().manual_seed(seed + 1)
    d = [(0.02 * torch.randn(N, c, generator=g)).to(dev) for c in (3, 4, 3)]
    # a rigid-ish (N,4,4) transform per Gaussian: identity + small rotation-like and translation terms, last row (0,0,0,1) + noise
    T = torch.eye(4).repeat(N, 1, 1) + 0.03 * torch.randn(N, 4, 4, generator=g)
    T[:, 3, :3] *= 0.1
    oc = torch.rand(N, 3, generator=g).to(dev)
    mask = (torch.rand(N, generator=g) < 0.6).to(dev)
    gi = (torch.randn(3, H, W, generator=g) / (W * H)).to(dev)
    gf = (torch.randn(32, H, W, generator=g) / (W * H)).to(dev)
    return pc, cam, SynthPipe(), d, T.to(dev), oc, mask, gi, gf


PATTERNS = ["override", "mask", "sixdof", "sixdof_float", "shs_python", "cov_python", "mask_override", "mask_sixdof"]


def _run(pattern, fused, monkeypatch, dev):
    from trase_amd import renderer
    pc, cam, pipe, d, T, oc, mask, gi, gf = _setup(dev)
    if not fused:
        monkeypatch.setattr(renderer, "_fusable", lambda *a, **k: False)
    leaves = [x.clone().requires_grad_(True) for x in d]
    kw, extra = {}, {}
    dx = leaves[0]
    if "override" in pattern:
        extra["oc"] = oc.clone().requires_grad_(True)
        kw["override_color"] = extra["oc"]
    if "mask" in pattern:
        kw["mask"] = mask
    if pattern.endswith("sixdof"):
        extra["T"] = T.clone().requires_grad_(True)
        dx, kw["is_6dof"] = extra["T"], True
    if pattern == "sixdof_float":
        dx, kw["is_6dof"] = 0.0, True
    pipe.convert_SHs_python = pattern == "shs_python"
    pipe.compute_cov3D_python = pattern == "cov_python"
    for p in pc.parameters():
        p.grad = None
    out = renderer.render(cam, pc, pipe, torch.tensor([0.1, 0.2, 0.3], device=dev), dx, leaves[1], leaves[2], **kw)
    torch.autograd.backward([out["render"], out["render_gaussian_features"]], [gi, gf])
    grads = {f"p{k}": (p.grad.clone() if p.grad is not None else None) for k, p in enumerate(pc.parameters())}
    grads.update({f"d{k}": (x.grad.clone() if x.grad is not None else None) for k, x in enumerate(leaves)})
    grads.update({k: (v.grad.clone() if v.grad is not None else None) for k, v in extra.items()})
    grads["m2d"] = out["viewspace_points"].grad.clone()
    maps = {k: out[k].detach().clone() for k in ("render", "render_gaussian_features", "depth", "radii", "visibility_filter")}
    monkeypatch.undo()
    return maps, grads, int(mask.sum())


@pytest.mark.parametrize("pattern", PATTERNS)
def test_fused_call_pattern_matches_the_operator_level_composition(pattern, monkeypatch):
    dev = torch.device("cuda", 0)
    ma, ga, nmask = _run(pattern, True, monkeypatch, dev)
    mb, gb, _ = _run(pattern, False, monkeypatch, dev)
    assert torch.equal(ma["radii"], mb["radii"]) and torch.equal(ma["visibility_filter"], mb["visibility_filter"])
    assert ma["radii"].shape[0] == (nmask if "mask" in pattern else N)
    assert int((ma["radii"] > 0).sum()) > 100
    for k in ("render", "render_gaussian_features", "depth"):
        err = (ma[k] - mb[k]).abs().amax(0)
        assert (err > 2e-5).float().mean().item() < 2e-3 and err.max().item() < 5e-2, f"{pattern}: {k} {err.max().item():.3e}"
    assert set(ga) == set(gb)
    for k in ga:
        a, b = ga[k], gb[k]
        if a is None or b is None:
            # a tensor that took no part: None on one side may be exact zeros on the other (autograd leaves .grad untouched)
            other = b if a is None else a
            assert other is None or float(other.abs().max()) == 0.0, f"{pattern}: gradient {k} exists on one path only"
            continue
        assert a.shape == b.shape, (pattern, k)
        num, den = float((a - b).norm()), float(b.norm())
        assert num <= 2e-4 * den + 1e-12, f"{pattern}: gradient {k} rel-L2 {num / max(den, 1e-30):.3e}"
    assert float(ga["m2d"].abs().max()) > 0 and tuple(ga["m2d"].shape) == (N, 3)          # full size, also under a mask (Appendix C.2)
    if "override" in pattern:
        assert ga["oc"] is not None and float(ga["oc"].abs().max()) > 0
    if pattern.endswith("sixdof"):
        assert ga["T"] is not None and float(ga["T"].abs().max()) > 0 and ga["d0"] is None


@pytest.mark.parametrize("pattern", ["plain", "override", "mask_override", "sixdof"])
def test_forward_under_no_grad_is_bit_identical_and_fused(pattern, monkeypatch):
    from trase_amd import renderer
    dev = torch.device("cuda", 0)
    pc, cam, pipe, d, T, oc, mask, _, _ = _setup(dev, seed=9)
    kw = {}
    dx = d[0]
    if "override" in pattern:
        kw["override_color"] = oc
    if "mask" in pattern:
        kw["mask"] = mask
    if pattern == "sixdof":
        dx, kw["is_6dof"] = T, True
    seen = []
    real = renderer._fusable
    monkeypatch.setattr(renderer, "_fusable", lambda *a, **k: seen.append(real(*a, **k)) or seen[-1])
    bg = torch.zeros(3, device=dev)
    a = renderer.render(cam, pc, pipe, bg, dx, d[1], d[2], **kw)
    with torch.no_grad():
        b = renderer.render(cam, pc, pipe, bg, dx, d[1], d[2], **kw)
    assert seen == [True, True]
    for k in ("render", "render_gaussian_features", "depth", "radii"):
        assert torch.equal(a[k].detach(), b[k]), f"{pattern}: {k} differs between the grad-enabled and the no_grad forward"
    assert float(a["render"].abs().max()) > 0


def test_every_reference_call_pattern_takes_the_fused_path(monkeypatch):
    from tests import test_gpu_render_prep as RP
    from trase_amd import renderer
    dev = torch.device("cuda", 0)
    real = renderer._fusable
    for name in RP.CASES:
        seen = []
        monkeypatch.setattr(renderer, "_fusable", lambda *a, **k: seen.append(real(*a, **k)) or seen[-1])
        pc = RP._model(dev)
        cam, pipe, d, kw = RP._call(name, dev, pc)
        with torch.no_grad():
            renderer.render(cam, pc, pipe, RP._T("bg", dev), *d, **kw)
        monkeypatch.undo()
        assert seen == [True], f"call pattern {name!r} fell back to the operator-level composition"
