"""gs_math.h (the arithmetic the HIP preprocess kernels run) compiled for the host and checked
against the float64 oracle: forward element-wise, backward against torch.autograd."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from tests.util import fptr, hs_view, np32, settings_for, small_case


def _run_forward(hostsim, st, act, colors=None, cov=None):
    n = act["means3D"].shape[0]
    out = np.zeros((n, 15), dtype=np.float32)
    v = hs_view(st)
    p = np32(act["means3D"]); sc = np32(act["scales"]); q = np32(act["rotations"])
    sh = None if colors is not None else np32(act["shs"])
    col = np32(colors)
    cv = np32(cov)
    hostsim.hs_forward(C.byref(v), n, fptr(p), fptr(None if cov is not None else sc),
                       fptr(None if cov is not None else q), fptr(cv), fptr(sh), fptr(col), fptr(out))
    return out


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
@pytest.mark.parametrize("d_rot", [0.0, 0.2])
def test_forward_matches_oracle(hostsim, deg, d_rot):
    act, cam = small_case(n=600, w=128, h=80, seed=deg, d_rot=d_rot)
    st = settings_for(cam, sh_degree=deg, scale_modifier=1.1)
    out = _run_forward(hostsim, st, act)
    g = ro.preprocess(st, act["means3D"].double(), act["shs"].double(), None, act["opacities"].double(),
                      act["scales"].double(), act["rotations"].double(), None)
    ok = ~g.frag_gauss
    vis = g.valid & ok
    radius = torch.from_numpy(out[:, 9]).to(torch.int32)
    assert vis.sum() > 100
    assert torch.equal(radius[ok], g.radii[ok])
    sel = vis.numpy()
    np.testing.assert_allclose(out[sel, 0:2], g.xy[vis].numpy(), rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(out[sel, 2], g.depth[vis].numpy(), rtol=1e-5)
    np.testing.assert_allclose(out[sel, 3:6], g.conic[vis].numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(out[sel, 6:9], g.rgb[vis].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(out[sel, 10:14], g.rect[vis].numpy().astype(np.float32))


def test_forward_precomputed_inputs(hostsim):
    act, cam = small_case(n=300, w=64, h=64, seed=5)
    st = settings_for(cam, sh_degree=3)
    cov = ro.cov3d_from_scale_rot(act["scales"].double(), act["rotations"].double(), 1.0).float()
    colors = torch.rand(300, 3)
    out = _run_forward(hostsim, st, act, colors=colors, cov=cov)
    g = ro.preprocess(st, act["means3D"].double(), None, colors.double(), act["opacities"].double(), None, None,
                      cov.double())
    vis = (g.valid & ~g.frag_gauss)
    np.testing.assert_allclose(out[vis.numpy(), 3:6], g.conic[vis].numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(out[vis.numpy(), 6:9], colors[vis].numpy(), rtol=0, atol=0)


@pytest.mark.parametrize("deg,use_cov", [(3, False), (1, False), (3, True)])
def test_backward_matches_autograd(hostsim, deg, use_cov):
    n = 500
    act, cam = small_case(n=n, w=128, h=96, seed=11 + deg, d_rot=0.15)
    st = settings_for(cam, sh_degree=deg, scale_modifier=0.9)
    p = act["means3D"].double().requires_grad_(True)
    sc = act["scales"].double().requires_grad_(True)
    q = act["rotations"].double().requires_grad_(True)
    sh = act["shs"].double().requires_grad_(True)
    m2d = torch.zeros(n, 3, dtype=torch.float64, requires_grad=True)
    cov = None
    if use_cov:
        cov = ro.cov3d_from_scale_rot(sc.detach(), q.detach(), st.scale_modifier).requires_grad_(True)
    g = ro.preprocess(st, p, sh, None, act["opacities"].double(), None if use_cov else sc, None if use_cov else q,
                      cov, means2D=m2d)
    gen = torch.Generator().manual_seed(3)
    gconic = torch.randn(n, 3, generator=gen, dtype=torch.float64)
    grgb = torch.randn(n, 3, generator=gen, dtype=torch.float64)
    gndc = torch.randn(n, 2, generator=gen, dtype=torch.float64)
    vis = g.valid
    loss = ((g.conic * gconic).sum(-1) + (g.rgb * grgb).sum(-1) + (g.ndc * gndc).sum(-1))[vis].sum()
    loss.backward()
    # forward on the host build to get the clamp bits
    fwd = _run_forward(hostsim, st, {k: (v.detach().float() if torch.is_tensor(v) else v) for k, v in
                                     dict(means3D=p, scales=sc, rotations=q, shs=sh).items()},
                       cov=None if not use_cov else cov.detach().float())
    gin = np.zeros((n, 9), dtype=np.float32)
    gin[:, 0:3] = gconic.numpy(); gin[:, 3:5] = gndc.numpy(); gin[:, 5:8] = grgb.numpy()
    gout = np.zeros((n, 16), dtype=np.float32)
    dsh = np.zeros((n, 16, 3), dtype=np.float32)
    v = hs_view(st)
    a_p = np32(p); a_s = np32(sc); a_q = np32(q); a_sh = np32(sh); a_cov = np32(cov) if use_cov else None
    clamped = np.ascontiguousarray(fwd[:, 14])
    hostsim.hs_backward(C.byref(v), n, fptr(a_p), fptr(None if use_cov else a_s), fptr(None if use_cov else a_q),
                        fptr(a_cov), fptr(a_sh), fptr(clamped), fptr(gin), fptr(gout), fptr(dsh))
    sel = (vis & ~g.frag_gauss).numpy()
    assert sel.sum() > 100

    def close(a, b, what, rtol=2e-3):
        a = a[sel]; b = b[sel]
        scale = np.abs(b).max(axis=-1, keepdims=True) + 1e-6
        err = np.abs(a - b) / scale
        assert err.max() < rtol, f"{what}: max rel err {err.max():.3e}"

    close(gout[:, 0:3], p.grad.numpy(), "d_means3D")
    close(dsh.reshape(n, -1), sh.grad.numpy().reshape(n, -1), "d_shs")
    assert np.abs(m2d.grad.numpy()[:, :2][sel] - gndc.numpy()[sel]).max() == 0
    if use_cov:
        close(gout[:, 10:16], cov.grad.numpy(), "d_cov3D")
    else:
        close(gout[:, 3:6], sc.grad.numpy(), "d_scales")
        close(gout[:, 6:10], q.grad.numpy(), "d_rotations")
