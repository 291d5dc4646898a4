"""The three lineage switches (where the absent fork of the CUDA extension is most likely to differ
from the public lineage) flip on BOTH sides -- HIP kernels (settings.variant bits 0x100 / 0x10000 / 0x20000,
``trase_amd.rasterizer.set_lineage``) and oracle (``OracleOptions``) -- and stay in parity: maps and every gradient."""
import pytest
import torch

from oracle import raster_oracle as ro
from tests import test_gpu_parity as T
from tests.util import settings_for, small_case

pytestmark = pytest.mark.gpu

NAMES = ["means3D", "means2D", "opacities", "scales", "rotations", "shs", "sh_objs"]


def _run(feat, set_kw, opt, with_depth_cot, seed=21, n=900, w=120, h=88):
    from trase_amd import rasterizer as R
    act, cam = small_case(n=n, w=w, h=h, feat=feat, seed=seed, scale_mult=0.9, d_rot=0.05)
    st = settings_for(cam, bg=(0.2, 0.4, 0.1))
    try:
        R.set_lineage(**set_kw)
        g, gl = T._gpu_call(act, st)
        o, ol = T._oracle_call(act, st, gpu=g, opt=opt)
        T._check_maps(g, o)
        gen = torch.Generator().manual_seed(seed)
        gi = T._masked(torch.randn(3, h, w, generator=gen), o)
        gf = T._masked(torch.randn(max(feat, 1), h, w, generator=gen), o)[:feat]
        gd = T._masked(torch.randn(1, h, w, generator=gen), o)
        loss = (o.image * gi.double()).sum()
        outs, cots = [g[0]], [gi.cuda()]
        if feat:
            loss = loss + (o.feats * gf.double()).sum()
            outs.append(g[2]); cots.append(gf.cuda())
        if with_depth_cot:
            loss = loss + (o.depth * gd.double()).sum()
        outs.append(g[3]); cots.append(gd.cuda())          # the device ALWAYS receives a depth cotangent
        loss.backward()
        torch.autograd.backward(outs, cots)
        T._check_grads(gl, ol, o, NAMES if feat else NAMES[:-1])
        return g, o
    finally:
        R.set_lineage()


@pytest.mark.parametrize("feat", [32, 16])
def test_default_lineage_ignores_the_depth_cotangent(feat):
    """Public lineage: the depth output carries no gradient -- a depth cotangent must not move any gradient."""
    _run(feat, {}, ro.OracleOptions(), with_depth_cot=False)


@pytest.mark.parametrize("feat", [32, 16, 0])
def test_switch_depth_gradient(feat):
    _run(feat, dict(depth_grad=True), ro.OracleOptions(), with_depth_cot=True)


@pytest.mark.parametrize("feat", [32, 16])
def test_switch_feature_background(feat):
    g, o = _run(feat, dict(feats_bg=0.35), ro.OracleOptions(feats_bg=True, feat_bg_value=0.35), with_depth_cot=False)
    # the switch did something: the feature map differs from the default lineage's by T_final * 0.35
    base, _ = _run(feat, {}, ro.OracleOptions(), with_depth_cot=False)
    want = 0.35 * o.final_T.float()
    assert float(((g[2] - base[2]).cpu() - want[None]).abs().max()) < 1e-5 and float(want.max()) > 1e-3


@pytest.mark.parametrize("feat,depth_grad", [(32, True), (32, False), (0, True)])
def test_switch_normalised_depth(feat, depth_grad):
    g, o = _run(feat, dict(depth_normalised=True, depth_grad=depth_grad), ro.OracleOptions(depth_normalised=True),
                with_depth_cot=depth_grad)
    covered = o.final_T < 0.5
    assert bool(covered.any())
    # normalised depth of a well-covered pixel lies inside the depth range of the scene (camera at radius 4, cube +-1.3)
    assert float(g[3][0][covered.cuda()].min()) > 1.5


def test_all_three_switches_together():
    _run(32, dict(feats_bg=-0.2, depth_normalised=True, depth_grad=True),
         ro.OracleOptions(feats_bg=True, feat_bg_value=-0.2, depth_normalised=True), with_depth_cot=True)
