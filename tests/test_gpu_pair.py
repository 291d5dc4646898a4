"""Two views per launch sequence (trase_rast_forward_raw_pair, trase_amd.renderer.render_views): the two
views' depth sorts are ONE sort of 2 P keys with the view index in the sign bit of the float32 depth key; everything else,
and the whole backward, runs per view.  Outputs, point lists and gradients must be bit-identical to two serial render() calls
-- at the headline size too -- and the gradients of the shared parameters must be the serial sum in a fixed order."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(n, w, h, pairwise, with_deform, seed=0):
    from gaussian_renderer import render
    from trase_amd import rasterizer as R
    from trase_amd.renderer import render_views
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    dev = torch.device("cuda", 0)
    pc = SynthGaussianModel(make_scene(n, feat_dim=32, seed=seed, scale_mult=0.27 if n > 50_000 else 0.8).to(dev))
    cams = [orbit_camera(w, h, angle=0.3 + 1.1 * k, fid=0.25 * k).to(dev) for k in range(2)]
    bg = torch.zeros(3, device=dev)
    pipe = SynthPipe()
    g = torch.Generator().manual_seed(seed + 1)
    gi = [(torch.randn(3, h, w, generator=g) / (w * h)).to(dev) for _ in range(2)]
    gf = [(torch.randn(32, h, w, generator=g) / (w * h)).to(dev) for _ in range(2)]
    if with_deform:      # each view has its own deformation (its own time), as in train.py:196-204
        d = [[(0.01 * torch.randn(n, c, generator=g)).to(dev).requires_grad_(True) for c in (3, 4, 3)] for _ in range(2)]
    else:
        d = [[0.0, 0.0, 0.0] for _ in range(2)]
    R.set_sync(True)
    caps = []
    with torch.no_grad():
        for c, dd in zip(cams, d):
            render(c, pc, pipe, bg, *[x.detach() if torch.is_tensor(x) else x for x in dd])
            caps.append(R.last_status()[2])
    R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
    try:
        for p in pc.parameters():
            p.grad = None
        if pairwise:
            outs = render_views(cams, pc, pipe, bg, [dd[0] for dd in d], [dd[1] for dd in d], [dd[2] for dd in d])
        else:
            outs = [render(c, pc, pipe, bg, *dd) for c, dd in zip(cams, d)]
        # one backward per view, view 0 first: the shared parameters accumulate in that fixed order either way
        for k, o in enumerate(outs):
            torch.autograd.backward([o["render"], o["render_gaussian_features"]], [gi[k], gf[k]])
        R.check_overflow()
        res = {"maps": [(o["render"].clone(), o["render_gaussian_features"].clone(), o["depth"].clone(), o["radii"].clone()) for o in outs],
               "grads": [p.grad.clone() for p in pc.parameters()],
               "m2d": [o["viewspace_points"].grad.clone() for o in outs],
               "vis": [o["visibility_filter"].clone() for o in outs],
               "d": [[x.grad.clone() for x in dd] for dd in d] if with_deform else []}
    finally:
        R.set_sync(True)
    return res


@pytest.mark.parametrize("n,w,h,with_deform", [(3000, 160, 96, False), (20_000, 320, 200, True), (300_000, 1920, 1080, True)],
                         ids=["small", "medium-deform", "S4-deform"])
def test_pair_forward_is_bit_identical_to_two_serial_views(n, w, h, with_deform):
    a = _run(n, w, h, pairwise=False, with_deform=with_deform)
    b = _run(n, w, h, pairwise=True, with_deform=with_deform)
    for k in range(2):
        for x, y, name in zip(a["maps"][k], b["maps"][k], ("image", "features", "depth", "radii")):
            assert torch.equal(x, y), f"view {k}: {name} differs between render_views and two render() calls"
        assert torch.equal(a["m2d"][k], b["m2d"][k]), f"view {k}: viewspace gradient differs"
        # the filter a training loop hands to add_densification_stats (it used to be formed before the deferred forward ran)
        assert torch.equal(a["vis"][k], b["vis"][k]) and torch.equal(b["vis"][k], b["maps"][k][3] > 0), f"view {k}: visibility_filter"
        assert bool(b["vis"][k].any())
    for i, (x, y) in enumerate(zip(a["grads"], b["grads"])):
        assert torch.equal(x, y), f"parameter {i}: gradient (sum over the two views) differs"
    for k in range(len(a["d"])):
        for x, y in zip(a["d"][k], b["d"][k]):
            assert torch.equal(x, y), f"view {k}: deformation gradient differs"
    assert float(a["maps"][0][0].abs().max()) > 0 and not torch.equal(a["maps"][0][0], a["maps"][1][0])


def test_render_views_handles_an_odd_view_and_unfused_views():
    from trase_amd import rasterizer as R
    from trase_amd.renderer import render_views
    from gaussian_renderer import render
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    dev = torch.device("cuda", 0)
    pc = SynthGaussianModel(make_scene(2000, feat_dim=32, seed=4, scale_mult=0.8).to(dev), requires_grad=False)
    cams = [orbit_camera(128, 80, angle=0.5 * k).to(dev) for k in range(3)]
    bg = torch.zeros(3, device=dev)
    R.set_sync(True)
    with torch.no_grad():
        render(cams[0], pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
        cap = R.last_status()[2]
        R.set_sync(False, capacity=4 * cap + 1024)
        try:
            serial = [render(c, pc, SynthPipe(), bg, 0.0, 0.0, 0.0)["render"].clone() for c in cams]
            batched = [o["render"] for o in render_views(cams, pc, SynthPipe(), bg, 0.0, 0.0, 0.0)]
            # a view that cannot take the fused path (an override colour) inside a pair: both views still come out right
            oc = torch.rand(2000, 3, device=dev)
            mixed_ref = [render(cams[0], pc, SynthPipe(), bg, 0.0, 0.0, 0.0, override_color=oc)["render"].clone(), serial[1]]
            mixed = [o["render"] for o in render_views(cams[:2], pc, SynthPipe(), bg, 0.0, 0.0, 0.0)]
        finally:
            R.set_sync(True)
    for x, y in zip(serial, batched):
        assert torch.equal(x, y)
    assert torch.equal(mixed[1], mixed_ref[1])
