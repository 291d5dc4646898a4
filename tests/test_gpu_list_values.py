"""What the sub-tile sort carries per (sub-tile, Gaussian) pair (trase_amd/csrc/common.h HDR_PACK): by default the packed
value (Gaussian id << jb) | index of the pair among the Gaussian's own pairs, so that the compositing kernels get the id
with a shift; the emit-order slot (ids through a second array) when some Gaussian has 2^jb pairs or more, or when a
variant that needs the slot form is selected.  The choice is made on the device and must never change a result."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NO_PACK = 0x100000


def _pack_bits():
    from trase_amd import rasterizer as R
    return int(R._Policy.last_geom[:256].view(torch.int32)[3].item())      # header word HDR_PACK


def _step(act, st, g_img, g_feat):
    from tests.test_gpu_fullsize import _render
    (img, radii, feats, depth), a, m2d = _render(act, st, need_grad=True)
    bits = _pack_bits()
    torch.autograd.backward([img, feats], [g_img, g_feat])
    out = {"img": img.detach().clone(), "feats": feats.detach().clone(), "depth": depth.detach().clone(),
           "radii": radii.clone(), "means2D": m2d.grad.clone()}
    out.update({k: v.grad.clone() for k, v in a.items() if v.grad is not None})
    return out, bits


def _ab(act, st, w, h, dev):
    from trase_amd import rasterizer as R
    torch.manual_seed(11)
    g_img = torch.randn(3, h, w, device=dev)
    g_feat = torch.randn(32, h, w, device=dev)
    res = {}
    try:
        for name, var in (("auto", 0), ("slots", NO_PACK)):
            R.set_variant(var)
            res[name] = _step(act, st, g_img, g_feat)
    finally:
        R.set_variant(0)
    return res


def test_packed_list_values_change_nothing():
    from tests.test_gpu_fullsize import _setup
    n, w, h = 60_000, 640, 360
    act, cam, dev, settings_for = _setup(n, w, h, seed=4)
    st = settings_for(cam, device=dev)
    res = _ab(act, st, w, h, dev)
    assert res["auto"][1] == 16 and res["slots"][1] == 0          # 60 000 ids need 16 bits: 16 are left for the pair index
    for k, v in res["auto"][0].items():
        assert torch.equal(v, res["slots"][0][k]), k
    assert float(res["auto"][0]["sh_objs"].abs().max()) > 0


def test_a_gaussian_with_too_many_pairs_falls_back_to_slots():
    """2^20 + 1 Gaussians leave 11 bits for the pair index; one Gaussian that covers all 3600 sub-tiles of the image
    has more pairs than that: the device picks the slot form (header word 0) and the results equal the forced-slot run."""
    from tests.test_gpu_fullsize import _setup
    n, w, h = (1 << 20) + 1, 640, 360
    act, cam, dev, settings_for = _setup(n, w, h, seed=6, scale_mult=0.1)
    act = dict(act)
    for k in ("means3D", "scales", "opacities"):
        act[k] = act[k].clone()
    act["means3D"][7] = 0.0
    act["scales"][7] = 40.0
    act["opacities"][7] = 0.5
    st = settings_for(cam, device=dev)
    res = _ab(act, st, w, h, dev)
    assert res["auto"][1] == 0 and res["slots"][1] == 0
    from trase_amd import rasterizer as R
    assert int(R.last_geom_view(n)["tiles"][7]) >= 2048
    for k, v in res["auto"][0].items():
        assert torch.equal(v, res["slots"][0][k]), k
    # without the giant the same scene packs
    act["scales"][7] = 0.01
    res2 = _ab(act, st, w, h, dev)
    assert res2["auto"][1] == 11
