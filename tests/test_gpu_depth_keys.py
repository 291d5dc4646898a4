"""Depth-sort keys (round 5): the default 27-bit key (float bits above the 0.2 near plane, three 9-bit radix passes) against the
raw float32 bits (four 8-bit passes): identical point lists, maps and gradients wherever no key saturates, and an automatic,
exact fallback where one does (view depth > 13 107)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _render_once(n, w, h, seed, world_scale=1.0, grads=True):
    from gaussian_renderer import render
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    dev = torch.device("cuda", 0)
    pc = SynthGaussianModel(make_scene(n, feat_dim=32, seed=seed, scale_mult=0.8, world_scale=world_scale).to(dev))
    cam = orbit_camera(w, h, angle=0.7, radius=4.0 * world_scale, zfar=100.0 * world_scale).to(dev)
    out = render(cam, pc, SynthPipe(), torch.zeros(3, device=dev), 0.0, 0.0, 0.0)
    res = [out["render"].detach().clone(), out["render_gaussian_features"].detach().clone(), out["depth"].detach().clone(),
           out["radii"].clone()]
    if grads:
        g = torch.Generator().manual_seed(seed)
        gi = torch.randn(3, h, w, generator=g).to(dev)
        gf = torch.randn(32, h, w, generator=g).to(dev)
        torch.autograd.backward([out["render"], out["render_gaussian_features"]], [gi, gf])
        res += [p.grad.clone() for p in pc.parameters()]
    return res


def test_27_bit_keys_give_the_float_key_order():
    from trase_amd import rasterizer as R
    try:
        R.set_depth_keys(27)
        a = _render_once(30_000, 320, 200, seed=3)
        assert not (R._Policy.variant & R.VARIANT_DEPTH32)
        R.set_depth_keys(32)
        b = _render_once(30_000, 320, 200, seed=3)
    finally:
        R.set_depth_keys(27)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"tensor {i} differs between the 27-bit and the 32-bit depth keys"


def test_saturated_depth_key_switches_to_float_keys_sync_policy():
    """A scene 16 000 units deep: the 27-bit key saturates, the synchronising policy repeats stage 1 on the raw float bits at
    once and stays there; the result equals the one of a process that used 32-bit keys from the start."""
    from trase_amd import rasterizer as R
    try:
        R.set_sync(True)
        R.set_depth_keys(32)
        want = _render_once(5000, 200, 120, seed=5, world_scale=4000.0)
        assert float(want[2].max()) > 13107.0, "the scene is not deep enough to saturate a key"
        R.set_depth_keys(27)
        got = _render_once(5000, 200, 120, seed=5, world_scale=4000.0)
        assert R._Policy.variant & R.VARIANT_DEPTH32, "the saturated key went unnoticed"
    finally:
        R.set_depth_keys(27)
    for i, (x, y) in enumerate(zip(got, want)):
        assert torch.equal(x, y), f"tensor {i}: the repeated forward differs from a 32-bit-key forward"


def test_saturated_depth_key_is_reported_by_the_sync_free_policy():
    from trase_amd import rasterizer as R
    try:
        R.set_sync(True)
        R.set_depth_keys(32)
        want = _render_once(5000, 200, 120, seed=5, world_scale=4000.0, grads=False)
        cap = R.last_status()[2]
        R.set_depth_keys(27)
        R.set_sync(False, capacity=2 * cap + 1024)
        _render_once(5000, 200, 120, seed=5, world_scale=4000.0, grads=False)
        with pytest.raises(RuntimeError, match="27-bit depth keys"):
            R.check_overflow()
        assert R._Policy.variant & R.VARIANT_DEPTH32
        got = _render_once(5000, 200, 120, seed=5, world_scale=4000.0, grads=False)      # every later forward: float keys
        R.check_overflow()
    finally:
        R.set_depth_keys(27)
        R.set_sync(True)
    for i, (x, y) in enumerate(zip(got, want)):
        assert torch.equal(x, y), f"tensor {i}"
