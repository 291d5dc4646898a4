"""A WHOLE training iteration inside one torch.cuda.CUDAGraph.  Round 5 turned the zero-fills of the
library's own graphed launch sequences into kernels because a captured hipMemsetAsync stopped clearing from the second replay on
(ROCm 7.0.2); the loss heads, KNN and densify entry points still issued memsets, so a user capturing a whole iteration re-entered
the defect silently.  Every zero-fill is a kernel now; this test captures one FEATURE-state iteration (MLP under no_grad ->
render with KNN-smoothed, normalised features -> mask statistics + contrastive head + norm regulariser -> backward) and one
GAUSSIAN-state iteration (MLP with gradients -> render -> L1 + SSIM -> backward), replays each eight times and compares every
replay with the eager iteration bit for bit.  Workspaces are poisoned (0xFF / NaN), so a clear that does not happen shows."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

N, W, H = 20_000, 480, 270


def _scene(dev):
    from trase_amd.synthetic import SynthGaussianModel, make_scene, orbit_camera
    pc = SynthGaussianModel(make_scene(N, feat_dim=32, seed=3, scale_mult=0.8).to(dev))
    cams = [orbit_camera(W, H, angle=0.4, fid=0.3).to(dev)]
    return pc, cams


def _digest(ts):
    return [t.detach().clone() for t in ts]


def _run(state, monkeypatch):
    from trase_amd import rasterizer as R
    from trase_amd.bench_iterations import capture_iteration, make_feature_iteration, make_gaussian_iteration
    from trase_amd import feature_head as FH
    monkeypatch.setattr(R, "_POISON", True)
    dev = torch.device("cuda", 0)
    pc, cams = _scene(dev)
    restore = lambda: None
    if state == "feature":
        # the sampler's draws are made ONCE, outside the iteration (the eager run and every replay then see the same pixels);
        # the smoothing's CPU randperm is re-seeded inside the iteration -- the capture freezes that draw anyway
        real = FH.get_sample_pixel_and_mask
        fixed = {}

        def frozen(sam, npx, nm, cover_count=None, rng="cpu"):
            if "v" not in fixed:
                fixed["v"] = real(sam, npx, nm, cover_count=cover_count, rng=rng)
            sp, sm = fixed["v"]
            return sp, sm
        monkeypatch.setattr(FH, "get_sample_pixel_and_mask", frozen)       # (before the harness binds the name)
        it0, restore = make_feature_iteration(pc, cams, W, H, dev, n_masks=30)

        def it(i):
            torch.default_generator.manual_seed(11)            # CPU generator only: the smoothing's randperm(K)
            out = it0(i)
            return out, [pc._gaussian_features.grad, out["render_gaussian_features"], out["render"], out["radii"]]
    else:
        it0 = make_gaussian_iteration(pc, cams, W, H, dev, image_scope=True)

        def it(i):
            out = it0(i)
            return out, [p.grad for p in pc.parameters() if p.grad is not None] + [out["render"], out["radii"], out["viewspace_points"].grad]
    R.set_sync(True)
    try:
        it(0)
        cap = R.last_status()[2]
        R.set_sync(False, capacity=2 * cap + 4096)
        for _ in range(2):
            it(0)
        o_, ts = it(0)
        eager = _digest(ts)
        del o_, ts                                               # (no autograd graph of an eager iteration may outlive this point)
        R.check_overflow()
        graph, (_, static) = capture_iteration(it, warm=2)
        for k in range(8):
            for t in static:                                     # a replay that wrote nothing would otherwise pass on the capture's values
                if t.dtype.is_floating_point:           # (.data: some are views handed out by an autograd node)
                    t.data.fill_(float("nan"))
                else:
                    t.data.fill_(-7)
            graph.replay()
            torch.cuda.synchronize()
            assert len(static) == len(eager)
            for j, (a, b) in enumerate(zip(eager, static)):
                assert torch.equal(a, b), f"{state}: replay {k}, tensor {j} differs from the eager iteration " \
                                          f"(max |d| {float((a.float() - b.float()).abs().max()):.3e})"
        assert float(eager[0].abs().max()) > 0
    finally:
        restore()
        R.set_sync(True)


def test_feature_iteration_captured_whole_replays_bit_identically(monkeypatch):
    _run("feature", monkeypatch)


def test_gaussian_iteration_captured_whole_replays_bit_identically(monkeypatch):
    _run("gaussian", monkeypatch)
