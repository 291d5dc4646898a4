"""Launch-graph replay (trase_rast_graph_mode, trase_amd.rasterizer.set_graph): a forward / backward whose argument
record repeats is replayed as one hipGraph launch.  Results must be bit-identical to the eager launch sequence and the
records of a steady loop must actually repeat (hits)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loop(iters, graph):
    from gaussian_renderer import render
    from trase_amd import rasterizer as R
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    dev = torch.device("cuda", 0)
    pc = SynthGaussianModel(make_scene(3000, feat_dim=32, seed=2, scale_mult=0.8).to(dev))
    cams = [orbit_camera(160, 96, angle=0.3 * k).to(dev) for k in range(3)]
    bg = torch.zeros(3, device=dev)
    g = torch.Generator().manual_seed(0)
    gi, gf = torch.randn(3, 96, 160, generator=g).to(dev), torch.randn(32, 96, 160, generator=g).to(dev)
    R.set_sync(True)
    out = render(cams[0], pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
    cap = R.last_status()[2]
    R.set_sync(False, capacity=2 * cap + 1024)
    R.set_graph(graph)
    res = []
    try:
        for i in range(iters):
            for p in pc.parameters():
                p.grad = None
            out = render(cams[i % 3], pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
            torch.autograd.backward([out["render"], out["render_gaussian_features"]], [gi, gf])
            # digests instead of clones: held device memory would change the allocator's pattern (and with it the pointers
            # of the next iteration's argument records); an integer checksum of the raw bits is an exact comparison
            ts = [out["render"], out["render_gaussian_features"], out["depth"], out["radii"]] + \
                 [p.grad for p in pc.parameters()] + [out["viewspace_points"].grad]
            res.append([int(t.contiguous().view(torch.int32).to(torch.int64).sum()) for t in ts])
            del out, ts
        R.check_overflow()
        stats = R.graph_stats()
    finally:
        R.set_graph("auto")           # the library default
        R.set_sync(True)
    return res, stats


def test_graph_replay_is_bit_identical_and_records_repeat():
    eager, _ = _loop(12, graph=False)
    graphed, stats = _loop(12, graph=True)
    for it, (a, b) in enumerate(zip(eager, graphed)):
        for k, (x, y) in enumerate(zip(a, b)):
            assert x == y, f"iteration {it}, tensor {k}: graph replay differs from the eager launches"
    assert stats["hits"] > 0, f"argument records never repeated: {stats}"
    assert stats["enabled"]


def test_auto_mode_is_the_default_and_replays_small_scenes():
    """Round 5: the library starts in the "auto" mode (trase_rast_graph_mode 2) -- a sync-free loop over a small scene (3000
    Gaussians <= TRASE_GRAPH_AUTO_P) is replayed without the caller asking, bit-identical to the eager launches."""
    from trase_amd import rasterizer as R
    assert R.graph_stats()["enabled"], "the library must start (and be left by the other tests) in a replaying mode"
    eager, _ = _loop(9, graph=False)
    auto, stats = _loop(9, graph="auto")
    assert eager == auto
    assert stats["hits"] > 0, f"auto mode never replayed: {stats}"


def test_graph_mode_is_ignored_by_the_synchronising_policy():
    """set_sync(True) reads the pair count between the two stages: nothing is graphed, nothing breaks."""
    from trase_amd import rasterizer as R
    from tests.util import settings_for, small_case
    from tests import test_gpu_parity as T
    act, cam = small_case(n=500, w=96, h=64, feat=32, seed=1)
    st = settings_for(cam)
    try:
        R.set_graph(True)
        a, _ = T._gpu_call(act, st, need_grad=False)
        b, _ = T._gpu_call(act, st, need_grad=False)
    finally:
        R.set_graph("auto")
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_graph_replay_clears_its_scratch_on_every_hit(monkeypatch):
    """Round 5 regression: the backward clears the pair flags of its gradient-row scratch at its start.  As a hipMemsetAsync
    captured into the launch graph that clear worked for the capture and the first replay only (ROCm 7.0.2): from the second
    hit on, rows of stale flags were summed into a quarter of the Gaussians' gradients.  It is a kernel now
    (launch_zero_bytes).  Poisoned workspaces (0xFF = every flag set, NaN rows) make any missing clear visible."""
    from trase_amd import rasterizer as R
    from tests import test_gpu_overlap as T
    monkeypatch.setattr(R, "_POISON", True)
    pc, pipe, cam, dev = T._scene()
    params = pc.parameters()
    try:
        R.set_graph(False)
        for p in params:
            p.grad = None
        T._backward(pc, pipe, cam, dev)
        base = [p.grad.clone() for p in params]
        R.set_graph("auto")
        for it in range(6):
            for p in params:
                p.grad = None
            T._backward(pc, pipe, cam, dev)
            for k, (p, b) in enumerate(zip(params, base)):
                assert torch.equal(p.grad, b), f"iteration {it}, parameter {k}: graph replay differs from the eager backward"
        assert R.graph_stats()["hits"] >= 4
    finally:
        R.set_graph("auto")
