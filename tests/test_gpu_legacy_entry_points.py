"""The first-generation entry points of include/trase_rast.h that the Python layer no longer calls -- trase_adam_step,
trase_densify_stats, trase_mlp_forward_train, trase_mlp_backward, trase_pairhead_forward, trase_pairhead_backward -- stay exported
for C-ABI callers as the "no guard / no row order / host-side count" cases of their successors.  Every exported symbol has to be
exercised on the GPU: here the wrappers are pointed at the old symbols (a shim around the loaded library that drops the
successor's extra argument, which must be NULL on these paths) and their results are compared bit for bit with the normal path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class _LegacyShim:
    """the loaded library, with the successors routed to the entry points they superseded"""

    def __init__(self, real):
        self._real = real
        self.calls = {}

    def __getattr__(self, name):
        return getattr(self._real, name)

    def _route(self, legacy, args, drop):
        assert args[drop] is None, f"{legacy}: the successor's extra argument must be NULL on this path, got {args[drop]!r}"
        self.calls[legacy] = self.calls.get(legacy, 0) + 1
        return getattr(self._real, legacy)(*(args[:drop] + args[drop + 1:]))

    def trase_mlp_forward_train_rows(self, *a):
        return self._route("trase_mlp_forward_train", a, 5)

    def trase_mlp_backward_rows(self, *a):
        return self._route("trase_mlp_backward", a, 2)

    def trase_pairhead_forward_n(self, *a):
        return self._route("trase_pairhead_forward", a, 10)

    def trase_pairhead_backward_n(self, *a):
        return self._route("trase_pairhead_backward", a, 4)

    def trase_adam_step_guarded(self, *a):
        return self._route("trase_adam_step", a, 11)

    def trase_densify_stats_guarded(self, *a):
        return self._route("trase_densify_stats", a, 6)


@pytest.fixture
def legacy(monkeypatch):
    from trase_amd import _lib
    shim = _LegacyShim(_lib.load())
    return shim, (lambda: monkeypatch.setattr(_lib, "load", lambda: shim)), (lambda: monkeypatch.undo())


def test_mlp_training_pair_through_the_first_entry_points(legacy):
    from trase_amd import deform as D
    from trase_amd.deform import deform_forward
    from trase_amd.synthetic import SynthDeformNetwork
    shim, on, off = legacy
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    n = 5000
    net = SynthDeformNetwork().to(dev)
    x = (torch.rand(n, 3, device=dev) * 2 - 1) * 1.3
    t = torch.tensor([[0.4]], device=dev).expand(n, -1)
    live = dict(net.named_parameters())
    g = [torch.randn(n, c, device=dev) for c in (3, 4, 3)]

    def step():
        net.zero_grad(set_to_none=True)
        outs = deform_forward(live, x, t)
        torch.autograd.backward(outs, g)
        return [o.detach().clone() for o in outs], {k: p.grad.clone() for k, p in live.items()}

    D.set_row_order("none")            # (index order: the row-order argument of the successors is then NULL)
    try:
        o_new, g_new = step()
        on()
        o_old, g_old = step()
        off()
    finally:
        D.set_row_order("morton")
    assert shim.calls.get("trase_mlp_forward_train") == 1 and shim.calls.get("trase_mlp_backward") == 1
    for a, b in zip(o_new, o_old):
        assert torch.equal(a, b)
    for k in g_new:
        assert torch.equal(g_new[k], g_old[k]), k


def test_pair_head_through_the_first_entry_points(legacy):
    from trase_amd.feature_head import contrastive_head, mask_stats
    shim, on, off = legacy
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4)
    H, W, N = 96, 128, 12
    sam = torch.zeros(N, H, W, dtype=torch.bool, device=dev)
    for k in range(N):
        y0, x0 = int(torch.randint(0, H - 20, (1,), generator=g)), int(torch.randint(0, W - 20, (1,), generator=g))
        sam[k, y0:y0 + 30, x0:x0 + 40] = True
    cover, size = mask_stats(sam)
    sp = torch.logical_and(torch.rand(H, W, generator=g).to(dev) < 0.05, cover != 0)      # no expected-count tag: host-side count
    sm = torch.ones(N, dtype=torch.bool, device=dev)
    feats0 = torch.randn(32, H, W, generator=g).to(dev)

    def run():
        f = feats0.clone().requires_grad_(True)
        lp, ln, ps, ns = contrastive_head(f, sam, sp, sm, "soft", 0.75, 0.5, mask_size=size)
        (lp + ln).backward()
        return [lp.detach().clone(), ln.detach().clone(), ps.clone(), ns.clone(), f.grad.clone()]

    new = run()
    on()
    old = run()
    off()
    assert shim.calls.get("trase_pairhead_forward") == 1 and shim.calls.get("trase_pairhead_backward") == 1
    for i, (a, b) in enumerate(zip(new, old)):
        assert torch.equal(a, b) or (torch.isnan(a).all() and torch.isnan(b).all()), i


def test_adam_and_densification_statistics_through_the_first_entry_points(legacy):
    from types import SimpleNamespace
    from trase_amd.densify import add_densification_stats
    from trase_amd.optim import FusedAdam
    shim, on, off = legacy
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    P = 4000

    def run():
        torch.manual_seed(7)
        ps = [torch.randn(P, 3, device=dev, requires_grad=True), torch.randn(P, 1, device=dev, requires_grad=True)]
        opt = FusedAdam([{"params": [p], "lr": 1e-2} for p in ps], eps=1e-15)
        for it in range(3):
            for p in ps:
                p.grad = torch.randn_like(p)
            opt.step(guard=None)
        stats = SimpleNamespace(xyz_gradient_accum=torch.zeros(P, 1, device=dev), denom=torch.zeros(P, 1, device=dev),
                                max_radii2D=torch.zeros(P, device=dev))
        vp = torch.zeros(P, 3, device=dev, requires_grad=True)
        vp.grad = torch.randn(P, 3, device=dev)
        radii = torch.randint(0, 40, (P,), device=dev, dtype=torch.int32)
        add_densification_stats(stats, vp, radii)
        return [p.detach().clone() for p in ps] + [stats.xyz_gradient_accum.clone(), stats.denom.clone(), stats.max_radii2D.clone()]

    new = run()
    on()
    old = run()
    off()
    assert shim.calls.get("trase_adam_step", 0) >= 3 and shim.calls.get("trase_densify_stats") == 1
    for i, (a, b) in enumerate(zip(new, old)):
        assert torch.equal(a, b), i
