"""The backward in two phases (trase_rast_backward_raw_compose + _gaussians over Gaussian-index ranges) and the overlapped
gradient exchange built on it (trase_amd.dp.FlatGradBucket.overlapped; SURVEY.md 8e, the reference is single-process).
On the one-GPU box: the ranges reproduce the one-call backward bit for bit, and the exchange runs through a one-rank RCCL
process group (the collectives are issued for real; summing over one rank must leave the gradients untouched)."""
import math
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n=20000, seed=3):
    from trase_amd.synthetic import make_scene, SynthGaussianModel, SynthPipe, orbit_camera
    dev = torch.device("cuda", 0)
    pc = SynthGaussianModel(make_scene(n, feat_dim=32, seed=seed).to(dev))
    cam = orbit_camera(320, 200, angle=0.4, fid=0.25).to(dev)
    return pc, SynthPipe(), cam, dev


def _backward(pc, pipe, cam, dev, seed=11):
    from gaussian_renderer import render
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = render(cam, pc, pipe, torch.zeros(3, device=dev), 0.0, 0.0, 0.0)
    gi = torch.randn(out["render"].shape, generator=g).to(dev)
    gf = torch.randn(out["render_gaussian_features"].shape, generator=g).to(dev)
    torch.autograd.backward([out["render"], out["render_gaussian_features"]], [gi, gf])


def test_chunk_ranges_partition():
    from trase_amd.renderer import chunk_ranges
    for P in (1, 63, 64, 65, 1000, 300000):
        for k in (1, 2, 4, 7, 100000):
            r = chunk_ranges(P, k)
            assert r[0][0] == 0 and r[-1][1] == P and len(r) <= k
            assert all(a % 64 == 0 for a, _ in r)
            assert all(r[i][1] == r[i + 1][0] for i in range(len(r) - 1))
            assert all(b > a for a, b in r)


@pytest.mark.parametrize("chunks", [2, 4, 7])
def test_ranged_tail_equals_the_single_call_bit_for_bit(chunks):
    from trase_amd.dp import FlatGradBucket
    from trase_amd.renderer import set_grad_sink
    pc, pipe, cam, dev = _scene()
    params = pc.parameters()
    for p in params:
        p.grad = None
    _backward(pc, pipe, cam, dev)
    ref = [p.grad.clone() for p in params]
    bucket = FlatGradBucket(params)
    calls = []
    kw = bucket.overlapped(chunks)
    hook = kw["on_chunk"]
    kw["on_chunk"] = lambda a, b, P, ids: (calls.append((a, b, P, set(ids))), hook(a, b, P, ids))
    try:
        set_grad_sink(**kw)
        bucket.detach_grads()
        _backward(pc, pipe, cam, dev)
        bucket.allreduce()
    finally:
        set_grad_sink(None)
    P = params[0].shape[0]
    assert [c[:2] for c in calls] == [tuple(x) for x in __import__("trase_amd.renderer", fromlist=["x"]).chunk_ranges(P, chunks)]
    assert all(c[3] == {id(p) for p in params} for c in calls)          # every parameter's buffer came from the sink
    assert bucket.adopted()
    for p, r in zip(params, ref):
        assert torch.equal(p.grad, r)


@pytest.mark.parametrize("chunks", [1, 3])
def test_sink_with_grads_still_attached_accumulates_instead_of_doubling(chunks):
    """A FlatGradBucket hands its slices out as ``.grad`` when it is built.  With the sink set and those ``.grad`` still in place
    (the accumulate-mode loop: ``bucket.zero()`` instead of ``p.grad = None``) the backward used to return the SAME bytes as the
    incoming gradient, and AccumulateGrad added the buffer to itself: every gradient doubled, silently.  Now the backward notices
    that ``.grad`` is the sink buffer and returns a fresh tensor: the bucket ends up with the plain gradients (and with their sum
    over two backwards)."""
    from trase_amd.dp import FlatGradBucket
    from trase_amd.renderer import set_grad_sink
    pc, pipe, cam, dev = _scene()
    params = pc.parameters()
    for p in params:
        p.grad = None
    _backward(pc, pipe, cam, dev)
    ref = [p.grad.clone() for p in params]
    bucket = FlatGradBucket(params)              # p.grad = the bucket's slices, zero
    try:
        set_grad_sink(**bucket.overlapped(chunks)) if chunks > 1 else set_grad_sink(bucket.sink())
        _backward(pc, pipe, cam, dev)
        assert bucket.adopted()
        for p, r in zip(params, ref):
            assert torch.equal(p.grad, r), "a gradient was not accumulated once"
        _backward(pc, pipe, cam, dev)            # no zero in between: the sum of two backwards
        for p, r in zip(params, ref):
            assert torch.equal(p.grad, r + r)
    finally:
        set_grad_sink(None)


def test_overlapped_exchange_through_a_one_rank_rccl_group():
    import torch.distributed as dist
    from trase_amd.dp import FlatGradBucket
    from trase_amd.renderer import set_grad_sink
    pc, pipe, cam, dev = _scene(n=30000, seed=5)
    params = pc.parameters()
    for p in params:
        p.grad = None
    _backward(pc, pipe, cam, dev)
    ref = [p.grad.clone() for p in params]
    extra = torch.nn.Parameter(torch.randn(1000, device=dev))          # stands in for the deformation MLP: never chunked
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        bucket = FlatGradBucket(list(params) + [extra])
        set_grad_sink(**bucket.overlapped(4, force_collectives=True))
        for _ in range(2):                                              # twice: the per-step state is reset properly
            bucket.detach_grads()
            _backward(pc, pipe, cam, dev)
            extra.grad = torch.full_like(extra, 2.0)                     # a gradient that arrives outside the sink
            assert len(bucket._pending) >= 1                             # the ranges' collectives are in flight
            bucket.allreduce()
            torch.cuda.synchronize()
            for p, r in zip(params, ref):
                assert torch.equal(p.grad, r)
            assert torch.equal(extra.grad, torch.full_like(extra, 2.0)) and bucket.adopted()
    finally:
        set_grad_sink(None)
        dist.destroy_process_group()


def test_phased_exchange_through_a_one_rank_rccl_group_with_the_mlp_underneath():
    """VERDICT r4 item 4 on the GPU box: FlatGradBucket.allreduce_phased through a one-rank RCCL group -- phase A (xyz + the
    deformation network), FusedAdam.step(only=...), the NEXT iteration's MLP training forward issued while phase B (the rest,
    only the active f_rest rows) is still on the side stream, wait_rest, the second optimizer half.  Summing over one rank
    must leave every gradient untouched, so parameters and MLP outputs after two iterations must equal, bit for bit, those of
    the plain loop (one allreduce, one whole optimizer step) -- with RCCL kernels and staging copies running beside
    mlp_fwd_train_kernel_blk (profiles/r5_two_streams.md is what allows that schedule)."""
    import torch.distributed as dist
    from trase_amd.deform import DeformNetworkHIP
    from trase_amd.dp import FlatGradBucket
    from trase_amd.optim import FusedAdam
    from trase_amd.synthetic import SynthDeformNetwork
    from gaussian_renderer import render
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    results = {}
    try:
        for mode in ("plain", "phased"):
            pc, pipe, cam, _ = _scene(n=30000, seed=5)
            torch.manual_seed(11)
            net = SynthDeformNetwork().to(dev)
            with torch.no_grad():
                for m in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling):
                    m.weight.mul_(0.01)
            hip_net = DeformNetworkHIP(net)
            gauss = [pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation]
            mlp = list(net.parameters())
            bucket = FlatGradBucket(gauss + mlp, exchange="direct")
            bucket._force = True
            opt = FusedAdam([{"params": [p], "lr": 1e-3} for p in gauss] + [{"params": mlp, "lr": 1e-3}], lr=0.0, eps=1e-15)
            first = [pc._xyz] + mlp
            rest = [p for p in gauss if p is not pc._xyz]
            N = pc._xyz.shape[0]
            t = cam.fid.reshape(1, 1).expand(N, -1)
            g = torch.Generator(device="cpu").manual_seed(3)
            d = hip_net(pc.get_xyz.detach(), t)
            for it in range(2):
                bucket.zero()
                out = render(cam, pc, pipe, torch.zeros(3, device=dev), *d)
                gi = torch.randn(out["render"].shape, generator=g).to(dev)
                out["render"].backward(gi)
                if mode == "plain":
                    bucket.allreduce()
                    opt.step(guard=None)
                    d = hip_net(pc.get_xyz.detach(), t)
                else:
                    ex = bucket.allreduce_phased(first=first, sh_rest=(pc._features_rest, 1))
                    ex.wait_first()
                    opt.step(guard=None, only=first)
                    d = hip_net(pc.get_xyz.detach(), t)            # the next iteration's MLP forward, underneath phase B
                    ex.wait_rest()
                    opt.step(guard=None, only=rest)
                    assert ex.bytes_first == 4 * (N * 3 + sum(p.numel() for p in mlp))
                    assert ex.bytes_rest == 4 * N * (3 + 9 + 1 + 3 + 4)     # f_dc, 3 active f_rest rows, opacity, scaling, rotation
            torch.cuda.synchronize()
            results[mode] = [p.detach().clone() for p in gauss + mlp] + [x.detach().clone() for x in d]
    finally:
        dist.destroy_process_group()
    for i, (a, b) in enumerate(zip(results["plain"], results["phased"])):
        assert torch.equal(a, b), f"tensor {i}: the phased loop differs from the plain one"


@pytest.mark.parametrize("n", [777, 4096 + 13])
def test_unaligned_parameter_storage_gives_identical_results(n):
    """The wave-cooperative row moves of the per-Gaussian kernels use 16-byte accesses when the base pointers allow it and
    fall back otherwise: parameters that are views at a 4-byte offset into a larger buffer (P not a multiple of 64) must
    give bit-identical maps and gradients."""
    from trase_amd.synthetic import make_scene, SynthGaussianModel, SynthPipe, orbit_camera
    dev = torch.device("cuda", 0)
    cam = orbit_camera(200, 136, angle=1.1, fid=0.5).to(dev)
    pipe = SynthPipe()
    results = []
    for shifted in (False, True):
        pc = SynthGaussianModel(make_scene(n, feat_dim=32, seed=9).to(dev))
        if shifted:
            for name in ("_features_rest", "_gaussian_features", "_features_dc", "_xyz", "_scaling", "_rotation", "_opacity"):
                p = getattr(pc, name)
                buf = torch.empty(p.numel() + 1, device=dev)
                view = buf[1:].view(p.shape)
                view.copy_(p.detach())
                assert view.data_ptr() % 16 == 4 and view.is_contiguous()
                setattr(pc, name, torch.nn.Parameter(view))
        for p in pc.parameters():
            p.grad = None
        from gaussian_renderer import render
        g = torch.Generator(device="cpu").manual_seed(2)
        out = render(cam, pc, pipe, torch.zeros(3, device=dev), 0.0, 0.0, 0.0)
        gi = torch.randn(out["render"].shape, generator=g).to(dev)
        gf = torch.randn(out["render_gaussian_features"].shape, generator=g).to(dev)
        torch.autograd.backward([out["render"], out["render_gaussian_features"]], [gi, gf])
        results.append(([out["render"].detach().clone(), out["render_gaussian_features"].detach().clone(),
                         out["radii"].clone()], [p.grad.clone() for p in pc.parameters()]))
    (m0, g0), (m1, g1) = results
    for a, b in zip(m0 + g0, m1 + g1):
        assert torch.equal(a, b)


def test_forward_scope_image_matches_the_full_render():
    """set_forward_scope("image"): same image / depth / radii (the colour-only kernels blend in a different instruction
    order: 2e-5) and the same gradients as an image-only cotangent through the full render."""
    from gaussian_renderer import render
    from trase_amd.renderer import set_forward_scope
    pc, pipe, cam, dev = _scene(n=15000, seed=21)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    outs = []
    for scope in ("all", "image"):
        set_forward_scope(scope)
        try:
            for p in pc.parameters():
                p.grad = None
            out = render(cam, pc, pipe, bg, 0.0, 0.0, 0.0)
            if scope == "all":
                gi = torch.randn(out["render"].shape, generator=g).to(dev)
            out["render"].backward(gi)
            outs.append((out, [None if p.grad is None else p.grad.clone() for p in pc.parameters()]))
        finally:
            set_forward_scope("all")
    (a, ga), (b, gb) = outs
    assert b["render_gaussian_features"].shape[0] == 0 and a["render_gaussian_features"].shape[0] == 32
    assert torch.equal(a["radii"], b["radii"])
    assert (a["render"] - b["render"]).abs().max() < 2e-5 and (a["depth"] - b["depth"]).abs().max() < 2e-4
    for x, y in zip(ga, gb):
        if x is None or y is None:
            assert (x is None or float(x.abs().max()) == 0.0) and (y is None or float(y.abs().max()) == 0.0)
            continue
        assert (x - y).abs().max() <= 2e-4 * max(float(x.abs().max()), 1e-12)
