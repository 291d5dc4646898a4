"""The secondary measurement window of SURVEY.md 8(d): one whole training iteration (train.py:157-303, iter_start ->
iter_end, without the optimizer step) in the two optimisation states, with every piece on the HIP path.  Used by
bench.py (the `iteration_ms` key of the bench line) and by profiles/bench_iteration*.py (which also time the reference's
own composition around the same rasterizer operator)."""
from __future__ import annotations

import math

import torch


def _small_deform_net(dev):
    from .synthetic import SynthDeformNetwork
    net = SynthDeformNetwork().to(dev)
    with torch.no_grad():                       # small deformations, as after the warm-up of the reference
        for m in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling):
            m.weight.mul_(0.01)
            m.bias.zero_()
    return net


def _time_input(cam, N, dev):
    """train.py:186-196: `fid = viewpoint_cam.fid` is a DEVICE tensor (scene/cameras.py:58) and
    `time_input = fid.unsqueeze(0).expand(N, -1)` a stride-0 view of it -- no host round trip.  (Reading the value back to
    build a new tensor, as this harness did before round 5, synchronised host and device once per iteration: 0.16 ms of idle
    GPU in front of every iteration's first kernel, profiles/r5_iteration_timeline.md.)"""
    fid = getattr(cam, "fid", None)
    if not isinstance(fid, torch.Tensor):
        fid = torch.tensor([0.3 if fid is None else float(fid)], device=dev)
    elif fid.device != dev:
        fid = fid.to(dev)
    return fid.reshape(1, 1).expand(N, -1)


def make_gaussian_iteration(pc, cams, W: int, H: int, dev, image_scope: bool = True):
    """GAUSSIAN state (train.py:196-243, :299): deformation MLP with gradients -> render() -> L1 + SSIM -> backward.
    image_scope: the state never reads the feature map (train.py:211), so the forward composites colour + depth only
    (trase_amd.renderer.set_forward_scope("image")) and the backward takes the image-only MFMA scope."""
    from .deform import DeformNetworkHIP
    from .losses import photometric_loss
    from .renderer import render, set_forward_scope
    from .synthetic import SynthPipe
    N = pc.get_xyz.shape[0]
    net = _small_deform_net(dev)
    hip_net = DeformNetworkHIP(net)
    params = pc.parameters() + list(net.parameters())
    bg = torch.zeros(3, device=dev)
    g = torch.Generator().manual_seed(7)
    gts = [torch.rand(3, H, W, generator=g).to(dev) for _ in range(2)]
    pipe = SynthPipe()

    def it(i):
        for p in params:
            p.grad = None
        cam = cams[i % len(cams)]
        t = _time_input(cam, N, dev)
        d_xyz, d_rot, d_scale = hip_net(pc.get_xyz.detach(), t)
        set_forward_scope("image" if image_scope else "all")
        try:
            out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
        finally:
            set_forward_scope("all")
        photometric_loss(out["render"], gts[i % 2], 0.2).backward()      # train.py:235-238, lambda_dssim = 0.2 (arguments/__init__.py)
        return out
    return it


def make_feature_iteration(pc, cams, W: int, H: int, dev, n_masks: int = 100):
    """FEATURE state (train.py:189-299 after the warm-up): MLP under no_grad -> render(normalised, KNN-smoothed features,
    smooth_K = 16) -> regulariser + sampling + pair losses + similarities on `n_masks` masks, 5000 sampled pixels ->
    backward.  Returns (callable, restore): the state trains the features only (scene/gaussian_model.py:303-315);
    restore() puts requires_grad back."""
    from .deform import DeformNetworkHIP
    from .feature_head import contrastive_head, get_sample_pixel_and_mask, mask_stats
    from .renderer import render
    from .synthetic import SynthPipe
    N = pc.get_xyz.shape[0]
    before = [(p, p.requires_grad) for p in pc.parameters()]
    for p in pc.parameters():
        p.requires_grad_(p is pc._gaussian_features)
    hip_net = DeformNetworkHIP(_small_deform_net(dev))
    bg = torch.zeros(3, device=dev)
    g = torch.Generator().manual_seed(0)
    sam = torch.zeros(n_masks, H, W, dtype=torch.bool, device=dev)
    for n in range(n_masks):
        y0, x0 = int(torch.randint(0, max(H - 50, 1), (1,), generator=g)), int(torch.randint(0, max(W - 50, 1), (1,), generator=g))
        h, w = int(torch.randint(40, 500, (1,), generator=g)), int(torch.randint(40, 700, (1,), generator=g))
        sam[n, y0:y0 + h, x0:x0 + w] = True
    pipe = SynthPipe()

    def it(i):
        pc._gaussian_features.grad = None
        cam = cams[i % len(cams)]
        with torch.no_grad():
            t = _time_input(cam, N, dev)
            d_xyz, d_rot, d_scale = hip_net(pc.get_xyz.detach(), t)
        out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale, norm_gaussian_features=True, is_smooth_gaussian_features=True, smooth_K=16)
        cover, size = mask_stats(sam)
        sp, sm = get_sample_pixel_and_mask(sam, 5000, 50, cover_count=cover, rng="cuda")
        lp, ln, ps, ns, reg = contrastive_head(out["render_gaussian_features"], sam, sp, sm, "soft", 0.75, 0.5, mask_size=size,
                                               with_norm_reg=True)
        (lp + ln + 1.0 * reg).backward()
        return out

    def restore():
        for p, rg in before:
            p.requires_grad_(rg)
    return it, restore


def time_iterations(fn, iters: int = 8, warm: int = 3) -> float:
    """ms per iteration (host clock around `iters` calls, device synchronised on both sides)."""
    import time
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def capture_iteration(fn, warm: int = 3, arg: int = 0):
    """One whole iteration (``fn(arg)``: forward, loss, backward -- every launch of it) captured into a ``torch.cuda.CUDAGraph``.
    Returns ``(graph, out)``: ``graph.replay()`` re-runs the iteration on the buffers of the capture (``out`` = what ``fn``
    returned, plus whatever ``.grad`` tensors the backward made: static tensors, refreshed by every replay).

    What a capture freezes: every HOST-side decision of the iteration -- the pair capacity (use ``rasterizer.set_sync(False,
    capacity=...)`` with headroom: an overflow is still flagged in the geom header and honoured by the guarded device-side
    consumers, but the host never hears of it), the camera, the smoothing's neighbour-slot draw (a CPU ``randperm``).  Device-side
    draws (``get_sample_pixel_and_mask(rng="cuda")``) advance with every replay (torch registers its CUDA generator with the
    graph).  The library's own launch-graph cache steps aside while the caller captures (``run_maybe_graphed``).  Warm-up and
    capture share one side stream, as torch's CUDA-graph recipe asks."""
    import gc
    dev = torch.cuda.current_device()
    # Autograd graphs of EARLIER iterations must be gone: a parameter's AccumulateGrad node lives as long as a graph references it
    # and keeps the stream it was created on -- the default stream, for iterations that ran there -- and the backward would then
    # synchronise the capture with that stream, which invalidates it (torch warns "AccumulateGrad node's stream does not match").
    # The caller drops its own references (outputs with a grad_fn); cyclic leftovers are collected here.
    gc.collect()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for _ in range(warm):
            out = fn(arg)
            del out
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = fn(arg)
    return g, out


def time_graph_replays(graph, iters: int = 16, warm: int = 8) -> float:
    import time
    for _ in range(warm):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
