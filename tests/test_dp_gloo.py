"""N>1 path on CPU: two gloo ranks produce different per-view gradients; the flat bucket's single
all-reduce must equal the sum of the single-view gradients (parity definition of SURVEY.md 8e)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd.dp import FlatGradBucket, allreduce_densify_stats
    torch.manual_seed(0)                      # identical replicas
    params = [torch.randn(50, 3, requires_grad=True), torch.randn(50, 1, 32, requires_grad=True), torch.randn(50, 4, requires_grad=True)]
    bucket = FlatGradBucket(params)
    bucket.zero()
    view_scale = float(rank + 1)              # "a different view per rank"
    loss = sum((p * p).sum() * view_scale for p in params)
    loss.backward()
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in params), ".grad must stay a view of the bucket"
    bucket.allreduce()
    want = [2 * p.detach() * sum(range(1, world + 1)) for p in params]
    ok = all(torch.allclose(p.grad, w) for p, w in zip(params, want))
    acc, den, rad = torch.full((5, 1), float(rank)), torch.ones(5, 1), torch.arange(5.0) * (rank + 1)
    allreduce_densify_stats(acc, den, rad)
    ok = ok and torch.equal(acc, torch.full((5, 1), float(sum(range(world))))) and torch.equal(den, torch.full((5, 1), float(world)))
    ok = ok and torch.equal(rad, torch.arange(5.0) * world)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_bucket_allreduce_equals_sum_of_view_grads():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


class _WritesIntoSink(torch.autograd.Function):
    """Stands in for the fused render backward: returns gradients that are fresh views of caller-provided buffers."""

    @staticmethod
    def forward(ctx, sink, *params):
        ctx.sink = sink
        ctx.ids = [id(p) for p in params]         # the sink is keyed by parameter OBJECT (FlatGradBucket.sink)
        ctx.save_for_backward(*params)
        return sum((p * p).sum() for p in params)

    @staticmethod
    def backward(ctx, g):
        outs = []
        for p, pid in zip(ctx.saved_tensors, ctx.ids):
            ref, buf = ctx.sink[pid]
            assert ref() is not None and id(ref()) == pid
            buf.copy_(2 * p * g)                 # "the kernel writes the gradient once, in place"
            outs.append(buf.view(buf.shape))     # a fresh view object: AccumulateGrad adopts it without a copy
        return (None, *outs)


def _sink_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd.dp import FlatGradBucket
    torch.manual_seed(0)
    params = [torch.randn(40, 3, requires_grad=True), torch.randn(40, 1, 32, requires_grad=True)]
    bucket = FlatGradBucket(params)
    bucket.flat.fill_(float("nan"))               # the sink path must not rely on a zero-filled bucket
    sink = bucket.sink()
    ok = True
    for step in range(2):
        bucket.detach_grads()
        (_WritesIntoSink.apply(sink, *params) * float(rank + 1)).backward()
        ok = ok and bucket.adopted()
        bucket.allreduce()
        want = [2 * p.detach() * sum(range(1, world + 1)) for p in params]
        ok = ok and all(torch.allclose(p.grad, w) for p, w in zip(params, want))
        ok = ok and torch.allclose(bucket.flat, torch.cat([w.reshape(-1) for w in want]))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_grad_sink_bucket_is_adopted_and_reduced_in_place():
    """The N>1 exchange with the gradient sink (trase_amd.renderer.set_grad_sink / FlatGradBucket.sink): gradients written
    straight into the bucket are adopted as .grad without a copy, and the single all-reduce acts on them in place."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sink_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_bucket_owns_every_gradient_before_the_exchange():
    """What the reference loop does between iterations must not leave the all-reduce acting on a stale buffer:
    optimizer.zero_grad(set_to_none=True) (train.py:384-386) detaches the views -> the next backward allocates fresh
    .grad tensors -> gather_grads() copies them in and re-attaches; a parameter without a gradient this step has its
    slice zeroed (not reduced as stale bytes); a changed parameter set (densification) raises."""
    from trase_amd.dp import FlatGradBucket
    torch.manual_seed(1)
    a, b, c = (torch.randn(7, 3, requires_grad=True), torch.randn(7, 1, 32, requires_grad=True), torch.randn(7, 4, requires_grad=True))
    bucket = FlatGradBucket([a, b, c])
    bucket.flat.fill_(123.0)                       # stale contents
    for p in (a, b, c):
        p.grad = None                              # zero_grad(set_to_none=True)
    ((a * a).sum() + (c * 3).sum()).backward()     # b receives no gradient this step
    assert not bucket.adopted()
    fixed = bucket.gather_grads()
    assert fixed == 3
    assert bucket._owns(a.grad) and bucket._owns(c.grad) and b.grad is None
    n_a, n_b = a.numel(), b.numel()
    assert torch.equal(bucket.flat[:n_a].view_as(a), 2 * a.detach())
    assert float(bucket.flat[n_a:n_a + n_b].abs().max()) == 0.0
    assert torch.equal(bucket.flat[n_a + n_b:].view_as(c), torch.full_like(c, 3.0))
    bucket.allreduce()                             # single process: a no-op after the ownership pass
    # a second backward now accumulates in place (views are attached again)
    (a.sum()).backward()
    assert bucket._owns(a.grad) and torch.equal(a.grad, 2 * a.detach() + 1)
    a.data = torch.randn(9, 3)                     # "densification": the parameter set changed
    a.grad = None
    try:
        bucket.gather_grads()
        raise AssertionError("a changed parameter shape must raise")
    except RuntimeError as e:
        assert "new bucket" in str(e)


def test_state_aware_bucket_sizes():
    """236 B / Gaussian in the GAUSSIAN state, 128 B in the FEATURE state, MLP parameters ride along."""
    from types import SimpleNamespace
    from trase_amd.dp import FlatGradBucket
    n = 11
    mk = lambda *s: torch.zeros(n, *s, requires_grad=True)
    pc = SimpleNamespace(_xyz=mk(3), _features_dc=mk(1, 3), _features_rest=mk(15, 3), _opacity=mk(1), _scaling=mk(3),
                         _rotation=mk(4), _gaussian_features=mk(1, 32))
    assert FlatGradBucket.for_state(pc, "GAUSSIAN").bytes_per_step == 236 * n
    assert FlatGradBucket.for_state(pc, "feature").bytes_per_step == 128 * n
    mlp = [torch.zeros(256, 84, requires_grad=True), torch.zeros(256, requires_grad=True)]
    assert FlatGradBucket.for_state(pc, "GAUSSIAN", extra=mlp).bytes_per_step == 236 * n + 4 * (256 * 84 + 256)


class _WritesIntoSinkInRanges(torch.autograd.Function):
    """Stands in for the fused backward in its range-by-range mode (trase_amd.renderer._RenderRaw with
    set_grad_sink(chunks=..., on_chunk=...)): the rows [a, b) of every sink buffer are written, then the hook is called."""

    @staticmethod
    def forward(ctx, kw, scale, *params):
        ctx.kw, ctx.scale = kw, scale
        ctx.save_for_backward(*params)
        return sum((p * p).sum() for p in params) * scale

    @staticmethod
    def backward(ctx, g):
        from trase_amd.renderer import chunk_ranges
        params = ctx.saved_tensors
        sink, hook = ctx.kw["sink"], ctx.kw["on_chunk"]
        P = params[0].shape[0]
        bufs = [sink[id(p)][1] for p in params]
        for a, b in chunk_ranges(P, ctx.kw["chunks"]):
            for p, buf in zip(params, bufs):
                buf[a:b] = 2 * p.detach()[a:b] * ctx.scale * g
            hook(a, b, P, {id(p) for p in params})
        return (None, None) + tuple(buf.view(buf.shape) for buf in bufs)


def _overlap_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd.dp import FlatGradBucket
    torch.manual_seed(0)
    P = 300
    gauss = [torch.randn(P, 3, requires_grad=True), torch.randn(P, 1, 32, requires_grad=True), torch.randn(P, requires_grad=True)]
    mlp = [torch.randn(7, 5, requires_grad=True), torch.randn(P, requires_grad=True)]   # the second LOOKS per-Gaussian but
    bucket = FlatGradBucket(gauss[:2] + [mlp[0]] + gauss[2:] + [mlp[1]])              # never goes through the sink
    kw = bucket.overlapped(3)
    ok = True
    for step in range(2):
        bucket.detach_grads()
        scale = float(rank + 1 + step)
        loss = _WritesIntoSinkInRanges.apply(kw, scale, *gauss) + sum((m * m).sum() for m in mlp) * scale
        loss.backward()
        ok = ok and len(bucket._pending) > 0 and not bucket._owns(mlp[0].grad)
        bucket.allreduce()
        tot = float(sum(r + 1 + step for r in range(world)))
        for p in gauss + mlp:
            ok = ok and torch.allclose(p.grad, 2 * p.detach() * tot) and bucket._owns(p.grad)
        ok = ok and bucket._pending == []
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_overlapped_range_exchange_plus_the_rest_equals_one_allreduce():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _algo_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd.dp import FlatGradBucket
    ok = True
    results = {}
    for algo in ("allreduce", "rs_ag", "direct"):
        torch.manual_seed(0)
        # sizes that do not divide by the world size or by 64: the padded shards must not leak into / lose payload
        params = [torch.randn(37, 3, requires_grad=True), torch.randn(37, 1, 32, requires_grad=True), torch.randn(5, 7, requires_grad=True)]
        bucket = FlatGradBucket(params, exchange=algo)
        bucket.zero()
        # integer-valued gradients: the sums are exact in any order, so the three algorithms must agree BIT FOR BIT
        loss = sum((p.detach().mul(8).round() * p).sum() * float(rank + 1) for p in params)
        loss.backward()
        bucket.allreduce()
        results[algo] = bucket.flat.clone()
        want = torch.cat([(p.detach().mul(8).round() * float(sum(range(1, world + 1)))).reshape(-1) for p in params])
        ok = ok and torch.equal(bucket.flat, want)
        ok = ok and all(bucket._owns(p.grad) for p in params)
        # the padding behind the payload took part in the shards: it must still be zeros
        ok = ok and float(bucket._store[bucket.numel:].abs().max()) == 0.0
    ok = ok and torch.equal(results["allreduce"], results["rs_ag"]) and torch.equal(results["allreduce"], results["direct"])
    # replicas hold identical bytes after the direct exchange (each shard is summed by ONE rank, then distributed)
    mine = results["direct"].clone()
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    ok = ok and all(torch.equal(o, mine) for o in other)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_reduce_scatter_all_gather_and_direct_exchange_equal_the_plain_allreduce():
    """the exchange step is selectable -- one all-reduce, reduce-scatter + all-gather on the padded flat
    bucket, or the same two phases as grouped point-to-point transfers to every peer at once (all seven xGMI links of a
    GPU busy instead of a ring's one).  All three must produce the same sums; world size 3 so that the shards are ragged."""
    world = 3
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_algo_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True, 2: True}


def test_exchange_model_and_chunk_recommendation():
    from trase_amd.dp import exchange_model_ms, recommended_chunks
    b = 300_000 * 364                                   # S4 bucket, every parameter: 109 MB
    ring, direct = exchange_model_ms(b, 8, "ring"), exchange_model_ms(b, 8, "direct")
    assert direct < ring / 3 and exchange_model_ms(b, 1, "ring") == 0.0
    assert 0.3 < direct < 0.7 and 2.0 < ring < 4.0      # ms, from the link figures of SURVEY.md section 5
    assert recommended_chunks(b, 8, "direct") >= 2      # longer than the backward's tail: overlap pays
    assert recommended_chunks(1 << 20, 8, "direct") == 1


def test_inplace_accumulation_into_another_slice_does_not_trip_the_overlap_check():
    """all views of one flat tensor share ONE version counter, so an `extra` parameter (the deformation
    MLP) accumulating in place into ITS bucket slice after the fused backward used to look like a modification of every
    range-exchanged gradient.  The slices now count their own writes: no spurious error; a real in-place write to an
    exchanged slice still raises."""
    from trase_amd.dp import FlatGradBucket
    torch.manual_seed(2)
    P = 128
    gauss = [torch.randn(P, 3, requires_grad=True), torch.randn(P, 1, 32, requires_grad=True)]
    extra = [torch.randn(9, 4, requires_grad=True)]
    bucket = FlatGradBucket(gauss + extra)
    kw = bucket.overlapped(2, force_collectives=False)
    bucket._collectives_on = lambda: True                # single process: pretend the ranges are being exchanged
    bucket._reduce_many = staticmethod(lambda tensors: [])
    bucket._exchange_flat = lambda: None
    import trase_amd.dp as dp_mod
    for p in gauss:
        p.grad = None
    # the extra parameter keeps its attached bucket view: autograd accumulates into it IN PLACE
    extra[0].grad.zero_()
    loss = _WritesIntoSinkInRanges.apply(kw, 1.0, *gauss) + (extra[0] * extra[0]).sum()
    loss.backward()
    (extra[0].sum()).backward()                          # a second in-place accumulation into the extra's slice, after the hand-off
    bucket.allreduce()                                   # must NOT raise
    assert torch.allclose(extra[0].grad, 2 * extra[0].detach() + 1)
    # ... whereas touching an exchanged gradient after its ranges left does raise
    for p in gauss:
        p.grad = None
    loss = _WritesIntoSinkInRanges.apply(kw, 1.0, *gauss)
    loss.backward()
    gauss[0].grad.add_(1.0)
    try:
        bucket.allreduce()
        raise AssertionError("an in-place write to an exchanged gradient must raise")
    except RuntimeError as e:
        assert "modified in place" in str(e)


class _FakeEvent:
    def record(self, *_):
        pass

    def query(self):
        return True

    def synchronize(self):
        pass


def _guard_worker(rank, world, port, out):
    """the MAX-reduced guard header must be refreshed for EVERY forward although the pinned header buffer
    (and hence its id) is the same object for consecutive forwards; and only the flag words are shared."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd import rasterizer as R
    saved = (R._Policy.sync, R._Policy.capacity, R._Policy.pending, R._Policy.rollbacks)
    R._Policy.sync, R._Policy.capacity = False, 1000 + rank
    pin = torch.zeros(32, dtype=torch.int32)                 # ONE pin object recycled for every forward (the LIFO ring)
    seen = []
    # which rank overflows in which iteration: 1 -> rank 1 only; 2 -> nobody; 3 -> rank 0 only; 4 -> rank 1 again
    plan = [(1,), (), (0,), (1,)]
    for it, who in enumerate(plan):
        R._release_last()
        hdr = torch.zeros(64, dtype=torch.int32)
        hdr[1] = int(rank in who)                            # overflow flag
        hdr[2] = 100 * (it + 1) + rank                       # pairs needed
        hdr[62] = 5000 + rank                                # a per-rank word (capacity): must stay local
        geom = hdr.view(torch.uint8).clone()
        R._Policy.pending = [(_FakeEvent(), pin, 1000 + rank)]
        R._Policy.rollbacks = {id(pin): []}
        R._Policy.last_geom = geom
        R._Policy.forward_seq += 1                           # what _after_render does for a sync-free forward
        g1, _ = R.current_guard()
        g2, _ = R.current_guard()                            # FusedAdam.step + add_densification_stats: ONE collective
        assert g1 is g2
        seen.append((int(g1[1]), int(g1[2]), int(g1[62]), int(pin[1])))
    want = [(1, 101, 5000 + rank, 1), (0, 201, 5000 + rank, 0), (1, 301, 5000 + rank, 1), (1, 401, 5000 + rank, 1)]
    out[rank] = seen == want
    R._Policy.sync, R._Policy.capacity, R._Policy.pending, R._Policy.rollbacks = saved
    R._Policy.last_geom = None
    dist.destroy_process_group()


def test_guard_header_is_reduced_for_every_forward_with_a_recycled_pin():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_guard_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _phased_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd.dp import FlatGradBucket
    P = 41                                            # ragged against world = 3 and against 64-float shards

    def make():
        torch.manual_seed(0)                          # identical replicas
        xyz = torch.randn(P, 3, requires_grad=True)
        f_dc = torch.randn(P, 1, 3, requires_grad=True)
        f_rest = torch.randn(P, 15, 3, requires_grad=True)
        opac = torch.randn(P, 1, requires_grad=True)
        mlp = [torch.randn(6, 5, requires_grad=True), torch.randn(6, requires_grad=True)]
        return xyz, f_dc, f_rest, opac, mlp

    def loss_of(params, it, degree):
        # a different "view" per rank and iteration; the SH coefficients above the active degree do not enter (their gradient is
        # identically zero on every rank, as in the rasterizer: train.py:160, scene/gaussian_model.py:219-221)
        xyz, f_dc, f_rest, opac, mlp = params
        k = (degree + 1) ** 2 - 1
        w = 0.37 * (rank + 1) + 0.11 * it
        d = torch.tanh(xyz @ mlp[0][:3, :3] + mlp[1][:3])                   # "the deformation network"
        return (torch.sin(w * (xyz + d)).sum() + torch.cos(w * f_dc).sum() + (torch.sin(w * f_rest[:, :k, :]) * 1.3).sum()
                + torch.sigmoid(opac * w).sum())

    results = {}
    for mode in ("plain", "phased"):
        params = make()
        xyz, f_dc, f_rest, opac, mlp = params
        flat_params = [xyz, f_dc, f_rest, opac] + mlp
        first = [xyz] + mlp
        rest = [f_dc, f_rest, opac]
        bucket = FlatGradBucket(flat_params, exchange="direct")
        opt_first = torch.optim.Adam(first, lr=1e-2, eps=1e-15)
        opt_rest = torch.optim.Adam(rest, lr=1e-2, eps=1e-15)
        sizes = []
        for it in range(3):
            degree = min(it, 3)                       # the ramp: degree 0, 1, 2
            bucket.zero()
            loss_of(params, it, degree).backward()
            if mode == "plain":
                bucket.allreduce(average=True)
                opt_first.step()
                opt_rest.step()
            else:
                ex = bucket.allreduce_phased(first=first, sh_rest=(f_rest, degree), average=True)
                ex.wait_first()
                opt_first.step()
                # (the next iteration's MLP forward would run here, on the updated xyz / MLP parameters)
                _ = torch.tanh(xyz.detach() @ mlp[0].detach()[:3, :3])
                ex.wait_rest()
                opt_rest.step()
                sizes.append((ex.bytes_first, ex.bytes_rest))
        results[mode] = [p.detach().clone() for p in flat_params]
        if mode == "phased":
            # phase A = xyz + MLP; phase B shrinks with the active SH degree: (3 + 1 + 3 k) floats per Gaussian
            ok_sizes = all(a == 4 * (P * 3 + 36) for a, _ in sizes)
            ok_sizes = ok_sizes and [b for _, b in sizes] == [4 * P * (3 + 1 + 3 * k) for k in (0, 3, 8)]
    ok = all(torch.equal(a, b) for a, b in zip(results["plain"], results["phased"]))
    # replicas stay bit-identical
    mine = torch.cat([t.reshape(-1) for t in results["phased"]])
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    ok = ok and all(torch.equal(o, mine) for o in other)
    out[rank] = bool(ok and ok_sizes)
    dist.destroy_process_group()


def test_phased_exchange_is_bit_identical_to_the_plain_exchange_over_three_iterations():
    """the exchange in two phases -- (xyz, MLP) first, the rest (with only the ACTIVE f_rest coefficients
    during the SH ramp) behind the next iteration's first kernels.  World size 3, ragged sizes, the "direct" algorithm: the
    parameters after three Adam iterations must equal, bit for bit, those of the un-overlapped exchange, on every replica."""
    world = 3
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_phased_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True, 2: True}


def _visible_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd.dp import FlatGradBucket
    P = 1000 + 37                                                     # not a multiple of the world size or of 8
    g = torch.Generator().manual_seed(3)
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4), (7, 5)]      # six per-Gaussian tensors + one "MLP" parameter
    base = [torch.randn(s, generator=g) for s in shapes]
    vis_all = [(torch.rand(P, generator=g) < f) for f in (0.6, 0.35, 0.8)][:world]
    vis_all[0][:40] = False; vis_all[1][:40] = False                  # rows nobody (of the first two) sees, a fully dead head for world 2

    def grads_of(r):            # rank r's "view": its visible rows non-zero (and rank-specific), every other row exactly zero
        out_ = []
        for k, b in enumerate(base):
            t = b * float(r + 1) + 0.01 * k
            if b.shape[0] == P:
                t = t * vis_all[r].reshape(P, *([1] * (b.dim() - 1))).to(t.dtype)
            out_.append(t)
        return out_
    results = {}
    for mode in ("dense", "visible"):
        params = [torch.zeros(s, requires_grad=True) for s in shapes]
        bucket = FlatGradBucket(params, exchange="direct")
        for v, t in zip(bucket._views, grads_of(rank)):
            v.copy_(t)
        if mode == "dense":
            bucket.allreduce()
        else:
            st = bucket.allreduce_visible(vis_all[rank])
            results["stats"] = st
        results[mode] = [v.clone() for v in bucket._views]
    ok = all(torch.equal(a, b) for a, b in zip(results["dense"], results["visible"]))
    # and both are the mathematical sum
    want = [sum(gs) for gs in zip(*[grads_of(r) for r in range(world)])]
    ok = ok and all(torch.allclose(a, w, rtol=1e-6, atol=1e-6) for a, w in zip(results["visible"], want))
    st = results["stats"]
    ok = ok and st["rows_sent"] < P and st["bytes_sent"] < st["bytes_dense"]
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_visible_set_exchange_equals_the_dense_rank_ordered_sum_bit_for_bit():
    """each rank contributes only the rows its view can have touched (radii > 0); the owner-ordered sums must
    equal the dense "direct" exchange bit for bit, at world 2 and 3, with rows that no rank sees and a ragged last shard."""
    for world in (2, 3):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_visible_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {r: True for r in range(world)}, (world, dict(out))
