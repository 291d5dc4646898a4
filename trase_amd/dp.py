"""Data parallelism over camera views: one process per GPU renders a different
view; the only exchange step is ONE all-reduce of a flat gradient bucket (RCCL over xGMI on the
GPU box, gloo in the CPU tests).  The reference itself is single-process.

The bucket holds exactly the parameters that receive a gradient in the current optimisation state
(scene/gaussian_model.py:303-315): 236 B / Gaussian in the GAUSSIAN state (xyz, f_dc, f_rest,
opacity, scaling, rotation), 128 B / Gaussian in the FEATURE state (gaussian features), plus --
when given -- the deformation network's parameters (2 MB), so that the step still ends with one
collective."""
from __future__ import annotations

import weakref
from typing import Iterable, List, Optional

import torch

GAUSSIAN_STATE_ATTRS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
FEATURE_STATE_ATTRS = ("_gaussian_features",)

# ---- the exchange step as a design ------------------------------------------------------------------------------------
# MI355X: every GPU of a node has ONE xGMI link to each of the other seven (7 links x ~153.6 GB/s
# bidirectional, i.e. ~76.8 GB/s per direction per link).  A ring all-reduce moves 2 (W-1)/W of the bucket over ONE link
# per GPU at a time; the direct algorithms (every rank reduces its own 1/W shard from W-1 peers at once, then hands the
# shard to W-1 peers at once) move (W-1)/W of the bucket per phase spread over W-1 links.
XGMI_LINK_GBS_PER_DIR = 76.8
XGMI_LINK_EFFICIENCY = 0.8          # achievable fraction of the link rate for MB-sized messages (assumption, unmeasured here)
COLLECTIVE_LATENCY_MS = 0.02        # launch + handshake of one collective (assumption)
EXCHANGE_ALGOS = ("allreduce", "rs_ag", "direct")


def exchange_model_ms(nbytes: int, world: int, algo: str = "direct") -> float:
    """Predicted duration of one gradient exchange of `nbytes` over `world` GPUs of one xGMI node (see the constants above;
    UNMEASURED on multi-GPU hardware -- profiles/r4_scaling_model.json tabulates it).  "ring": what a ring all-reduce or a
    ring reduce-scatter + ring all-gather costs; "direct": both phases over all W-1 links at once."""
    if world <= 1:
        return 0.0
    link = XGMI_LINK_GBS_PER_DIR * XGMI_LINK_EFFICIENCY * 1e9
    shard = nbytes / world
    if algo in ("ring", "allreduce", "rs_ag"):
        # 2 (W-1) steps, each moving one shard over one link per GPU
        return 2 * (world - 1) * (shard / link * 1e3 + COLLECTIVE_LATENCY_MS / (world - 1)) 
    if algo == "direct":
        # 2 phases, each moving one shard over each of the W-1 links concurrently
        return 2 * (shard / link * 1e3 + COLLECTIVE_LATENCY_MS)
    raise ValueError(algo)


def recommended_chunks(nbytes: int, world: int, algo: str, tail_ms: float = 0.18, chunk_cost_ms: float = 0.027, max_chunks: int = 4) -> int:
    """How many Gaussian-index ranges the overlapped exchange should use: ranges only pay where the modelled exchange is
    longer than what they cost (every extra range adds ~0.027 ms of kernel ramp / tail at S4, measured at N = 1) and can
    hide at most (K-1)/K of the backward's per-Gaussian tail (reduce_rows + preprocess_bwd, ~0.18 ms at S4)."""
    t = exchange_model_ms(nbytes, world, algo)
    best, best_gain = 1, 0.0
    for k in range(2, max_chunks + 1):
        gain = min(t, tail_ms) * (k - 1) / k - chunk_cost_ms * (k - 1)
        if gain > best_gain:
            best, best_gain = k, gain
    return best


class FlatGradBucket:
    """Makes ``.grad`` of every parameter a view into one contiguous buffer, so that the step ends
    with a single collective of sum(numel) floats.

    Two ways to fill it:
    * accumulate: ``zero()`` before the backward; autograd accumulates into the views in place;
    * sink: ``trase_amd.renderer.set_grad_sink(bucket.sink())`` + ``detach_grads()`` before the backward; the fused
      backward writes each gradient once, straight into the bucket, and autograd adopts the views as ``.grad``.

    ``allreduce()`` first makes the bucket own every gradient: a parameter whose ``.grad`` was replaced by a fresh tensor
    (``optimizer.zero_grad(set_to_none=True)`` of the reference loop, train.py:384-386, followed by a backward outside the
    sink; a KNN-smoothed feature tensor whose gradient reaches the leaf through autograd) is copied in and re-attached; one
    that received NO gradient has its slice zeroed instead of reducing stale bytes.  When the parameter set itself changed
    (densify / prune replaces every nn.Parameter) it raises: build a new bucket (``for_state``)."""

    PAD = 64 * 64       # slack behind the payload: shards of any world size <= 64 can be rounded up to 64 floats

    def __init__(self, params: Iterable[torch.Tensor], exchange: str = "allreduce"):
        self.params: List[torch.Tensor] = list(params)
        assert self.params, "no parameters"
        if exchange not in EXCHANGE_ALGOS:
            raise ValueError(f"exchange must be one of {EXCHANGE_ALGOS}")
        self.exchange = exchange
        self.time_exchange = False          # record HIP events around the collective(s) of allreduce()
        self.last_exchange_ms: Optional[float] = None
        self._ev = None
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self._store = torch.zeros(self.numel + self.PAD, device=dev, dtype=dt)
        self.flat = self._store[:self.numel]
        self._shapes = [tuple(p.shape) for p in self.params]
        self._refs = [weakref.ref(p) for p in self.params]
        self._views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            # NOT a view of self.flat: a tensor of its own on the same storage.  Views of one base share ONE version counter
            # (an in-place write through any of them -- AccumulateGrad into another parameter's slice, flat.zero_() -- bumps
            # all), which made the overlapped exchange's modified-after-hand-off check fire spuriously; tensors made with
            # set_() alias the bytes but count their own in-place writes.
            v = torch.empty(0, device=dev, dtype=dt).set_(self._store.untyped_storage(), off, tuple(p.shape))
            self._views.append(v)
            p.grad = v
            off += p.numel()

    @classmethod
    def for_state(cls, pc, state: str, extra: Optional[Iterable[torch.Tensor]] = None, exchange: str = "allreduce") -> "FlatGradBucket":
        """Bucket over what the given optimisation state trains (``"GAUSSIAN"`` / ``"FEATURE"``,
        scene/gaussian_model.py:303-315) plus ``extra`` (e.g. the deformation MLP's parameters)."""
        attrs = {"GAUSSIAN": GAUSSIAN_STATE_ATTRS, "FEATURE": FEATURE_STATE_ATTRS}[state.upper()]
        return cls([getattr(pc, a) for a in attrs] + list(extra or []), exchange=exchange)

    @property
    def bytes_per_step(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero(self):
        for v in self._views:      # through the parameters' own tensors: their version counters see it
            v.zero_()

    def sink(self):
        """{id(param): (weakref(param), view of its slice)} for ``trase_amd.renderer.set_grad_sink``.  Keyed by the
        parameter OBJECT (checked through the weak reference), not by its address: after densification a new tensor
        may reuse a freed allocation."""
        return {id(p): (r, v) for p, r, v in zip(self.params, self._refs, self._views)}

    def detach_grads(self):
        for p in self.params:
            p.grad = None

    def _owns(self, t: Optional[torch.Tensor]) -> bool:
        if t is None:
            return False
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * self.flat.element_size()
        return lo <= t.data_ptr() < hi

    def adopted(self) -> bool:
        """True when every parameter's ``.grad`` lives inside the bucket."""
        return all(self._owns(p.grad) for p in self.params)

    def gather_grads(self) -> int:
        """Make the bucket own every gradient (see the class docstring).  Returns how many slices had to be fixed."""
        fixed = 0
        for p, shape, v in zip(self.params, self._shapes, self._views):
            if tuple(p.shape) != shape:
                raise RuntimeError("FlatGradBucket: a parameter changed shape (densify / prune replaced the parameters) -- "
                                   "build a new bucket for the new parameter set")
            g = p.grad
            if g is None:
                v.zero_()                          # no gradient this step: do not reduce whatever the slice held before
                fixed += 1
            elif not (self._owns(g) and g.data_ptr() == v.data_ptr()):
                v.copy_(g)
                p.grad = v
                fixed += 1
        return fixed

    # ---- overlapped exchange -------------------------------------------------------------------------------------------
    def overlapped(self, chunks: int = 4, force_collectives: bool = False):
        """Keyword arguments for ``trase_amd.renderer.set_grad_sink``: the sink plus a per-range hook.  The fused backward
        then finishes the gradients of one Gaussian-index range after the other (``trase_rast_backward_raw_gaussians``)
        and this bucket starts the all-reduce of a range -- the matching rows of every per-Gaussian tensor that was
        written through the sink, one coalesced collective per range -- as soon as the range is in the stream: the
        exchange of the first ranges runs (on the process group's own stream) while the tail of the backward is still
        computing the last ones.  ``allreduce()`` afterwards waits for the ranges and reduces only what was NOT exchanged
        that way (e.g. the deformation MLP's parameters).  One fused backward per ``allreduce()``.
        ``force_collectives``: issue the collectives even in a world of one rank (tests)."""
        self._force = bool(force_collectives)
        self._pending: list = []
        self._exchanged: set = set()
        return dict(sink=self.sink(), chunks=int(chunks), on_chunk=self._on_chunk)

    def _collectives_on(self) -> bool:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or getattr(self, "_force", False)

    @staticmethod
    def _reduce_many(tensors):
        """One grouped all-reduce (SUM) of several tensors, asynchronous; returns the handles to wait on."""
        import torch.distributed as dist
        if not tensors:
            return []
        if dist.get_backend() == "nccl" and hasattr(dist, "_coalescing_manager"):
            with dist._coalescing_manager(device=tensors[0].device, async_ops=True) as cm:      # one ncclGroup
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return [cm]
        return [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in tensors]

    def _on_chunk(self, p_begin: int, p_end: int, P: int, used_ids):
        """Hook of the fused backward: the entries [p_begin, p_end) of every sink buffer in ``used_ids`` are final in
        stream order."""
        if p_begin == 0:
            self._exchanged = set()
        if not self._collectives_on():
            return
        if p_begin == 0 and self.time_exchange and self.flat.is_cuda:
            # the timed window opens where the first range is handed off: with overlapped ranges `exchange_ms` spans first
            # hand-off -> last collective done, i.e. it INCLUDES the part of the backward's tail that runs underneath
            if self._ev is None:
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
            self._window_open = True
        slices = []
        for p, v in zip(self.params, self._views):
            if id(p) in used_ids and p.dim() >= 1 and p.shape[0] == P:
                slices.append(v[p_begin:p_end])              # rows of a contiguous tensor: contiguous
                if p_end == P:
                    self._exchanged.add(id(p))
        self._pending.extend(self._reduce_many(slices))
        if p_end == P:
            # a later in-place accumulation into a view (a second backward, a regulariser on the parameter) would race with
            # the collectives in flight and never be reduced: remember the views' versions as the last range leaves them
            self._versions = {id(p): v._version for p, v in zip(self.params, self._views) if id(p) in self._exchanged}

    def _finish_phased(self):
        # a previous allreduce_phased() whose phase B nobody waited for: the side stream may still be reducing self.flat (and
        # using the staging buffer this path shares) while gather_grads() / the next backward write it -- make the current
        # stream wait (a loop that forgot wait_rest(), or mixed the two APIs, raced silently)
        ex = getattr(self, "_last_phased", None)
        if ex is not None:
            ex.wait_rest()
            self._last_phased = None

    def allreduce(self, average: bool = False):
        import torch.distributed as dist
        self._finish_phased()
        pending, exchanged = getattr(self, "_pending", []), getattr(self, "_exchanged", set())
        for w in pending:
            w.wait()                                         # the current stream waits for the ranges' collectives
        self._pending, self._exchanged = [], set()
        # a range exchange only counts for a parameter whose gradient still IS the bucket view it was written to
        versions = getattr(self, "_versions", {})
        self._versions = {}
        for p, v in zip(self.params, self._views):
            if id(p) in exchanged and v._version != versions.get(id(p), v._version):
                raise RuntimeError("FlatGradBucket: a gradient was modified in place after its ranges had been handed to the "
                                   "overlapped exchange (a second backward or a direct loss term on the parameter?); that "
                                   "contribution raced with the collective and is not reduced -- use one all-reduce after the "
                                   "backward (exchange chunks = 1) for such a loop")
        done = {id(p) for p, v in zip(self.params, self._views)
                if id(p) in exchanged and p.grad is not None and p.grad.data_ptr() == v.data_ptr()}
        self.gather_grads()
        if not self._collectives_on():
            return
        timed = self.time_exchange and self.flat.is_cuda
        if timed and not getattr(self, "_window_open", False):
            if self._ev is None:
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        self._window_open = False
        if not done:
            self._exchange_flat()
        else:
            # what the ranges did not cover, as maximal contiguous runs of the flat buffer
            runs, off, start = [], 0, None
            for p in self.params:
                if id(p) in done:
                    if start is not None:
                        runs.append(self.flat[start:off])
                        start = None
                elif start is None:
                    start = off
                off += p.numel()
            if start is not None:
                runs.append(self.flat[start:off])
            for w in self._reduce_many(runs):
                w.wait()
        if timed:
            self._ev[1].record()
            self._timed_pending = True
        if average:
            self.flat.div_(dist.get_world_size())

    def exchange_ms(self) -> Optional[float]:
        """Duration of the most recent allreduce()'s collectives on the device (HIP events on the current stream, which waits
        for the process group's stream); needs ``time_exchange = True``.  Synchronises on the end event."""
        if getattr(self, "_timed_pending", False):
            self._ev[1].synchronize()
            self.last_exchange_ms = float(self._ev[0].elapsed_time(self._ev[1]))
            self._timed_pending = False
        return self.last_exchange_ms

    # ---- the whole-bucket exchange ------------------------------------------------------------------------------------
    def _exchange_flat(self):
        """SUM of the flat bucket over the ranks, by the configured algorithm:
        "allreduce"  one dist.all_reduce (RCCL picks ring / tree itself);
        "rs_ag"      reduce_scatter_tensor + all_gather_into_tensor on the padded buffer (each rank reduces 1/W of it);
        "direct"     the same two phases as grouped point-to-point transfers to / from EVERY peer at once (on RCCL one
                     ncclGroup per phase = all W-1 xGMI links busy; the shard sums are formed in rank order by ONE rank and
                     then distributed, so the replicas' results are bit-identical by construction).
        gloo (the CPU tests) has no reduce_scatter: "rs_ag" takes the point-to-point route there."""
        import torch.distributed as dist
        world = dist.get_world_size()
        if self.exchange == "allreduce" or world == 1 and self.exchange == "direct":
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            return
        shard = (self.numel + world - 1) // world
        shard = (shard + 63) // 64 * 64
        assert shard * world <= self._store.numel()
        self._exchange_padded(self._store[:shard * world], shard)

    def _exchange_padded(self, buf: torch.Tensor, shard: int):
        """The "rs_ag" / "direct" exchange of a buffer of exactly world * shard elements (see _exchange_flat)."""
        import torch.distributed as dist
        world = dist.get_world_size()
        rank = dist.get_rank()
        mine = buf[rank * shard:(rank + 1) * shard]
        if self.exchange == "rs_ag" and dist.get_backend() != "gloo":
            out = torch.empty_like(mine)
            dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM)
            dist.all_gather_into_tensor(buf, out)
            return
        # phase 1: every rank receives its shard from every peer and sums in rank order
        tmp = getattr(self, "_tmp", None)
        if tmp is None or tmp.numel() < shard * world or tmp.device != buf.device:
            self._tmp = tmp = torch.empty(shard * world, device=buf.device, dtype=buf.dtype)
        tmp = tmp[:shard * world]
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            ops.append(dist.P2POp(dist.isend, buf[peer * shard:(peer + 1) * shard], peer))
            ops.append(dist.P2POp(dist.irecv, tmp[peer * shard:(peer + 1) * shard], peer))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        tmp[rank * shard:(rank + 1) * shard].copy_(mine)
        torch.sum(tmp.view(world, shard), dim=0, out=mine)        # fixed (rank) order on the one rank that owns the shard
        # phase 2: every rank hands its reduced shard to every peer
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            ops.append(dist.P2POp(dist.isend, mine, peer))
            ops.append(dist.P2POp(dist.irecv, buf[peer * shard:(peer + 1) * shard], peer))
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def _exchange_run(self, t: torch.Tensor):
        """SUM of one contiguous 1-D tensor (a run of the flat buffer, or a packed copy) by the configured algorithm.  The
        shard algorithms work on a padded staging copy: a run's length is not a multiple of world x 64."""
        import torch.distributed as dist
        world = dist.get_world_size()
        if self.exchange == "allreduce" or world == 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return
        n = t.numel()
        shard = ((n + world - 1) // world + 63) // 64 * 64
        st = getattr(self, "_stage", None)
        if st is None or st.numel() < shard * world or st.device != t.device:
            self._stage = st = torch.empty(shard * world, device=t.device, dtype=t.dtype)
        buf = st[:shard * world]
        buf[:n].copy_(t)
        buf[n:].zero_()
        self._exchange_padded(buf, shard)
        t.copy_(buf[:n])

    # ---- visible-set exchange: only the rows a rank can have touched --------------------------------------------------------
    def allreduce_visible(self, visible: torch.Tensor, average: bool = False) -> dict:
        """SUM over the ranks like ``allreduce()``, but a rank CONTRIBUTES only the rows of the per-Gaussian tensors that its
        view can have touched: ``visible`` = this rank's ``radii > 0`` (P bools).  A Gaussian the view culled has an exactly-zero
        gradient row in every per-Gaussian tensor (gaussian_renderer/__init__.py:152; the fused backward writes zeros there), so
        leaving those rows out changes nothing -- and a view touches 55-70 % of the Gaussians at S4.

          1. the ranks all-gather their visibility bit masks (P / 8 bytes each);
          2. reduce-scatter by OWNER: Gaussian i belongs to rank i // ceil(P / W); every rank sends each owner the rows
             {visible here} n {owner's shard} of all per-Gaussian tensors, packed (no index travels: the masks say which rows);
          3. the owner adds the contributions in RANK ORDER onto zeros (rows nobody saw stay zero) -- the same sums, in the same
             order, as the "direct" dense exchange forms them: replicas bit-identical by construction;
          4. all-gather of the reduced rows: an owner sends every peer the rows of its shard that ANY rank saw (the union).
        Parameters without a leading Gaussian dimension (the deformation MLP) take the ordinary dense exchange.
        Point-to-point transfers to / from every peer at once (one ncclGroup per phase on RCCL; gloo in the CPU tests).
        Returns {"rows_sent", "rows_received_reduce", "rows_received_gather", "bytes_sent", "bytes_dense"} of this rank.
        UNMEASURED on multi-GPU hardware; tests/test_dp_gloo.py proves the sum equals the dense rank-ordered one bit for bit."""
        import torch.distributed as dist
        self._finish_phased()
        if getattr(self, "_pending", None):
            raise RuntimeError("FlatGradBucket.allreduce_visible: ranges of an overlapped exchange are pending; use allreduce()")
        self.gather_grads()
        P = int(visible.numel())
        per_g = [(p, v) for p, v in zip(self.params, self._views) if p.dim() >= 1 and p.shape[0] == P]
        rest = [v for p, v in zip(self.params, self._views) if not (p.dim() >= 1 and p.shape[0] == P)]
        widths = [v.numel() // P for _, v in per_g]
        D = sum(widths)
        stats = {"rows_sent": 0, "rows_received_reduce": 0, "rows_received_gather": 0, "bytes_sent": 0,
                 "bytes_dense": 2 * self.bytes_per_step}
        if not self._collectives_on():
            return stats
        world, rank = dist.get_world_size(), dist.get_rank()
        dev = self.flat.device
        vis = visible.to(device=dev, dtype=torch.bool).reshape(P)
        # 1. everybody's mask (bit-packed)
        pad = (-P) % 8
        bits = torch.nn.functional.pad(vis, (0, pad)).reshape(-1, 8).to(torch.uint8)
        packed = (bits * (1 << torch.arange(8, device=dev, dtype=torch.uint8))).sum(dim=1, dtype=torch.uint8)
        allp = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(allp, packed)
        unpack = lambda q: ((q.unsqueeze(1) >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(-1)[:P].to(torch.bool)
        masks = [unpack(q) for q in allp]
        shard = (P + world - 1) // world
        lo = lambda o: min(o * shard, P)
        hi = lambda o: min((o + 1) * shard, P)
        rows2d = [v.reshape(P, w) for (_, v), w in zip(per_g, widths)]

        def pack(idx):            # rows `idx` of every per-Gaussian tensor, tensor-major
            return torch.cat([r.index_select(0, idx).reshape(-1) for r in rows2d]) if idx.numel() else self.flat.new_empty(0)
        # 2. my visible rows to their owners; the peers' visible rows of MY shard to me
        send_idx = [torch.nonzero(masks[rank][lo(o):hi(o)]).reshape(-1) + lo(o) for o in range(world)]
        recv_idx = [torch.nonzero(masks[r][lo(rank):hi(rank)]).reshape(-1) + lo(rank) for r in range(world)]
        send = [pack(send_idx[o]) if o != rank else None for o in range(world)]
        recv = [self.flat.new_empty(recv_idx[r].numel() * D) if r != rank else None for r in range(world)]
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            if send[peer].numel():
                ops.append(dist.P2POp(dist.isend, send[peer], peer))
            if recv[peer].numel():
                ops.append(dist.P2POp(dist.irecv, recv[peer], peer))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        # 3. the owner's sums, contributions added in rank order onto zeros
        n_mine = hi(rank) - lo(rank)
        acc = [self.flat.new_zeros(n_mine, w) for w in widths]
        for r in range(world):
            idx = recv_idx[r] - lo(rank)
            if idx.numel() == 0:
                continue
            off = 0
            for a, r2, w in zip(acc, rows2d, widths):
                rows = r2.index_select(0, recv_idx[r]) if r == rank else recv[r][off:off + idx.numel() * w].view(idx.numel(), w)
                a.index_add_(0, idx, rows)          # (idx holds every row once: nothing is accumulated in an unspecified order)
                off += idx.numel() * w
        # 4. the union rows of every shard to everybody
        union = torch.stack(masks).any(dim=0)
        uni_idx = [torch.nonzero(union[lo(o):hi(o)]).reshape(-1) + lo(o) for o in range(world)]
        mine_u = uni_idx[rank] - lo(rank)
        out_chunk = torch.cat([a.index_select(0, mine_u).reshape(-1) for a in acc]) if mine_u.numel() else self.flat.new_empty(0)
        got = [self.flat.new_empty(uni_idx[o].numel() * D) if o != rank else out_chunk for o in range(world)]
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            if out_chunk.numel():
                ops.append(dist.P2POp(dist.isend, out_chunk, peer))
            if got[peer].numel():
                ops.append(dist.P2POp(dist.irecv, got[peer], peer))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for o in range(world):
            n = uni_idx[o].numel()
            if n == 0:
                continue
            off = 0
            for r2, w in zip(rows2d, widths):
                r2.index_copy_(0, uni_idx[o], got[o][off:off + n * w].view(n, w))
                off += n * w
        # rows outside every union: zero on every rank already (nobody saw them)
        for v in rest:
            self._exchange_run(v.reshape(-1))
        if average:
            self.flat.div_(world)
        es = self.flat.element_size()
        stats["rows_sent"] = int(sum(send_idx[o].numel() for o in range(world) if o != rank))
        stats["rows_received_reduce"] = int(sum(recv_idx[r].numel() for r in range(world) if r != rank))
        stats["rows_received_gather"] = int(sum(uni_idx[o].numel() for o in range(world) if o != rank))
        stats["bytes_sent"] = (stats["rows_sent"] + (world - 1) * int(mine_u.numel())) * D * es + (world - 1) * int(packed.numel())
        return stats

    # ---- phased exchange: hidden behind the NEXT iteration --------------------------------------------------------------
    def allreduce_phased(self, first: Iterable[torch.Tensor], sh_rest=None, average: bool = False) -> "PhasedExchange":
        """The exchange in two phases, for a loop that hides most of it behind the next iteration's first kernels.

        The reference's next iteration starts with the deformation MLP on ``xyz.detach()`` and ``t`` (train.py:202-204): of
        everything this exchange produces, only the gradients of ``_xyz`` and of the MLP's own parameters are needed before
        that forward may run (through their optimizer steps).  So:

        * phase A = the slices of ``first`` (GAUSSIAN state: ``_xyz`` 12 B / Gaussian + the MLP's 2.0 MB), exchanged first;
          ``wait_first()`` before ``Adam(first)``;
        * phase B = everything else (f_dc, f_rest, opacity, scaling, rotation: 224 B / Gaussian), exchanged behind phase A on
          the same side stream; ``wait_rest()`` only before the next ``render()``'s preprocess -- i.e. phase B runs underneath
          the next iteration's MLP training forward (0.4 ms at 300k Gaussians) and the optimizer step of ``first``.
          FEATURE state (``first=[]``): the whole feature bucket is phase B, behind the ``no_grad`` MLP forward.
        * ``sh_rest=(f_rest_parameter, active_sh_degree)``: during the SH ramp (train.py:160, one degree per 1000 iterations;
          scene/gaussian_model.py:219-221) the coefficients above the active degree receive an identically zero gradient on
          every rank -- 76 % of the GAUSSIAN bucket at degree 0.  Only the active ``(d + 1)^2 - 1`` of the 15 coefficient rows
          are exchanged (packed, reduced, scattered back); degree 0 exchanges nothing of f_rest.

        On a GPU the collectives and their staging copies run on a side stream (the current stream is free to run the next
        iteration); the handle's waits are stream waits, not host waits.  gloo (CPU tests): synchronous.  With the "direct"
        algorithm every element is summed in rank order wherever it sits, so the result is bit-identical to allreduce()."""
        import torch.distributed as dist
        if getattr(self, "_pending", None):
            raise RuntimeError("FlatGradBucket.allreduce_phased: ranges of an overlapped exchange are pending; use allreduce()")
        self._finish_phased()
        self.gather_grads()
        ex = PhasedExchange(self, average)
        self._last_phased = ex
        if not self._collectives_on():
            return ex
        first_ids = {id(p) for p in first}
        unknown = first_ids - {id(p) for p in self.params}
        if unknown:
            raise ValueError("allreduce_phased: a tensor of `first` is not a parameter of this bucket")
        sh_param, sh_k = None, None
        if sh_rest is not None:
            sh_param, deg = sh_rest
            if id(sh_param) not in {id(p) for p in self.params} or sh_param.dim() != 3:
                raise ValueError("allreduce_phased: sh_rest must name the (P, 15, 3) f_rest parameter of this bucket")
            sh_k = min((int(deg) + 1) ** 2 - 1, sh_param.shape[1])
            if sh_k >= sh_param.shape[1]:
                sh_param = None                                     # full degree: an ordinary slice
        runs_a, runs_b, off = [], [], 0
        for p in self.params:                                       # maximal contiguous runs of the flat buffer per phase
            n = p.numel()
            if sh_param is not None and p is sh_param:
                off += n
                continue
            dst = runs_a if id(p) in first_ids else runs_b
            if dst and dst[-1][1] == off:
                dst[-1][1] = off + n
            else:
                dst.append([off, off + n])
            off += n
        cuda = self.flat.is_cuda
        if cuda:
            side = getattr(self, "_side", None)
            if side is None:
                self._side = side = torch.cuda.Stream(device=self.flat.device)
            side.wait_stream(torch.cuda.current_stream(self.flat.device))      # the gradients are final in current-stream order
            ctx = torch.cuda.stream(side)
        else:
            import contextlib
            side, ctx = None, contextlib.nullcontext()
        world = dist.get_world_size()
        with ctx:
            for a, b in runs_a:
                self._exchange_run(self.flat[a:b])
                if average:
                    self.flat[a:b].div_(world)
            if cuda:
                ex._ev_first = torch.cuda.Event()
                ex._ev_first.record(side)
            for a, b in runs_b:
                self._exchange_run(self.flat[a:b])
                if average:
                    self.flat[a:b].div_(world)
            if sh_param is not None and sh_k > 0:
                view = self._views[[id(p) for p in self.params].index(id(sh_param))]
                packed = view[:, :sh_k, :].contiguous()
                self._exchange_run(packed.view(-1))
                if average:
                    packed.div_(world)
                view[:, :sh_k, :].copy_(packed)
                ex._keep = packed                                   # alive until the side stream is done with it
            if cuda:
                ex._ev_rest = torch.cuda.Event()
                ex._ev_rest.record(side)
        ex.bytes_first = sum(b - a for a, b in runs_a) * self.flat.element_size()
        ex.bytes_rest = (sum(b - a for a, b in runs_b) + (0 if sh_param is None else sh_param.shape[0] * sh_k * sh_param.shape[2])) * self.flat.element_size()
        return ex


class PhasedExchange:
    """Handle of FlatGradBucket.allreduce_phased(): ``wait_first()`` before the optimizer step of the `first` parameters,
    ``wait_rest()`` before anything reads the other gradients (their optimizer step, i.e. before the next render()).  On a GPU
    both make the CURRENT stream wait for the side stream's event; on the CPU the exchange has already happened."""

    def __init__(self, bucket: "FlatGradBucket", average: bool):
        self.bucket, self.average = bucket, average
        self._ev_first = self._ev_rest = None
        self._keep = None
        self.bytes_first = self.bytes_rest = 0

    def wait_first(self):
        if self._ev_first is not None:
            torch.cuda.current_stream(self.bucket.flat.device).wait_event(self._ev_first)
            self._ev_first = None

    def wait_rest(self):
        self.wait_first()
        if self._ev_rest is not None:
            torch.cuda.current_stream(self.bucket.flat.device).wait_event(self._ev_rest)
            self._ev_rest = None
        self._keep = None


def allreduce_densify_stats(xyz_gradient_accum, denom, max_radii2D):
    """Keeps replicas' densify/prune decisions identical (train.py:364-366): SUM, SUM, MAX."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)


# --------------------------------------------------------------------------------------------------------------------
# second axis (BASELINE config 5): ONE view sharded over the ranks by rows of 16x16 tiles
# --------------------------------------------------------------------------------------------------------------------
TILE = 16


def tile_row_partition(image_height: int, world: int, loads=None):
    """[(begin, end)] tile-row ranges, one per rank: contiguous, disjoint, covering ceil(H / 16) rows.

    loads=None: sizes differing by at most one row.  loads = per-tile-row cost (``trase_amd.rasterizer.last_tile_row_loads()``
    of the previous view: binned pairs per row): the contiguous partition that MINIMISES THE LARGEST strip load -- rows of a
    scene are far from uniform (sky vs the subject), and the slowest strip sets the step time.  A constant per-row term
    (``+ mean``) keeps empty rows from piling up in one strip.  Ranks beyond the number of rows get an empty range."""
    rows = (int(image_height) + TILE - 1) // TILE
    if loads is None:
        base, extra = divmod(rows, world)
        out, b = [], 0
        for r in range(world):
            e = b + base + (1 if r < extra else 0)
            out.append((b, e))
            b = e
        return out
    w = [float(x) for x in (loads.tolist() if hasattr(loads, "tolist") else loads)]
    if len(w) != rows:
        raise ValueError(f"loads has {len(w)} entries, the image has {rows} tile rows")
    mean = sum(w) / max(rows, 1)
    w = [x + 0.05 * mean for x in w]                        # pixels / epilogue: a row is never free
    pre = [0.0]
    for x in w:
        pre.append(pre[-1] + x)
    k = min(world, rows)
    # best[j][i] = smallest possible largest-strip load when the first i rows form j strips (rows <= ~70: a tiny table)
    INF = float("inf")
    best = [[INF] * (rows + 1) for _ in range(k + 1)]
    cut = [[0] * (rows + 1) for _ in range(k + 1)]
    best[0][0] = 0.0
    for j in range(1, k + 1):
        for i in range(j, rows + 1):
            for m in range(j - 1, i):
                c = max(best[j - 1][m], pre[i] - pre[m])
                if c < best[j][i]:
                    best[j][i], cut[j][i] = c, m
    bounds, i = [], rows
    for j in range(k, 0, -1):
        bounds.append((cut[j][i], i))
        i = cut[j][i]
    out = bounds[::-1]
    return out + [(rows, rows)] * (world - k)


def strip_pixel_rows(part, rank: int, image_height: int, halo_px: int = 0):
    """Pixel rows [y0, y1) of a rank's strip, optionally widened by a halo (clipped to the image): the 11x11 SSIM window
    of utils/loss_utils.py:56-86 needs 5 rows of its neighbours' pixels."""
    b, e = part[rank]
    y0, y1 = min(b * TILE, image_height), min(e * TILE, image_height)
    return max(0, y0 - halo_px), min(image_height, y1 + halo_px)


def allgather_strips(local: torch.Tensor, part, image_height: int):
    """Every rank contributes the rows of ITS strip of a (C, H, W) map (what ``render`` under ``tile_rows`` returned: zeros
    elsewhere); every rank gets the full map.  One ``all_gather_into_tensor`` of the strips padded to the tallest one
    (a zero-padded all-reduce of the whole map moves twice the bytes).  Differentiable: the gradient of a rank's strip is
    the matching slice of the full map's gradient."""
    return _AllGatherStrips.apply(local, part, image_height)


def allreduce_viewspace_grad(viewspace_points: torch.Tensor):
    """Tile-row sharding only: every rank's ``viewspace_points.grad`` is its strip's PARTIAL sum.  The reference's
    ``add_densification_stats`` takes a norm of it (scene/gaussian_model.py:637-639), so the partial sums must be added
    BEFORE the statistics (sum of norms >= norm of sum) -- call this first and do NOT call ``allreduce_densify_stats``
    on this axis (every rank then holds the same statistics already)."""
    import torch.distributed as dist
    if viewspace_points.grad is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dist.all_reduce(viewspace_points.grad, op=dist.ReduceOp.SUM)


class _AllGatherStrips(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, part, image_height):
        import torch.distributed as dist
        ctx.rows = None
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return local.detach().clone()
        world = dist.get_world_size()
        spans = [strip_pixel_rows(part, r, image_height) for r in range(world)]
        ctx.rows = spans[dist.get_rank()]
        y0, y1 = ctx.rows
        tall = max(b - a for a, b in spans)
        C, _, W = local.shape
        mine = local.new_zeros(C, tall, W)
        mine[:, :y1 - y0] = local.detach()[:, y0:y1]
        gathered = local.new_empty(world, C, tall, W)
        if dist.get_backend() == "gloo":          # the CPU test backend has no all_gather_into_tensor
            dist.all_gather(list(gathered.unbind(0)), mine)
        else:
            dist.all_gather_into_tensor(gathered, mine)
        full = torch.empty_like(local)
        for r, (a, b) in enumerate(spans):
            full[:, a:b] = gathered[r, :, :b - a]
        return full

    @staticmethod
    def backward(ctx, g):
        if ctx.rows is None:
            return g, None, None
        y0, y1 = ctx.rows
        out = torch.zeros_like(g)
        out[:, y0:y1] = g[:, y0:y1]
        return out, None, None
