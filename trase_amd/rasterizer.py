"""Host-side mirror of the reference's operator interface for the rasterizer path.

Same names, argument meaning, return order and error behaviour as the
``diff_gaussian_rasterization`` extension the reference imports at
gaussian_renderer/__init__.py:21 and calls at :58-73 / :137-146 (also
gui_standalone.py:59,390-478): ``GaussianRasterizationSettings`` (12-field
record) and ``GaussianRasterizer(raster_settings)(means3D, means2D, shs,
sh_objs, colors_precomp, opacities, scales, rotations, cov3D_precomp) ->
(image, radii, feats, depth)``.

All compute goes through the C-ABI HIP library (include/trase_rast.h); PyTorch
only provides device memory, the current stream and the autograd hook.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# --------------------------------------------------------------------------------------
# capacity policy for the (sub-tile, Gaussian) pair buffers
# --------------------------------------------------------------------------------------
class _Policy:
    """sync=True  (default): like the reference, read the pair count back after stage 1 and allocate exactly
                  (one host synchronisation per forward).
    sync=False: no host synchronisation on the steady path.  Buffers are sized from ``capacity``; the very first
                forward (capacity unknown) sizes itself with one synchronising read.  After every forward the geom
                header (pair count, overflow flag, binning guards) is copied to pinned host memory behind an event;
                the NEXT forwards poll those events without stalling, grow the capacity to 1.25 x the largest count seen,
                and RAISE if a previous forward overflowed (its images / gradients were incomplete): the caller re-runs
                that iteration -- the capacity has already been grown.  ``check_overflow()`` drains the queue blocking."""
    sync = os.environ.get("TRASE_RAST_SYNC", "1") != "0"
    capacity = 0
    variant = int(os.environ.get("TRASE_RAST_VARIANT", "0"), 0)
    last_geom: Optional[torch.Tensor] = None
    last_bin: Optional[torch.Tensor] = None     # bin workspace of the most recent forward (for last_tile_row_loads)
    last_hw = None
    last_strip = (0, 0)         # tile-row strip the most recent forward ran with
    guard_reduced = None        # (forward_seq, device int32[64]) -- header with its FLAG words MAX-reduced over the ranks (see current_guard)
    forward_seq = 0             # counts sync-free forwards: the key of guard_reduced (pinned buffers are recycled, ids repeat)
    last_capacity = 0
    pending: list = []          # [(event, pinned int32[32] header copy, capacity of that call)]
    rollbacks: dict = {}        # id(pinned header) -> callbacks to run if that forward turns out to have overflowed
    max_pending = 8
    tile_rows = (0, 0)          # strip of 16x16-tile rows to render; (0, 0) = the whole image
    feat_bg = 0.0               # background value of the feature channels (lineage switch, variant bit 0x10000)


def set_sync(flag: bool, capacity: int = 0):
    """Choose the capacity policy.  ``capacity`` (pairs) seeds the sync-free policy; 0 = measure on the first call."""
    _Policy.sync = bool(flag)
    _Policy.capacity = int(capacity)
    _Policy.pending = []
    _Policy.rollbacks = {}


def set_graph(flag="auto"):
    """Launch-graph replay (include/trase_rast.h ``trase_rast_graph_mode``): with the sync-free policy, a forward or backward
    whose argument record (sizes, every pointer) repeats is replayed as ONE hipGraph launch instead of ~45 kernel launches.
    Pays on small workloads, where the launches cost more host time than the kernels run (BASELINE configs 1 and 2: 1 k and
    150 k Gaussians); at the 1080p headline the step is GPU-bound either way.  ``"auto"`` -- the LIBRARY DEFAULT (round 5:
    a drop-in user of render() gets the small-scene numbers without asking) -- replays calls of at most 200 000 Gaussians
    (``TRASE_GRAPH_AUTO_P``); True: every call; False: never.  Only the sync-free entry points (``set_sync(False)``) are
    replayed; records repeat when the allocator hands out the same blocks every iteration (a steady training loop); if they
    do not, the library switches the mode off by itself.  Environment: ``TRASE_GRAPH=0|1|auto``."""
    mode = 2 if flag == "auto" else (1 if flag else 0)
    _lib.check(_lib.load().trase_rast_graph_mode(mode), "trase_rast_graph_mode")


def graph_stats() -> dict:
    st = (C.c_int64 * 4)()
    _lib.check(_lib.load().trase_rast_graph_stats(C.byref(st)), "trase_rast_graph_stats")
    return {"hits": int(st[0]), "misses": int(st[1]), "cached": int(st[2]), "enabled": bool(st[3])}


def set_variant(v: int):
    _Policy.variant = int(v)


# bits of TraseRastSettings.variant (include/trase_rast.h `TraseVariant`)
VARIANT_DEPTH_GRAD, VARIANT_FEATS_BG, VARIANT_DEPTH_NORM = 0x100, 0x10000, 0x20000      # lineage switches
VARIANT_FEATURES_ONLY_BWD = 0x400                                                       # backward scope
VARIANT_VALU_BACKWARD, VARIANT_VALU_FORWARD, VARIANT_SLOT_LISTS = 0x40, 0x2000, 0x100000  # cross-check formulations
VARIANT_SPARSE_STRIP_GRADS = 0x200000                                                   # tile-row strips: live rows only
VARIANT_FORWARD_ONLY = 0x800000           # a forward under no_grad: backward-only state is not stored (set per call by the fused render())
VARIANT_DEPTH32 = 0x400000                # depth sort on the raw float32 depth bits (the fallback of the 27-bit keys, see set_depth_keys)


def set_depth_keys(bits: int = 27):
    """Depth-sort keys.  27 (default): the float32 depth bits above those of the 0.2 near-cull plane, in three 9-bit radix
    passes -- the same order as the lineage's float keys for every view depth below 13 107 scene units, three launches and
    16 us per view cheaper than 32: the raw float bits in four 8-bit passes.  The library watches for a saturated key and
    switches the process to 32 by itself (sync policy: the forward is repeated at once; sync-free: that iteration is
    reported like a pair-buffer overflow -- guarded consumers skip it -- and every later forward uses 32)."""
    if bits not in (27, 32):
        raise ValueError("depth keys are 27 or 32 bits")
    _Policy.variant = (_Policy.variant & ~VARIANT_DEPTH32) | (VARIANT_DEPTH32 if bits == 32 else 0)


def set_sparse_strip_grads(flag: bool = True):
    """Tile-row strips (``tile_rows``) through the fused ``render()``: the backward writes ONLY the gradient rows of the
    Gaussians that have a pair in the strip (~1 / world of them) into persistent per-parameter tensors that are zero
    everywhere else -- zero-filled once; the rows a backward wrote are cleared again before the next one writes
    (``trase_rast_zero_live_rows``).  The per-Gaussian tail of a strip's backward then costs what the strip holds instead
    of what the scene does (2.5 M Gaussians: 0.53 -> 0.16 ms at 8 strips).  The returned gradients are the usual dense
    tensors -- but, as with ``set_grad_sink``, they are REUSED: a ``.grad`` kept from an earlier iteration is overwritten by
    the next strip backward.  Off by default."""
    v = _Policy.variant & ~VARIANT_SPARSE_STRIP_GRADS
    _Policy.variant = v | (VARIANT_SPARSE_STRIP_GRADS if flag else 0)


def set_lineage(feats_bg: Optional[float] = None, depth_normalised: bool = False, depth_grad: bool = False):
    """Lineage switches (the CUDA fork's source is absent, these are the three places it is most
    likely to differ from the public lineage; oracle/raster_oracle.py ``OracleOptions`` has the same switches):

    feats_bg         None (default, gaussian-grouping: no background term) or a float b: feats[c] += T_final * b
    depth_normalised False (default, Deformable-3DGS: blended depth as is) or True: depth = sum(w z) / (1 - T_final)
    depth_grad       False (default, lineage: the depth output carries no gradient) or True: dL/ddepth is honoured

    They flip both directions of the HIP path; forward and backward of one call always use the same setting."""
    v = _Policy.variant & ~(VARIANT_DEPTH_GRAD | VARIANT_FEATS_BG | VARIANT_DEPTH_NORM)
    if feats_bg is not None:
        v |= VARIANT_FEATS_BG
    if depth_normalised:
        v |= VARIANT_DEPTH_NORM
    if depth_grad:
        v |= VARIANT_DEPTH_GRAD
    _Policy.variant = v
    _Policy.feat_bg = 0.0 if feats_bg is None else float(feats_bg)


def set_tile_rows(begin: int = 0, end: int = 0):
    """Render only the rows [begin, end) of 16x16 tiles (the tile-sharding axis of SURVEY.md 8e: one view split over
    ranks; ``trase_amd.dp.tile_row_partition`` gives each rank its range).  Pixels outside the strip come back as zeros,
    per-Gaussian gradients are the strip's partial sums (the ranks' gradients add up to the whole view's -- the same
    all-reduce as for view parallelism), ``radii`` stay whole-image.  (0, 0) restores the whole image."""
    if begin < 0 or end < begin:
        raise ValueError("tile rows must satisfy 0 <= begin <= end")
    _Policy.tile_rows = (int(begin), int(end))


class tile_rows:
    """``with tile_rows(b, e): out = render(...); loss.backward()`` -- the backward uses the strip its forward ran with."""

    def __init__(self, begin: int, end: int):
        self.rows = (begin, end)

    def __enter__(self):
        self.saved = _Policy.tile_rows
        set_tile_rows(*self.rows)
        return self

    def __exit__(self, *exc):
        _Policy.tile_rows = self.saved


class IterationSkipped(RuntimeError):
    """Raised (by a LATER call: forward, check_overflow) when a sync-free forward turns out to have produced incomplete results
    -- its pair buffer overflowed, or a 27-bit depth key saturated.  The remedy has already been applied (capacity grown / float
    depth keys selected) and the guarded consumers of that iteration skipped it on the device, so a training loop can catch this
    and simply go on::

        try:
            out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
        except rasterizer.IterationSkipped:
            out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)      # the report concerned an EARLIER iteration

    A RuntimeError subclass: code that catches RuntimeError keeps working."""


def _header_verdict(h, cap: int, what: str):
    """h: 32 int32 words of a geom header.  Grows the capacity; raises on overflow / tripped binning guards."""
    r_eff = int(h[2]) & 0xffffffff
    if _Policy.capacity:
        _Policy.capacity = max(_Policy.capacity, int(r_eff * 1.25) + 1024)
    if int(h[16]) or int(h[20]):
        raise RuntimeError(f"trase_amd rasterizer: binning guard tripped in {what} (key flag {int(h[16])}, slot flag {int(h[20])})")
    if int(h[1]) & 2:          # bit 1 (OR-ed over the ranks under data parallelism): a saturated 27-bit depth key
        _Policy.variant |= VARIANT_DEPTH32
        raise IterationSkipped(f"trase_amd rasterizer: a Gaussian of {what} lies beyond the range of the 27-bit depth keys (view depth > "
                           f"13 107): that call's depth order was not exact beyond that distance.  This process now sorts on the raw "
                           f"float32 depth bits (rasterizer.set_depth_keys(32)).  Guarded consumers of that iteration "
                           f"(FusedAdam.step, add_densification_stats) skipped it on the device; unguarded ones have used its "
                           f"gradients: re-run the iteration, or call set_depth_keys(32) / set_sync(True) up front for such scenes.")
    if int(h[1]) & 1:
        raise IterationSkipped(f"trase_amd rasterizer: the pair buffer overflowed in {what}: {r_eff} (sub-tile, Gaussian) pairs "
                           f"needed, capacity was {cap}; that call's outputs and gradients were incomplete.  The capacity "
                           f"has been grown to {_Policy.capacity}.  Guarded consumers of that iteration (FusedAdam.step, "
                           f"add_densification_stats) skipped it on the device on every rank (under data parallelism the "
                           f"flag is MAX-reduced over the ranks first): parameters and moments are unchanged.  Consumers "
                           f"that are NOT guarded (torch.optim.Adam, your own statistics) have used the incomplete "
                           f"gradients: re-run the iteration from a checkpoint, or use set_sync(True).")


def _poll_pending(block: bool = False):
    keep = []
    err = None
    for k, (ev, pin, cap) in enumerate(_Policy.pending):
        must = block or (len(_Policy.pending) - k) > _Policy.max_pending
        if must:
            ev.synchronize()
        if must or ev.query():
            undo = _Policy.rollbacks.pop(id(pin), [])
            try:
                _header_verdict(pin.tolist(), cap, "a previous sync-free forward")
            except RuntimeError as e:      # (IterationSkipped or a tripped guard) report the first, still drain the rest
                err = err or e
                for fn in undo:            # host-side bookkeeping of steps the device skipped (Adam step counters)
                    fn()
            if len(_PIN_RING) < 16:
                _PIN_RING.append((pin, ev))
        else:
            keep.append((ev, pin, cap))
    _Policy.pending = keep
    if err is not None:
        raise err


def current_guard():
    """(geom workspace, on_overflow) of the most recent forward under the sync-free policy, else None.  The geom header is
    what the guarded device-side consumers (``FusedAdam.step``, ``add_densification_stats``) test: a forward that
    overflowed its pair buffer produced incomplete gradients, and the optimizer step that would consume them is skipped ON
    THE DEVICE (the reference skips ``optimizer.step()`` on a bad iteration, train.py:298-301, :378).  ``on_overflow(fn)``
    registers host-side bookkeeping to undo once the overflow is reported."""
    if _Policy.sync or _Policy.last_geom is None or not _Policy.pending:
        return None
    ev, pin, _cap = _Policy.pending[-1]
    guard = _Policy.last_geom
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # Data-parallel replicas must take the SAME decision: the gradients every rank applies are the all-reduced ones, so
        # an overflow on ANY rank contaminates all of them.  The header is MAX-reduced over the ranks once per forward (a
        # 256-byte collective, stream-ordered, no host synchronisation); the guarded kernels test that copy, and the pinned
        # host copy the overflow report reads is refreshed from it, so the host-side rollbacks agree across ranks as well.
        # Every rank calls this the same number of times in the same order (FusedAdam.step / add_densification_stats).
        # The cache is keyed on the forward's sequence number: pinned header buffers are recycled (LIFO) as soon as their
        # verdict has been read, so the SAME pin object serves consecutive forwards and its id says nothing (with
        # id(pin) as the key the all-reduce was skipped for every later forward on the ranks that happened to reuse a pin --
        # stale flags, and a collective some ranks issued and others did not).
        red = _Policy.guard_reduced
        if red is None or red[0] != _Policy.forward_seq:
            buf = guard[:256].view(torch.int32).clone()
            # only the FLAG words are shared (overflow, pairs needed, key / slot guards): the rest of the header (capacity,
            # live count, pack bits of THIS rank's strip) stays local, so reports and capacity growth describe this rank
            idx = _flag_index(buf.device)
            flags = buf.index_select(0, idx)
            # word 1 is a BIT FIELD (bit 0 pair overflow, bit 1 saturated depth key): a MAX over the ranks of the word itself
            # turns {1, 2} into 2 and loses the overflow -- its two bits travel as words of their own and meet again
            bits = torch.stack([flags[0] & 1, flags[0] & 2])
            flags = torch.cat([flags, bits])
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            flags[0] = flags[-2] | flags[-1]
            buf.index_copy_(0, idx, flags[:-2])
            pin.copy_(buf[:32], non_blocking=True)
            if guard.is_cuda:
                ev.record(torch.cuda.current_stream(guard.device))     # the poll now waits for the reduced copy
            _Policy.guard_reduced = red = (_Policy.forward_seq, buf)
        guard = red[1]
    return guard, _Policy.rollbacks.setdefault(id(pin), []).append


_FLAG_WORDS = (1, 2, 16, 20)     # geom header: overflow flag, pairs needed, key guard, slot guard (common.h GeomHeader)
_FLAG_INDEX: dict = {}


def _flag_index(device) -> torch.Tensor:
    t = _FLAG_INDEX.get(device)
    if t is None:
        t = _FLAG_INDEX[device] = torch.tensor(_FLAG_WORDS, dtype=torch.int64, device=device)
    return t


def check_overflow():
    """Blocking: wait for every sync-free forward issued so far and raise if one of them overflowed its pair buffer."""
    _poll_pending(block=True)


def _pick_capacity(lib, ws, stream, rerun_stage1=None) -> int:
    """Pair capacity of the forward whose stage 1 has just been enqueued on `stream`.  rerun_stage1(): repeats stage 1 with
    32-bit depth keys -- called when the status read finds a saturated 27-bit key (set_depth_keys)."""
    if _Policy.sync or _Policy.capacity <= 0:
        st = (C.c_int64 * 3)()
        _lib.check(lib.trase_rast_status(C.byref(ws), C.byref(st), stream), "trase_rast_status")
        if int(st[1]) & 2 and rerun_stage1 is not None:
            _Policy.variant |= VARIANT_DEPTH32
            rerun_stage1()
            _lib.check(lib.trase_rast_status(C.byref(ws), C.byref(st), stream), "trase_rast_status")
        need = max(int(st[2]), 1)              # pairs after exact sub-tile culling
        if _Policy.sync:
            return need
        _Policy.capacity = int(need * 1.5) + 1024      # first sync-free call: measured once, with headroom
        return _Policy.capacity
    if not _capturing():
        _poll_pending()
    return max(int(_Policy.capacity), 1)


_PIN_RING: list = []       # pinned header buffers + their events, recycled once their verdict has been read
_SIZES: dict = {}          # (P, W, H, F, capacity) -> workspace sizes (a ctypes call per lookup otherwise)


def _sizes(lib, P: int, W: int, H: int, F: int, capacity: int):
    key = (P, W, H, F, capacity)
    v = _SIZES.get(key)
    if v is None:
        sz = _lib.RastSizes()
        _lib.check(lib.trase_rast_sizes(P, W, H, F, capacity, C.byref(sz)), "trase_rast_sizes")
        v = (sz.geom_bytes, sz.bin_bytes, sz.img_bytes, sz.pre_bytes, sz.tmp_bytes, sz.bwd_tmp_bytes)
        if len(_SIZES) > 256:
            _SIZES.clear()
        _SIZES[key] = v
    return v


def _release_last():
    """Called at the START of a forward: drop the references to the previous forward's geom / bin workspaces (kept for
    last_status / last_tile_row_loads / current_guard, all of which are asked between a forward and the next one).  Holding
    them across the next forward's allocations made the caching allocator alternate between two sets of blocks: every
    second launch-graph record missed, and the peak memory was one bin buffer higher."""
    _Policy.last_geom = None
    _Policy.last_bin = None
    _Policy.guard_reduced = None


def _after_render(geom: torch.Tensor, capacity: int, binb: Optional[torch.Tensor] = None, hw=None):
    _Policy.last_geom, _Policy.last_capacity = geom, capacity
    _Policy.last_bin, _Policy.last_hw = binb, hw
    _Policy.last_strip = tuple(_Policy.tile_rows)
    _Policy.guard_reduced = None
    if not _Policy.sync and _capturing():
        # inside torch.cuda.graph: nothing may be read back (a pinned copy + event recorded here would be replayed, never polled).
        # The overflow flag still lives in the geom header and the guarded consumers (FusedAdam.step, add_densification_stats)
        # test it on the device at every replay; the HOST never hears of it -- size the capacity generously before capturing.
        _Policy.forward_seq += 1
        return
    if not _Policy.sync:
        _Policy.forward_seq += 1
        if _PIN_RING:
            pin, ev = _PIN_RING.pop()
        else:
            pin, ev = torch.empty(32, dtype=torch.int32).pin_memory(), torch.cuda.Event()
        pin.copy_(geom[:128].view(torch.int32), non_blocking=True)
        ev.record(torch.cuda.current_stream(geom.device))
        _Policy.pending.append((ev, pin, capacity))
        _Policy.rollbacks[id(pin)] = []


def last_status():
    """(num_rendered, overflow, num_rendered_after_culling) of the most recent forward. Synchronises."""
    if _Policy.last_geom is None:
        return None
    lib = _lib.load()
    ws = _lib.RastWorkspace()
    ws.geom = _lib.ptr(_Policy.last_geom)
    ws.geom_bytes = _Policy.last_geom.numel()
    st = (C.c_int64 * 3)()
    _lib.check(lib.trase_rast_status(C.byref(ws), C.byref(st), _stream(_Policy.last_geom.device)), "trase_rast_status")
    return int(st[0]), int(st[1]), int(st[2])


def last_tile_row_loads(allreduce: bool = False) -> torch.Tensor:
    """Binned (8x8 sub-tile, Gaussian) pairs per ROW of 16x16 tiles of the most recent forward -- what a tile-row strip
    costs to composite.  ``trase_amd.dp.tile_row_partition(H, world, loads=...)`` turns the previous view's loads into a
    load-balanced partition (consecutive training views see similar loads).  Synchronises (one small D2H copy).

    The loads must describe the WHOLE image on every rank, or the ranks compute different partitions (gaps / overlaps in
    ``allgather_strips``).  A forward that ran under ``tile_rows`` binned only its own strip: then pass ``allreduce=True``
    (the ranks' strips are disjoint, their loads are SUMMED over the process group -- every rank must call it), otherwise
    this raises."""
    if _Policy.last_bin is None or _Policy.last_hw is None:
        raise RuntimeError("no forward has run yet")
    if _Policy.last_strip != (0, 0) and not allreduce:
        raise RuntimeError(f"last_tile_row_loads: the last forward rendered only the tile rows {_Policy.last_strip}; its loads "
                           "cover that strip alone.  Take the loads from a whole-image forward, or pass allreduce=True to sum "
                           "the ranks' strips")
    H, W = _Policy.last_hw
    gx8, gy8 = (W + 7) // 8, (H + 7) // 8
    cap = int(_Policy.last_capacity)
    off = 2 * ((4 * cap + 255) // 256 * 256)              # BinBuf: point_list | pair_slot | ranges (trase_amd/csrc/api.hip carve_bin)
    rng = _Policy.last_bin[off:off + 8 * gx8 * gy8].view(torch.int32).reshape(gy8, gx8, 2).to(torch.int64)
    per_sub_row = (rng[..., 1] - rng[..., 0]).clamp_min(0).sum(dim=1)
    if gy8 % 2:
        per_sub_row = torch.cat([per_sub_row, per_sub_row.new_zeros(1)])
    loads = per_sub_row.reshape(-1, 2).sum(dim=1)
    if allreduce and _Policy.last_strip != (0, 0):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(loads, op=dist.ReduceOp.SUM)
    return loads.cpu()


def geom_view(geom: torch.Tensor, P: int) -> dict:
    """Typed views of the per-Gaussian forward state inside a geom workspace (introspection for the parity tests;
    the counterpart of inspecting the reference's geomBuffer): xy (P,2), conic_opacity (P,4), rgb_depth (P,4),
    tiles (P,) int32.  Entries of culled Gaussians (radii == 0) are undefined; `rgb_depth` is not written by a fused render()
    that ran under torch.no_grad() (TRASE_VARIANT_FORWARD_ONLY: only a backward reads it)."""
    off = (C.c_int64 * 6)()
    _lib.check(_lib.load().trase_rast_geom_layout(int(P), C.byref(off)), "trase_rast_geom_layout")

    def sl(o, nbytes, dtype, *shape):
        return geom[o:o + nbytes].view(dtype).reshape(*shape)
    return {"xy": sl(off[1], 8 * P, torch.float32, P, 2), "conic_opacity": sl(off[2], 16 * P, torch.float32, P, 4),
            "rgb_depth": sl(off[3], 16 * P, torch.float32, P, 4), "tiles": sl(off[4], 4 * P, torch.int32, P)}


def last_geom_view(P: int) -> dict:
    """geom_view of the most recent forward's geom workspace."""
    if _Policy.last_geom is None:
        raise RuntimeError("no forward has run yet")
    return geom_view(_Policy.last_geom, P)


def _capturing() -> bool:
    """True while the current stream is being captured into a graph (whole-iteration capture with torch.cuda.graph)."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


_LAST_STREAM: dict = {}     # device index -> the stream of the library's most recent launch sequence on that device
_UNORDERED_STREAMS = os.environ.get("TRASE_UNORDERED_STREAMS", "0") != "0"   # experiments only (profiles/r4_two_streams.md): no cross-stream wait


def set_stream_ordering(flag: bool = True):
    """True (default): launch sequences of the library on DIFFERENT streams of one device are kept in order (a device-side wait on a
    change of stream) -- the conservative guard of rounds 4-5.  False: every call simply goes to the caller's current stream, like
    any torch operator; views on two streams then share the chip (+5 % views/s at S4, `views_in_flight_2` of the bench line).
    Safe since round 6: the corruption the guard was built against hit compiler-generated packed-FP32 VALU instructions beside
    MFMA + transposing-read kernels, and the library is built without them (profiles/r6_two_streams.md; 288 views in flight on 2 / 3
    streams bit-identical to the serial run).  Gradients of views in flight must not share `.grad` tensors (use
    torch.autograd.grad, or per-stream parameters): autograd's accumulation across streams is the caller's business."""
    global _UNORDERED_STREAMS
    _UNORDERED_STREAMS = not bool(flag)


def _stream(device) -> C.c_void_p:
    """The caller's current HIP stream -- every launch sequence of the library goes to it, in order.

    Launch sequences of ONE device are also kept in order ACROSS streams: when the current stream is not the one the previous
    sequence went to, it first waits (on the device, no host synchronisation) for that stream.  Two sequences of the library
    sharing the chip is not a supported mode: measured in round 4, the per-Gaussian kernels produce wrong values while a
    compositing kernel (transposing LDS reads, `ds_read_b64_tr_b16`) of another sequence is resident on their CU
    (profiles/r4_two_streams.md).  Costs nothing while the caller stays on one stream."""
    cur = torch.cuda.current_stream(device)
    if _capturing():
        # stream capture (torch.cuda.graph): the capture stream was ordered behind the caller's stream by whoever began the capture;
        # an event wait recorded here would tie the graph to work outside it.  The bookkeeping below stays as it was before the capture.
        return C.c_void_p(cur.cuda_stream)
    prev = _LAST_STREAM.get(cur.device_index)
    if prev is None or prev.cuda_stream != cur.cuda_stream:
        if prev is not None and not _UNORDERED_STREAMS:
            cur.wait_stream(prev)
        _LAST_STREAM[cur.device_index] = cur
    return C.c_void_p(cur.cuda_stream)


def _prep(t: Optional[torch.Tensor], name: str, device) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous():      # the steady-state case: nothing to do
        return t
    if t.device != device:
        raise ValueError(f"{name} must live on {device}, got {t.device}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_POISON = os.environ.get("TRASE_POISON", "0") != "0"   # debug: fill every workspace with 0xFF to expose reads of unwritten memory


def _bytes(n: int, device) -> torch.Tensor:
    if _POISON:
        return torch.full((max(int(n), 1),), 255, dtype=torch.uint8, device=device)
    return torch.empty(max(int(n), 1), dtype=torch.uint8, device=device)


def _output_maps(F: int, H: int, W: int, device, zero: bool):
    """(image (3,H,W), feats (F,H,W), depth (1,H,W)) as three slices of ONE allocation (one allocator round trip instead
    of three; a strip render starts from zeros because it leaves the other rows untouched)."""
    buf = (torch.zeros if zero else torch.empty)(3 + F + 1, H, W, device=device)
    return buf[:3], buf[3:3 + F], buf[3 + F:]


def _fill_settings(rs: GaussianRasterizationSettings, device, keep: list) -> _lib.RastSettings:
    s = _lib.RastSettings()
    s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
    s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree = int(rs.sh_degree)
    s.prefiltered, s.debug = int(bool(rs.prefiltered)), int(rs.debug)   # debug=2 additionally names every kernel on stderr
    s.device = device.index if device.index is not None else torch.cuda.current_device()
    s.variant = _Policy.variant
    s.tile_row_begin, s.tile_row_end = _Policy.tile_rows
    s.feat_bg = _Policy.feat_bg
    for name in ("bg", "viewmatrix", "projmatrix", "campos"):
        t = _prep(getattr(rs, name), name, device)
        if t is None:
            raise ValueError(f"raster_settings.{name} is required")
        keep.append(t)
        setattr(s, name, t.data_ptr())
    return s


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, sh_objs, colors_precomp, opacities, scales, rotations,  # noqa: C901
                cov3Ds_precomp, raster_settings):
        lib = _lib.load()
        _release_last()
        device = means3D.device
        if device.type != "cuda":
            raise RuntimeError("trase_amd rasterizer runs on the GPU only (there is no CPU path); "
                               f"means3D is on {device}")
        means3D = _prep(means3D, "means3D", device)
        P = means3D.shape[0]
        sh = _prep(sh, "shs", device)
        sh_objs = _prep(sh_objs, "sh_objs", device)
        colors_precomp = _prep(colors_precomp, "colors_precomp", device)
        opacities = _prep(opacities, "opacities", device)
        scales = _prep(scales, "scales", device)
        rotations = _prep(rotations, "rotations", device)
        cov3Ds_precomp = _prep(cov3Ds_precomp, "cov3D_precomp", device)
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        M = sh.shape[1] if sh is not None else 0
        F = sh_objs.shape[-1] if sh_objs is not None else 0
        if F == 0:
            sh_objs = None

        keep: list = []
        s = _fill_settings(raster_settings, device, keep)
        inp = _lib.RastInputs()
        inp.P, inp.M, inp.F = P, M, F
        inp.means3D = _lib.ptr(means3D)
        inp.shs, inp.sh_objs, inp.colors_precomp = _lib.ptr(sh), _lib.ptr(sh_objs), _lib.ptr(colors_precomp)
        inp.opacities, inp.scales, inp.rotations = _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations)
        inp.cov3D_precomp = _lib.ptr(cov3Ds_precomp)

        image, feats, depth = _output_maps(F, H, W, device, bool(s.tile_row_begin or s.tile_row_end))
        radii = torch.empty(P, dtype=torch.int32, device=device)
        out = _lib.RastOutputs()
        out.image, out.radii, out.depth = _lib.ptr(image), _lib.ptr(radii), _lib.ptr(depth)
        out.feats = _lib.ptr(feats) if F > 0 else None

        geom_b, _, img_b, pre_b, _, _ = _sizes(lib, P, W, H, F, 1)
        geom = _bytes(geom_b, device)
        pre = _bytes(pre_b, device)
        img = _bytes(img_b, device)
        ws = _lib.RastWorkspace()
        ws.geom, ws.geom_bytes = _lib.ptr(geom), geom.numel()
        ws.pre, ws.pre_bytes = _lib.ptr(pre), pre.numel()
        ws.img, ws.img_bytes = _lib.ptr(img), img.numel()
        stream = _stream(device)

        one_call = (not _Policy.sync) and _Policy.capacity > 0      # capacity known beforehand: one boundary crossing
        if not one_call:
            _lib.check(lib.trase_rast_preprocess(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws), stream),
                       "trase_rast_preprocess")
        def _again():              # a saturated 27-bit depth key: stage 1 once more on the raw float bits
            s.variant |= VARIANT_DEPTH32
            _lib.check(lib.trase_rast_preprocess(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws), stream), "trase_rast_preprocess")
        capacity = _pick_capacity(lib, ws, stream, None if one_call else _again)
        _, bin_b, _, _, tmp_b, _ = _sizes(lib, P, W, H, F, capacity)
        binb = _bytes(bin_b, device)
        tmp = _bytes(tmp_b, device)
        ws.bin, ws.bin_bytes = _lib.ptr(binb), binb.numel()
        ws.tmp, ws.tmp_bytes = _lib.ptr(tmp), tmp.numel()
        ws.capacity = capacity
        if one_call:
            _lib.check(lib.trase_rast_forward(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws), stream), "trase_rast_forward")
        else:
            _lib.check(lib.trase_rast_render(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws), stream),
                       "trase_rast_render")
        _after_render(geom, capacity, binb, (H, W))

        ctx.raster_settings = raster_settings
        ctx.variant, ctx.tile_rows, ctx.feat_bg = s.variant, (s.tile_row_begin, s.tile_row_end), s.feat_bg
        ctx.capacity = capacity
        ctx.dims = (P, M, F, H, W)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        # the normalised-depth switch with a depth gradient needs the forward's depth map again
        keep_depth = (s.variant & (VARIANT_DEPTH_NORM | VARIANT_DEPTH_GRAD)) == (VARIANT_DEPTH_NORM | VARIANT_DEPTH_GRAD)
        ctx.save_for_backward(means3D, sh, sh_objs, colors_precomp, opacities, scales, rotations,
                              cov3Ds_precomp, radii, geom, binb, img, pre, depth if keep_depth else None)
        return image, radii, feats, depth

    @staticmethod
    def backward(ctx, grad_image, grad_radii, grad_feats, grad_depth):
        lib = _lib.load()
        (means3D, sh, sh_objs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
         radii, geom, binb, img, pre, depth_out) = ctx.saved_tensors
        P, M, F, H, W = ctx.dims
        device = means3D.device
        keep: list = []
        s = _fill_settings(ctx.raster_settings, device, keep)
        s.variant = ctx.variant                # the forward's variant (a global change in between must not split the pair)
        s.tile_row_begin, s.tile_row_end = ctx.tile_rows
        s.feat_bg = ctx.feat_bg
        inp = _lib.RastInputs()
        inp.P, inp.M, inp.F = P, M, F
        inp.means3D = _lib.ptr(means3D)
        inp.shs, inp.sh_objs, inp.colors_precomp = _lib.ptr(sh), _lib.ptr(sh_objs), _lib.ptr(colors_precomp)
        inp.opacities, inp.scales, inp.rotations = _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations)
        inp.cov3D_precomp = _lib.ptr(cov3Ds_precomp)
        out = _lib.RastOutputs()
        out.radii = _lib.ptr(radii)
        out.depth = _lib.ptr(depth_out)

        tmp = _bytes(_sizes(lib, P, W, H, F, ctx.capacity)[5], device)
        ws = _lib.RastWorkspace()
        ws.geom, ws.geom_bytes = _lib.ptr(geom), geom.numel()
        ws.bin, ws.bin_bytes = _lib.ptr(binb), binb.numel()
        ws.img, ws.img_bytes = _lib.ptr(img), img.numel()
        ws.pre, ws.pre_bytes = _lib.ptr(pre), pre.numel()
        ws.tmp, ws.tmp_bytes = _lib.ptr(tmp), tmp.numel()
        ws.capacity = ctx.capacity

        grad_image = _prep(grad_image, "grad_image", device)
        grad_feats = _prep(grad_feats, "grad_feats", device) if F > 0 else None
        grad_depth = _prep(grad_depth, "grad_depth", device)
        need = ctx.needs_input_grad   # order of forward()'s arguments

        def alloc(flag, present, *shape):
            return torch.empty(*shape, device=device) if (flag and present) else None

        # means3D / means2D gradients are always produced: the reference reads
        # viewspace_points.grad for densification (scene/gaussian_model.py:637-639)
        d_means3D = torch.empty(P, 3, device=device)
        d_means2D = torch.empty(P, 3, device=device)
        d_sh = alloc(need[2], sh is not None, P, M, 3)
        d_sh_objs = alloc(need[3], sh_objs is not None, *(sh_objs.shape if sh_objs is not None else (0,)))
        d_colors = alloc(need[4], colors_precomp is not None, P, 3)
        d_opac = alloc(need[5], True, *opacities.shape)
        d_scales = alloc(need[6], scales is not None, P, 3)
        d_rot = alloc(need[7], rotations is not None, P, 4)
        d_cov = alloc(need[8], cov3Ds_precomp is not None, P, 6)

        g = _lib.RastGrads()
        g.dL_dimage, g.dL_dfeats, g.dL_ddepth = _lib.ptr(grad_image), _lib.ptr(grad_feats), _lib.ptr(grad_depth)
        g.dL_dmeans3D, g.dL_dmeans2D = _lib.ptr(d_means3D), _lib.ptr(d_means2D)
        g.dL_dshs, g.dL_dsh_objs, g.dL_dcolors = _lib.ptr(d_sh), _lib.ptr(d_sh_objs), _lib.ptr(d_colors)
        g.dL_dopacities, g.dL_dscales, g.dL_drotations = _lib.ptr(d_opac), _lib.ptr(d_scales), _lib.ptr(d_rot)
        g.dL_dcov3D = _lib.ptr(d_cov)
        _lib.check(lib.trase_rast_backward(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws), C.byref(g),
                                           _stream(device)), "trase_rast_backward")
        if P == 0:
            for t in (d_means3D, d_means2D, d_sh, d_sh_objs, d_colors, d_opac, d_scales, d_rot, d_cov):
                if t is not None:
                    t.zero_()
        return (d_means3D if need[0] else None, d_means2D if need[1] else None, d_sh, d_sh_objs, d_colors,
                d_opac, d_scales, d_rot, d_cov, None)


def rasterize_gaussians(means3D, means2D, sh, sh_objs, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, sh_objs, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test of the lineage API (unused by TRASE): view-space z > 0.2."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix
            z = positions @ vm[:3, 2] + vm[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, sh_objs=None, colors_precomp=None,
                scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
           ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, sh_objs, colors_precomp, opacities, scales,
                                   rotations, cov3D_precomp, rs)


# --------------------------------------------------------------------------------------
# simple_knn._C.distCUDA2 (scene/gaussian_model.py:237)
# --------------------------------------------------------------------------------------
def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 runs on the GPU only (there is no CPU path)")
    pts = points.detach().float().contiguous()
    n = pts.shape[0]
    out = torch.empty(n, device=pts.device)
    nbytes = C.c_size_t()
    _lib.check(lib.trase_knn_sizes(n, C.byref(nbytes)), "trase_knn_sizes")
    ws = _bytes(nbytes.value, pts.device)
    dev = pts.device.index if pts.device.index is not None else torch.cuda.current_device()
    _lib.check(lib.trase_knn_dist2(_lib.ptr(pts), n, _lib.ptr(out), _lib.ptr(ws), ws.numel(), dev,
                                   _stream(pts.device)), "trase_knn_dist2")
    return out


# --------------------------------------------------------------------------------------
# pytorch3d.ops.knn_points (scene/gaussian_model.py:88-92, render.py:222, utils/loss_utils.py:141,192)
# --------------------------------------------------------------------------------------
class _KNN(NamedTuple):
    dists: torch.Tensor
    idx: torch.Tensor
    knn: Optional[torch.Tensor]


def knn_points(p1: torch.Tensor, p2: torch.Tensor, lengths1=None, lengths2=None, norm: int = 2, K: int = 1,
               version: int = -1, return_nn: bool = False, return_sorted: bool = True):
    """Same call contract as pytorch3d.ops.knn_points for the shapes TRASE uses: p1 (B,N1,3),
    p2 (B,N2,3) -> (dists (B,N1,K) squared L2, idx (B,N1,K) int64, knn).  K <= 16, L2 only, full lengths."""
    if norm != 2 or lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("knn_points: only norm=2 with full-length clouds is supported")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[0] != p2.shape[0] or p1.shape[2] != 3 or p2.shape[2] != 3:
        raise ValueError("knn_points expects p1 (B,N1,3) and p2 (B,N2,3)")
    if p1.device.type != "cuda":
        raise RuntimeError("knn_points runs on the GPU only (there is no CPU path)")
    lib = _lib.load()
    B, N1, N2 = p1.shape[0], p1.shape[1], p2.shape[1]
    dev = p1.device
    dists = torch.empty(B, N1, K, device=dev)
    idx = torch.empty(B, N1, K, dtype=torch.int64, device=dev)
    nbytes = C.c_size_t()
    _lib.check(lib.trase_knn_sizes(N2, C.byref(nbytes)), "trase_knn_sizes")
    ws = _bytes(nbytes.value, dev)
    d = dev.index if dev.index is not None else torch.cuda.current_device()
    for b in range(B):
        a, q = p1[b].detach().float().contiguous(), p2[b].detach().float().contiguous()
        _lib.check(lib.trase_knn_points(_lib.ptr(a), N1, _lib.ptr(q), N2, K, _lib.ptr(idx[b]), _lib.ptr(dists[b]),
                                        _lib.ptr(ws), ws.numel(), d, _stream(dev)), "trase_knn_points")
    knn = None
    if return_nn:
        knn = torch.gather(p2[:, None].expand(B, N1, N2, 3), 2, idx[..., None].expand(B, N1, K, 3))
    return _KNN(dists=dists, idx=idx, knn=knn)


# --------------------------------------------------------------------------------------
# per-kernel timing (HIP events on the launch stream) for bench.py's roofline leg
# --------------------------------------------------------------------------------------
def profile_enable(mode: int):
    """0 off, 1 every kernel, 2 only the compositing kernels (render_fwd / render_bwd)."""
    _lib.check(_lib.load().trase_prof_enable(int(mode)), "trase_prof_enable")


def profile_report() -> dict:
    import json
    buf = C.create_string_buffer(1 << 16)
    _lib.check(_lib.load().trase_prof_report(buf, len(buf)), "trase_prof_report")
    return json.loads(buf.value.decode())


def selftest(device=None) -> str:
    dev = torch.cuda.current_device() if device is None else device
    buf = C.create_string_buffer(4096)
    rc = _lib.load().trase_selftest(dev, _stream(torch.device("cuda", dev)), buf, len(buf))
    msg = buf.value.decode()
    if rc != 0:
        raise RuntimeError(f"trase_selftest failed ({rc}):\n{msg}\n{_lib.last_error()}")
    return msg
