"""``render()`` with the reference's signature and return dict (gaussian_renderer/__init__.py:37-155) on
top of the fused raw-parameter path: the A1 "prep" (activations, deformation add, SH concat, feature
L2-normalisation) runs inside the per-Gaussian HIP kernels instead of ~25 small PyTorch kernels per
iteration.  Importable as ``gaussian_renderer.render`` through the top-level shim,
so train.py / train_style_transfer_nnfm.py / render.py keep their ``from gaussian_renderer import render``.

Anything the fused kernels do not cover (is_6dof, override_color, mask, the Python SH / covariance
fallbacks, KNN-smoothed features) takes the reference's own composition of PyTorch ops around the HIP
``GaussianRasterizer`` -- still the HIP rasterizer, never a CPU path."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .sh import sh_colors_python
from .smooth import smoothed_gaussian_features
from .rasterizer import VARIANT_SPARSE_STRIP_GRADS as _r_VARIANT_SPARSE
from .rasterizer import VARIANT_DEPTH32 as _r_VARIANT_DEPTH32
from .rasterizer import VARIANT_FORWARD_ONLY as _r_VARIANT_FORWARD_ONLY
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, _Policy, _after_render, _release_last, _bytes, _fill_settings,
                         _output_maps, _pick_capacity, _prep, _sizes, _stream)


def set_backward_scope(scope: str = "all") -> None:
    """``"features"``: the backward produces only dL/d(gaussian features) -- everything else (positions, covariances,
    opacities, colours, and the screen-space gradient the densifier reads) comes back as zeros.  For FEATURE-state
    iterations once densification has ended (``iteration >= opt.densify_until_iter``): no other parameter requires a
    gradient then (scene/gaussian_model.py:303-315) and ``viewspace_point_tensor.grad`` is no longer read
    (train.py:361-366).  The blend-weight part of the backward (transmittance scan + the weights x cotangent GEMM) stays;
    the channel contraction for dL/dalpha, its scan and the moment sums are skipped.  ``"all"`` (default) restores the
    reference's behaviour."""
    from . import rasterizer as _r
    if scope not in ("all", "features"):
        raise ValueError("scope must be 'all' or 'features'")
    v = _r._Policy.variant & ~_r.VARIANT_FEATURES_ONLY_BWD
    _r.set_variant(v | (_r.VARIANT_FEATURES_ONLY_BWD if scope == "features" else 0))


_FORWARD_SCOPE = "all"


def set_forward_scope(scope: str = "all") -> None:
    """``"image"``: the fused ``render()`` composites colour and depth only -- ``render_gaussian_features`` comes back as an
    empty (0, H, W) tensor.  For GAUSSIAN-state iterations: train.py:211 reads ``render``, ``viewspace_points``,
    ``visibility_filter`` and ``radii`` of the returned dict and nothing else, while the rasterizer still blends the 32
    feature channels per pixel (0.28 -> 0.16 ms forward at 300k Gaussians / 1080p).  ``"all"`` (default) restores the
    reference's behaviour.  Opt-in, like ``set_backward_scope``; the operator-level ``GaussianRasterizer`` is unaffected."""
    global _FORWARD_SCOPE
    if scope not in ("all", "image"):
        raise ValueError("scope must be 'all' or 'image'")
    _FORWARD_SCOPE = scope


# sparse strip gradients (rasterizer.set_sparse_strip_grads): persistent gradient tensors, zero outside the rows the most recent
# strip backward wrote; `prev` = (geom, pre) workspaces of the forward whose backward wrote them (its live-Gaussian list)
# `where` = (P, device) every cached buffer was made for; `fwd_seq` = number of the most recent strip forward that asked for
# sparse gradients (a backward of an OLDER forward allocates densely: the persistent buffers belong to the newest one)
_SPARSE: dict = {"bufs": {}, "prev": None, "where": None, "fwd_seq": 0}


def _sparse_buf(name: str, like: torch.Tensor) -> torch.Tensor:
    where = (int(like.shape[0]), like.device)
    if _SPARSE["where"] != where:
        # the parameter set changed (densify / prune) or moved: EVERY cached buffer is for the old Gaussian count, also the
        # ones this backward does not ask for -- the row-clearing pass takes all of them and indexes rows up to the new P
        # (an out-of-bounds device write when only the same-named buffer was dropped)
        _SPARSE["bufs"], _SPARSE["prev"], _SPARSE["where"] = {}, None, where
    b = _SPARSE["bufs"].get(name)
    if b is None or b.shape != like.shape:
        if b is not None:                                     # same P, another trailing shape (feature width changed)
            _SPARSE["prev"] = None
        b = _SPARSE["bufs"][name] = torch.zeros_like(like)
    return b


_GRAD_SINK: dict = {}
_GRAD_CHUNKS = None        # (number of Gaussian-index chunks, callback(p_begin, p_end, P)) of the overlapped exchange


def set_grad_sink(sink=None, chunks: int = 1, on_chunk=None) -> None:
    """``sink``: {id(parameter): (weakref(parameter), preallocated gradient buffer of the parameter's shape)} or None
    (``trase_amd.dp.FlatGradBucket.sink()``).  The fused backward then writes the gradients of those parameters directly
    into the given buffers (and autograd adopts them as ``.grad`` when ``.grad`` is None) instead of allocating fresh
    tensors, so that the view-parallel all-reduce bucket is filled without a zero-fill and an accumulation pass.  A
    parameter is recognised by object identity (checked through the weak reference), never by its address.

    ``chunks`` > 1 with ``on_chunk``: the per-Gaussian tail of the backward (row reduction + activation chain) runs in
    that many Gaussian-index ranges and ``on_chunk(p_begin, p_end, P, ids)`` (``ids`` = the parameters whose sink buffers this
    backward writes) is called after each range has been ENQUEUED on
    the current stream -- every gradient entry of the Gaussians [p_begin, p_end) is final in stream order at that point,
    so the caller can start exchanging them while the remaining ranges are still being computed
    (``trase_amd.dp.FlatGradBucket.overlapped``)."""
    global _GRAD_SINK, _GRAD_CHUNKS
    _GRAD_SINK = dict(sink) if sink else {}
    _GRAD_CHUNKS = (int(chunks), on_chunk) if (sink and on_chunk is not None and int(chunks) > 1) else None


def chunk_ranges(P: int, chunks: int):
    """[(begin, end)] partition of [0, P) into at most ``chunks`` ranges whose starts are multiples of 64 (what
    ``trase_rast_backward_raw_gaussians`` accepts), sizes within 64 of each other."""
    blocks = (int(P) + 63) // 64
    k = max(1, min(int(chunks), blocks))
    base, extra = divmod(blocks, k)
    out, b = [], 0
    for c in range(k):
        e = b + base + (1 if c < extra else 0)
        out.append((b * 64, min(e * 64, int(P))))
        b = e
    return out


class _RenderRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, d_xyz, f_dc, f_rest, opacity, scaling, d_scaling, rotation, d_rotation, gfeat, means2D,
                raster_settings, norm_features, override_color=None, mask=None, se3=None, sh_dir_raw=False, fwd_only=False):
        # override_color (P,3) / mask (P) bool / se3 (P,4,4): render()'s inference call patterns (gaussian_renderer/__init__.py:75-80,
        # :112-113, :123-135), sh_dir_raw: pipe.convert_SHs_python (:103-108) -- all inside the per-Gaussian kernels (round 6)
        lib = _lib.load()
        device = xyz.device
        if device.type != "cuda":
            raise RuntimeError("trase_amd render runs on the GPU only (there is no CPU path)")
        _release_last()
        param_ids = dict(xyz=id(xyz), f_dc=id(f_dc), f_rest=id(f_rest), opacity=id(opacity), scaling=id(scaling),
                         rotation=id(rotation), gfeat=id(gfeat))
        T = lambda t, n: _prep(t, n, device)
        xyz, f_dc, f_rest, opacity = T(xyz, "xyz"), T(f_dc, "features_dc"), T(f_rest, "features_rest"), T(opacity, "opacity")
        scaling, rotation = T(scaling, "scaling"), T(rotation, "rotation")
        d_xyz, d_scaling, d_rotation = T(d_xyz, "d_xyz"), T(d_scaling, "d_scaling"), T(d_rotation, "d_rotation")
        gfeat = T(gfeat, "gaussian_features")
        override_color, se3 = T(override_color, "override_color"), T(se3, "d_xyz (is_6dof)")
        P = xyz.shape[0]
        F = gfeat.shape[-1] if gfeat is not None else 0
        if f_rest.shape[1] != 15 or f_dc.shape[1] != 1:
            raise ValueError("fused render expects features_dc (P,1,3) and features_rest (P,15,3)")
        if override_color is not None and tuple(override_color.shape) != (P, 3):
            raise ValueError(f"override_color must be ({P}, 3), got {tuple(override_color.shape)}")
        if se3 is not None and tuple(se3.shape) != (P, 4, 4):
            raise ValueError(f"is_6dof: d_xyz must be ({P}, 4, 4), got {tuple(se3.shape)}")
        mask_u8 = None
        if mask is not None:
            if mask.dtype != torch.bool or tuple(mask.shape) != (P,) or mask.device != device:
                raise ValueError(f"mask must be a bool tensor of shape ({P},) on {device}")
            mask_u8 = mask.contiguous().view(torch.uint8)
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        keep: list = []
        s = _fill_settings(raster_settings, device, keep)
        featn = torch.empty(P, max(F, 1), device=device)
        raw = _lib.RastRawInputs()
        raw.P, raw.F, raw.norm_features = P, F, int(bool(norm_features))
        raw.xyz, raw.d_xyz = _lib.ptr(xyz), _lib.ptr(d_xyz)
        raw.features_dc, raw.features_rest, raw.opacity = _lib.ptr(f_dc), _lib.ptr(f_rest), _lib.ptr(opacity)
        raw.scaling, raw.d_scaling = _lib.ptr(scaling), _lib.ptr(d_scaling)
        raw.rotation, raw.d_rotation = _lib.ptr(rotation), _lib.ptr(d_rotation)
        raw.gaussian_features = _lib.ptr(gfeat) if F > 0 else None
        raw.featn = _lib.ptr(featn)
        raw.colors_precomp, raw.mask, raw.d_xyz_se3 = _lib.ptr(override_color), _lib.ptr(mask_u8), _lib.ptr(se3)
        raw.sh_dir_undeformed = int(bool(sh_dir_raw))
        # fwd_only: render() was called under torch.no_grad() -- nobody will differentiate this forward, and the state only a
        # backward reads is not written (TRASE_VARIANT_FORWARD_ONLY).  (Decided by the caller: inside an autograd Function's
        # forward the grad mode is always off.)
        if fwd_only:
            s.variant |= _r_VARIANT_FORWARD_ONLY

        image, feats, depth = _output_maps(F, H, W, device, bool(s.tile_row_begin or s.tile_row_end))
        radii = torch.empty(P, dtype=torch.int32, device=device)
        out = _lib.RastOutputs()
        out.image, out.radii, out.depth = _lib.ptr(image), _lib.ptr(radii), _lib.ptr(depth)
        out.feats = _lib.ptr(feats) if F > 0 else None
        geom_b, _, img_b, pre_b, _, _ = _sizes(lib, P, W, H, F, 1)
        geom, pre, img = _bytes(geom_b, device), _bytes(pre_b, device), _bytes(img_b, device)
        ws = _lib.RastWorkspace()
        ws.geom, ws.geom_bytes = _lib.ptr(geom), geom.numel()
        ws.pre, ws.pre_bytes = _lib.ptr(pre), pre.numel()
        ws.img, ws.img_bytes = _lib.ptr(img), img.numel()
        stream = _stream(device)
        one_call = (not _Policy.sync) and _Policy.capacity > 0      # capacity known beforehand: one boundary crossing
        if not one_call:
            _lib.check(lib.trase_rast_preprocess_raw(C.byref(s), C.byref(raw), C.byref(out), C.byref(ws), stream),
                       "trase_rast_preprocess_raw")
        def _again():              # a saturated 27-bit depth key: stage 1 once more on the raw float bits
            s.variant |= _r_VARIANT_DEPTH32
            _lib.check(lib.trase_rast_preprocess_raw(C.byref(s), C.byref(raw), C.byref(out), C.byref(ws), stream),
                       "trase_rast_preprocess_raw")
        capacity = _pick_capacity(lib, ws, stream, None if one_call else _again)
        _, bin_b, _, _, tmp_b, _ = _sizes(lib, P, W, H, F, capacity)
        binb, tmp = _bytes(bin_b, device), _bytes(tmp_b, device)
        ws.bin, ws.bin_bytes = _lib.ptr(binb), binb.numel()
        ws.tmp, ws.tmp_bytes = _lib.ptr(tmp), tmp.numel()
        ws.capacity = capacity
        if _PAIR is not None:
            # render_views(): the launch sequence of this view is issued together with the next view's (one depth sort for both);
            # the record keeps every buffer the argument structs point at alive until then
            if not one_call or s.tile_row_begin or s.tile_row_end:
                raise RuntimeError("trase_amd.renderer.render_views needs the sync-free capacity policy with a known capacity "
                                   "(rasterizer.set_sync(False, capacity=...)) and whole-image views")
            _PAIR.append(dict(s=s, raw=raw, out=out, ws=ws, P=P, device=device, stream=stream, after=(geom, capacity, binb, (H, W)),
                              keep=(keep, xyz, d_xyz, f_dc, f_rest, opacity, scaling, d_scaling, rotation, d_rotation, gfeat, featn,
                                    image, feats, depth, radii, geom, pre, img, binb, tmp)))
        elif one_call:
            _lib.check(lib.trase_rast_forward_raw(C.byref(s), C.byref(raw), C.byref(out), C.byref(ws), stream),
                       "trase_rast_forward_raw")
        else:
            _lib.check(lib.trase_rast_render_raw(C.byref(s), C.byref(raw), C.byref(out), C.byref(ws), stream),
                       "trase_rast_render_raw")
        if _PAIR is None:
            _after_render(geom, capacity, binb, (H, W))
        ctx.raster_settings, ctx.capacity, ctx.dims = raster_settings, capacity, (P, F, H, W)
        ctx.variant, ctx.tile_rows, ctx.feat_bg = s.variant, (s.tile_row_begin, s.tile_row_end), s.feat_bg
        if (s.variant & _r_VARIANT_SPARSE) and (s.tile_row_begin != 0 or s.tile_row_end != 0):
            _SPARSE["fwd_seq"] += 1
            ctx.sparse_seq = _SPARSE["fwd_seq"]
        ctx.param_ids = param_ids              # which parameter OBJECTS the gradients belong to (grad-sink lookup)
        ctx.norm_features = bool(norm_features)
        ctx.opt = (d_xyz is not None, d_scaling is not None, d_rotation is not None, gfeat is not None)
        ctx.extra = (override_color is not None, mask_u8 is not None, se3 is not None, bool(sh_dir_raw))
        ctx.set_materialize_grads(False)
        z = xyz.new_empty(0)
        # under `mask` the reference hands the rasterizer the SUBSET (gaussian_renderer/__init__.py:123-135) and gets the subset's
        # radii back: the dict's `radii` / `visibility_filter` have mask.sum() entries.  The kernels keep full-size arrays (a
        # removed Gaussian has radius 0); the subset is taken here -- the one synchronising op of this path (the reference's
        # seven boolean indexings synchronise seven times).
        radii_out = radii[mask] if mask is not None else radii
        ctx.mark_non_differentiable(radii_out)
        ctx.save_for_backward(xyz, d_xyz if d_xyz is not None else z, f_dc, f_rest, opacity, scaling,
                              d_scaling if d_scaling is not None else z, rotation,
                              d_rotation if d_rotation is not None else z, gfeat if gfeat is not None else z,
                              radii, geom, binb, img, pre, featn,
                              depth if (s.variant & 0x20100) == 0x20100 else None,   # normalised depth + depth gradient
                              override_color if override_color is not None else z,
                              mask_u8 if mask_u8 is not None else z, se3 if se3 is not None else z)
        return image, radii_out, feats, depth

    @staticmethod
    def backward(ctx, grad_image, grad_radii, grad_feats, grad_depth):
        lib = _lib.load()
        (xyz, d_xyz, f_dc, f_rest, opacity, scaling, d_scaling, rotation, d_rotation, gfeat,
         radii, geom, binb, img, pre, featn, depth_out, override_color, mask_u8, se3) = ctx.saved_tensors
        has_dxyz, has_dscale, has_drot, has_feat = ctx.opt
        has_color, has_mask, has_se3, sh_dir_raw = ctx.extra
        P, F, H, W = ctx.dims
        device = xyz.device
        keep: list = []
        s = _fill_settings(ctx.raster_settings, device, keep)
        s.variant = ctx.variant                # the forward's variant and strip, not whatever the globals say now
        s.tile_row_begin, s.tile_row_end = ctx.tile_rows
        s.feat_bg = ctx.feat_bg
        raw = _lib.RastRawInputs()
        raw.P, raw.F, raw.norm_features = P, F, int(ctx.norm_features)
        raw.xyz, raw.d_xyz = _lib.ptr(xyz), (_lib.ptr(d_xyz) if has_dxyz else None)
        raw.features_dc, raw.features_rest, raw.opacity = _lib.ptr(f_dc), _lib.ptr(f_rest), _lib.ptr(opacity)
        raw.scaling, raw.d_scaling = _lib.ptr(scaling), (_lib.ptr(d_scaling) if has_dscale else None)
        raw.rotation, raw.d_rotation = _lib.ptr(rotation), (_lib.ptr(d_rotation) if has_drot else None)
        raw.gaussian_features = _lib.ptr(gfeat) if (has_feat and F > 0) else None
        raw.featn = _lib.ptr(featn)
        raw.colors_precomp = _lib.ptr(override_color) if has_color else None
        raw.mask = _lib.ptr(mask_u8) if has_mask else None
        raw.d_xyz_se3 = _lib.ptr(se3) if has_se3 else None
        raw.sh_dir_undeformed = int(sh_dir_raw)
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
        need = ctx.needs_input_grad   # xyz0 d_xyz1 f_dc2 f_rest3 opacity4 scaling5 d_scaling6 rotation7 d_rotation8 gfeat9 means2D10

        pid = ctx.param_ids
        used_sink: set = set()

        sparse = bool(s.variant & _r_VARIANT_SPARSE) and (s.tile_row_begin != 0 or s.tile_row_end != 0) and _GRAD_CHUNKS is None and P > 0
        if sparse and getattr(ctx, "sparse_seq", -1) != _SPARSE["fwd_seq"]:
            # another strip forward has run since this one: the persistent buffers (and the rows to clear) belong to ITS
            # backward; handing them out here would overwrite gradients autograd has not consumed yet.  Dense tensors instead.
            sparse = False
        leaf_of = {"xyz": xyz, "f_dc": f_dc, "f_rest": f_rest, "opacity": opacity, "scaling": scaling, "rotation": rotation,
                   "gfeat": gfeat}
        sparse_used: dict = {}

        def alloc(flag, like, name=None, sname=None):
            if not flag:
                return None
            if sparse and sname:
                b = sparse_used[sname] = _sparse_buf(sname, like)
                leaf = leaf_of.get(sname)
                if leaf is not None and leaf.grad is not None and leaf.grad.data_ptr() == b.data_ptr():
                    raise RuntimeError(
                        f"trase_amd: sparse strip gradients: the .grad of `{sname}` still IS the persistent gradient buffer of the "
                        "previous strip backward (kept, or zeroed in place).  This backward clears and rewrites that buffer, and "
                        "autograd would then add it to itself.  Drop the gradients between backwards "
                        "(`p.grad = None` / `zero_grad(set_to_none=True)`), clone what you keep, or accumulate with "
                        "set_sparse_strip_grads(False).")
                return b.view(b.shape)
            ent = _GRAD_SINK.get(pid[name]) if (_GRAD_SINK and name) else None
            if ent is not None:
                ref, buf = ent
                p = ref()
                if p is not None and id(p) == pid[name] and buf.shape == like.shape and buf.device == like.device:
                    if p.grad is not None and p.grad.data_ptr() == buf.data_ptr():
                        # .grad already IS the sink buffer (a FlatGradBucket hands its slices out as .grad when it is built, and
                        # bucket.zero() keeps them there): autograd will ACCUMULATE into it, and handing the same bytes out as
                        # the incoming gradient would add the buffer to itself -- silently doubled gradients (round 5, found by
                        # a combination sweep).  A fresh tensor instead: the sum lands in the bucket through AccumulateGrad.
                        return torch.empty_like(like)
                    used_sink.add(pid[name])
                    # a FRESH view object: autograd's AccumulateGrad then adopts it as .grad without a copy (it clones a
                    # gradient that somebody else still references), so the gradient is written once, in place, into the
                    # caller's buffer (trase_amd.dp.FlatGradBucket: the all-reduce bucket)
                    return buf.view(buf.shape)
            return torch.empty_like(like)

        # the kernel always produces dL/dxyz (the position chain needs it); it only lands in the sink when it is asked for
        g_xyz = alloc(True, xyz, "xyz" if need[0] else None, "xyz")
        g_dxyz = alloc(need[1] and has_dxyz, xyz, None, "dxyz")    # the deformation offsets are not bucket parameters
        g_m2d = alloc(True, xyz, None, "m2d") if sparse else torch.empty(P, 3, device=device)
        g_dc, g_rest = alloc(need[2] and not has_color, f_dc, "f_dc", "f_dc"), alloc(need[3] and not has_color, f_rest, "f_rest", "f_rest")
        g_color = torch.empty_like(override_color) if (has_color and need[13]) else None
        g_se3 = torch.empty_like(se3) if (has_se3 and need[15]) else None
        g_op, g_sc, g_rot = (alloc(need[4], opacity, "opacity", "opacity"), alloc(need[5], scaling, "scaling", "scaling"),
                             alloc(need[6 + 1], rotation, "rotation", "rotation"))
        g_dsc = alloc(need[6] and has_dscale, scaling, None, "dscaling")
        g_drot = alloc(need[8] and has_drot, rotation, None, "drotation")
        g_feat = alloc(need[9] and has_feat and F > 0, gfeat, "gfeat", "gfeat") if has_feat else None
        g = _lib.RastRawGrads()
        g.dL_dimage = _lib.ptr(_prep(grad_image, "grad_image", device))
        g.dL_dfeats = _lib.ptr(_prep(grad_feats, "grad_feats", device)) if F > 0 else None
        g.dL_ddepth = _lib.ptr(_prep(grad_depth, "grad_depth", device))
        g.dL_dxyz, g.dL_dd_xyz, g.dL_dmeans2D = _lib.ptr(g_xyz), _lib.ptr(g_dxyz), _lib.ptr(g_m2d)
        g.dL_dfeatures_dc, g.dL_dfeatures_rest, g.dL_dopacity = _lib.ptr(g_dc), _lib.ptr(g_rest), _lib.ptr(g_op)
        g.dL_dscaling, g.dL_dd_scaling = _lib.ptr(g_sc), _lib.ptr(g_dsc)
        g.dL_drotation, g.dL_dd_rotation = _lib.ptr(g_rot), _lib.ptr(g_drot)
        g.dL_dgaussian_features = _lib.ptr(g_feat)
        g.dL_dcolors_precomp, g.dL_dd_xyz_se3 = _lib.ptr(g_color), _lib.ptr(g_se3)
        if sparse:
            # the rows the PREVIOUS strip backward wrote go back to zero first (every buffer of the cache, whether or not this
            # backward uses it); then this one writes the rows of its own live Gaussians
            prev = _SPARSE["prev"]
            if prev is not None and prev[2] == P:
                pgeom, ppre, _, pF = prev
                zg = _lib.RastRawGrads()
                by = {k: v for k, v in _SPARSE["bufs"].items() if v.shape[0] == P}
                zg.dL_dxyz, zg.dL_dd_xyz, zg.dL_dmeans2D = _lib.ptr(by.get("xyz")), _lib.ptr(by.get("dxyz")), _lib.ptr(by.get("m2d"))
                zg.dL_dfeatures_dc, zg.dL_dfeatures_rest, zg.dL_dopacity = _lib.ptr(by.get("f_dc")), _lib.ptr(by.get("f_rest")), _lib.ptr(by.get("opacity"))
                zg.dL_dscaling, zg.dL_dd_scaling = _lib.ptr(by.get("scaling")), _lib.ptr(by.get("dscaling"))
                zg.dL_drotation, zg.dL_dd_rotation = _lib.ptr(by.get("rotation")), _lib.ptr(by.get("drotation"))
                zg.dL_dgaussian_features = _lib.ptr(by.get("gfeat"))
                zraw = _lib.RastRawInputs()
                zraw.P, zraw.F = P, (by["gfeat"].shape[-1] if by.get("gfeat") is not None else 0)
                zws = _lib.RastWorkspace()
                zws.geom, zws.geom_bytes = _lib.ptr(pgeom), pgeom.numel()
                zws.pre, zws.pre_bytes = _lib.ptr(ppre), ppre.numel()
                _lib.check(lib.trase_rast_zero_live_rows(C.byref(s), C.byref(zraw), C.byref(zws), C.byref(zg), _stream(device)),
                           "trase_rast_zero_live_rows")
            _SPARSE["prev"] = (geom, pre, P, F)
        if _GRAD_CHUNKS is not None and P > 0:
            # compositing backward once, then the per-Gaussian tail range by range: the caller's hook sees every range as
            # soon as it is in the stream (the view-parallel exchange of that range overlaps the rest of the tail)
            n_chunks, hook = _GRAD_CHUNKS
            _lib.check(lib.trase_rast_backward_raw_compose(C.byref(s), C.byref(raw), C.byref(out), C.byref(ws), C.byref(g),
                                                           _stream(device)), "trase_rast_backward_raw_compose")
            for (pb, pe) in chunk_ranges(P, n_chunks):
                _lib.check(lib.trase_rast_backward_raw_gaussians(C.byref(s), C.byref(raw), C.byref(out), C.byref(ws),
                                                                 C.byref(g), pb, pe, _stream(device)),
                           "trase_rast_backward_raw_gaussians")
                hook(pb, pe, P, used_sink)
        else:
            _lib.check(lib.trase_rast_backward_raw(C.byref(s), C.byref(raw), C.byref(out), C.byref(ws), C.byref(g),
                                                   _stream(device)), "trase_rast_backward_raw")
        if P == 0:
            for t in (g_xyz, g_dxyz, g_m2d, g_dc, g_rest, g_op, g_sc, g_dsc, g_rot, g_drot, g_feat, g_color, g_se3):
                if t is not None:
                    t.zero_()
        return (g_xyz if need[0] else None, g_dxyz, g_dc, g_rest, g_op, g_sc, g_dsc, g_rot, g_drot, g_feat,
                g_m2d if need[10] else None, None, None, g_color, None, g_se3, None, None)


# ---- two views per launch sequence ---------------------------------------------------------------------------------------
_PAIR = None        # render_views(): list of deferred forward records


def _flush_pair(recs):
    lib = _lib.load()
    if len(recs) == 2 and recs[0]["P"] == recs[1]["P"] and recs[0]["device"] == recs[1]["device"] and recs[0]["P"] > 0:
        a, b = recs
        nbytes = C.c_size_t()
        _lib.check(lib.trase_rast_pair_sizes(a["P"], C.byref(nbytes)), "trase_rast_pair_sizes")
        pair_ws = _bytes(nbytes.value, a["device"])
        _lib.check(lib.trase_rast_forward_raw_pair(C.byref(a["s"]), C.byref(a["raw"]), C.byref(a["out"]), C.byref(a["ws"]),
                                                   C.byref(b["s"]), C.byref(b["raw"]), C.byref(b["out"]), C.byref(b["ws"]),
                                                   _lib.ptr(pair_ws), pair_ws.numel(), a["stream"]), "trase_rast_forward_raw_pair")
    else:
        for r in recs:
            _lib.check(lib.trase_rast_forward_raw(C.byref(r["s"]), C.byref(r["raw"]), C.byref(r["out"]), C.byref(r["ws"]), r["stream"]),
                       "trase_rast_forward_raw")
    for r in recs:
        _after_render(*r["after"])


def render_views(viewpoint_cameras, pc, pipe, bg_color, d_xyz, d_rotation, d_scaling, **kwargs):
    """``[render(cam, pc, pipe, bg, d_xyz, d_rotation, d_scaling, **kwargs) for cam in viewpoint_cameras]`` with the views taken two
    at a time through ONE launch sequence (``trase_rast_forward_raw_pair``: the two views' depth sorts -- the latency-bound
    part of a view -- are one sort).  No counterpart in the reference (train.py:180 renders one view per iteration); a loop
    that accumulates the losses of two views per optimizer step (or an evaluation sweep) can call this instead of two
    ``render()``s.  Every entry of the returned list is an ordinary ``render()`` dict with its own autograd history: outputs
    and gradients are bit-identical to the serial calls (``tests/test_gpu_pair.py``).  ``d_xyz`` / ``d_rotation`` /
    ``d_scaling``: one value for all views, or a list with one entry per view (each view's deformation at its own time).
    Needs ``rasterizer.set_sync(False, capacity=...)`` (the workspaces of both views are sized before anything runs); views
    that cannot take the fused path, and an odd last view, go through ``render()`` as they are."""
    global _PAIR
    cams = list(viewpoint_cameras)
    per = lambda d, i: d[i] if isinstance(d, (list, tuple)) else d
    if kwargs.get("mask") is not None:       # (the subset's radii are taken right behind the forward: nothing to defer)
        return [render(c, pc, pipe, bg_color, per(d_xyz, i), per(d_rotation, i), per(d_scaling, i), **kwargs) for i, c in enumerate(cams)]
    outs = []
    for i in range(0, len(cams), 2):
        if i + 1 >= len(cams):
            outs.append(render(cams[i], pc, pipe, bg_color, per(d_xyz, i), per(d_rotation, i), per(d_scaling, i), **kwargs))
            break
        if _PAIR is not None:
            raise RuntimeError("render_views is not re-entrant")
        _PAIR = []
        try:
            pair = [render(cams[j], pc, pipe, bg_color, per(d_xyz, j), per(d_rotation, j), per(d_scaling, j), **kwargs) for j in (i, i + 1)]
            recs, _PAIR = _PAIR, None
            _flush_pair(recs)
            # render() formed `radii > 0` while the forward was only RECORDED (radii still unwritten: the compare ran ahead of the
            # kernels that fill it -- ADVICE r5); now that the launch sequence is in the stream, form it again
            for o in pair:
                o["visibility_filter"] = o["radii"] > 0
        finally:
            _PAIR = None
        outs.extend(pair)
    return outs


_ZERO_CACHE: dict = {}


def _zero_dummy(like: torch.Tensor) -> torch.Tensor:
    key = (tuple(like.shape), like.dtype, like.device)
    base = _ZERO_CACHE.get(key)
    if base is None:
        if len(_ZERO_CACHE) > 8:
            _ZERO_CACHE.clear()
        base = _ZERO_CACHE[key] = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
    return base.view(base.shape).requires_grad_(True)


def _fusable(pc, pipe, d_xyz, d_rotation, d_scaling, is_6dof, override_color, mask, is_smooth) -> bool:
    """Every call pattern of the reference's render() takes the fused raw-parameter path (round 6: override_color, mask, is_6dof and
    the two Python fallbacks of the pipe included); what is left for the operator-level composition are malformed deformation
    arguments and SH layouts other than (P,1,3) + (P,15,3)."""
    N = pc._xyz.shape[0]
    if is_6dof:
        if torch.is_tensor(d_xyz) and tuple(d_xyz.shape) != (N, 4, 4):
            return False
    elif torch.is_tensor(d_xyz) and d_xyz.dim() != 2:
        return False
    elif not torch.is_tensor(d_xyz) and float(d_xyz) != 0.0:
        return False
    for d in (d_rotation, d_scaling):
        if torch.is_tensor(d) and d.dim() != 2:
            return False
        if not torch.is_tensor(d) and float(d) != 0.0:
            return False
    if override_color is not None and not (torch.is_tensor(override_color) and tuple(override_color.shape) == (N, 3)):
        return False
    if mask is not None and not (torch.is_tensor(mask) and mask.dtype == torch.bool and tuple(mask.shape) == (N,)):
        return False
    return pc._features_rest.shape[1] == 15 and pc._features_dc.shape[1] == 1


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, d_xyz, d_rotation, d_scaling, is_6dof=False,
           scaling_modifier=1.0, override_color=None, mask=None, norm_gaussian_features=True,
           is_smooth_gaussian_features=False, smooth_K=16):
    """Same contract as the reference's render() (gaussian_renderer/__init__.py:37-155)."""
    xyz = pc.get_xyz
    # The reference builds `zeros_like(xyz, requires_grad=True) + 0` (gaussian_renderer/__init__.py:48-52): a dummy whose VALUES
    # nobody reads -- the rasterizer returns the screen-space gradient through it and train.py reads `.grad`.  Here: a fresh
    # LEAF that aliases a cached block of zeros -- same values, `.grad` filled by autograd, and the fill + add kernels of
    # every view (~7 us at 300k Gaussians, launch-bound) are gone.
    screenspace_points = _zero_dummy(xyz)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False,
        debug=getattr(pipe, "debug", False))

    if _fusable(pc, pipe, d_xyz, d_rotation, d_scaling, is_6dof, override_color, mask, is_smooth_gaussian_features):
        T = lambda d: d if torch.is_tensor(d) else None
        # KNN-smoothed features (FEATURE state, gaussian_renderer/__init__.py:118): one HIP gather, then the fused path
        if _FORWARD_SCOPE == "image":
            gfeat = None                       # colour + depth only (set_forward_scope)
        else:
            gfeat = smoothed_gaussian_features(pc, K=smooth_K, dropout=0.5) if is_smooth_gaussian_features \
                else pc._gaussian_features
        cov_python = bool(getattr(pipe, "compute_cov3D_python", False))
        # pipe.compute_cov3D_python (gaussian_renderer/__init__.py:93-94): pc.get_covariance(scaling_modifier) is built from the
        # UNDEFORMED scaling and rotation (scene/gaussian_model.py:216-217) -- the kernels' own Sigma = R S S R^T of exp(_scaling) *
        # modifier and normalize(_rotation), i.e. the fused path with the two deformation terms left out
        # pipe.convert_SHs_python (:103-108): clamp_min(eval_sh(...) + 0.5, 0) in the direction of the UNDEFORMED position
        sh_py = bool(getattr(pipe, "convert_SHs_python", False)) and override_color is None
        rendered_image, radii, rendered_feats, depth = _RenderRaw.apply(
            pc._xyz, None if is_6dof else T(d_xyz), pc._features_dc, pc._features_rest, pc._opacity, pc._scaling,
            None if cov_python else T(d_scaling), pc._rotation, None if cov_python else T(d_rotation), gfeat, screenspace_points,
            raster_settings, norm_gaussian_features, override_color, mask, T(d_xyz) if is_6dof else None, sh_py,
            not torch.is_grad_enabled())
    else:
        # the reference's own composition around the (HIP) rasterizer
        rasterizer = GaussianRasterizer(raster_settings=raster_settings)
        if is_6dof:
            if torch.is_tensor(d_xyz) is False:
                means3D = pc.get_xyz
            else:
                ones = torch.ones_like(pc.get_xyz[:, :1])
                hom = torch.cat([pc.get_xyz, ones], dim=-1)
                res = torch.bmm(d_xyz, hom.unsqueeze(-1)).squeeze(-1)
                means3D = res[:, :3] / res[:, 3:]
        else:
            means3D = pc.get_xyz + d_xyz
        means2D, opacity = screenspace_points, pc.get_opacity
        scales = rotations = cov3D_precomp = None
        if getattr(pipe, "compute_cov3D_python", False):
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales = pc.get_scaling + d_scaling
            rotations = pc.get_rotation + d_rotation
        shs = colors_precomp = None
        if override_color is None:
            if getattr(pipe, "convert_SHs_python", False):
                colors_precomp = sh_colors_python(pc, viewpoint_camera.camera_center)
            else:
                shs = pc.get_features
        else:
            colors_precomp = override_color
        sh_objs = pc.get_gaussian_features if not is_smooth_gaussian_features else \
            smoothed_gaussian_features(pc, K=smooth_K, dropout=0.5)
        if norm_gaussian_features:
            sh_objs = sh_objs / (sh_objs.norm(dim=2, keepdim=True) + 1e-9)
        if mask is not None:
            means3D, means2D, opacity, sh_objs = means3D[mask], means2D[mask], opacity[mask], sh_objs[mask]
            if colors_precomp is not None:
                colors_precomp = colors_precomp[mask]
            else:
                shs = shs[mask]
            scales, rotations = scales[mask], rotations[mask]
            if cov3D_precomp is not None:
                cov3D_precomp = cov3D_precomp[mask]
        rendered_image, radii, rendered_feats, depth = rasterizer(
            means3D=means3D, means2D=means2D, shs=shs, sh_objs=sh_objs, colors_precomp=colors_precomp,
            opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    # (render_views: the forward is deferred, `radii` not yet written -- the filter is formed after the flush)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": (radii > 0) if _PAIR is None else None,
            "radii": radii, "render_gaussian_features": rendered_feats, "depth": depth}
