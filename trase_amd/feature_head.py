"""The FEATURE-state loss head of train.py:251-296 without any S x S matrix.

The reference builds, per iteration, ``C_matrix`` (utils/feature_utils.py:40-49), ``C_F_matrix`` (:51-57) and
``pixel_weights`` (:28-38, through an N x H x W int64 temporary), all S x S with S ~ 5000, feeds them to the two pair
losses (utils/loss_utils.py:275-406) and indexes them again for the two logged similarities (train.py:295-296): about
10 ms of PyTorch around a 1.6 ms rasterizer.  Every entry of those matrices is a function of per-pixel factors, so the
kernels of ``csrc/pairhead.hip`` evaluate them on the fly:

    cover, sizes   = mask_stats(sam_masks)                                     # one pass over the masks
    sampled_pixel, sampled_mask = get_sample_pixel_and_mask(sam_masks, 5000, 50, cover_count=cover)
    loss_pos, loss_neg, pos_sim, neg_sim = contrastive_head(rendered_features, sam_masks, sampled_pixel, sampled_mask,
                                                            mode="soft", positive_th=0.75, negative_th=0.5,
                                                            mask_size=sizes)
    loss = loss_pos + loss_neg + opt.rfn * feature_norm_reg(full_res_rendered_features)

``rendered_features`` is the (32, H, W) feature image at mask resolution (after the reference's bilinear
``interpolate`` when the sizes differ, train.py:284).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .rasterizer import _bytes, _stream

_MODES = {"soft": 0, "all": 1, "hard": 2}


def _dev_index(dev):
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _masks_u8(sam_masks: torch.Tensor) -> torch.Tensor:
    if sam_masks.device.type != "cuda":
        raise RuntimeError("trase_amd.feature_head runs on the GPU only (there is no CPU path)")
    if sam_masks.dim() != 3:
        raise ValueError("sam_masks must be [N, H, W]")
    m = sam_masks if sam_masks.dtype == torch.bool else (sam_masks != 0)
    return m.contiguous().view(torch.uint8)


@torch.no_grad()
def mask_stats(sam_masks: torch.Tensor):
    """(cover_count [H, W] int32, mask_size [N] int32): ``sam_masks.sum(dim=0)`` and ``sam_masks.sum(-1).sum(-1)``
    (utils/feature_utils.py:23, :30) in one pass over the masks."""
    m = _masks_u8(sam_masks)
    N, H, W = m.shape
    dev = m.device
    cover = torch.empty((H, W), dtype=torch.int32, device=dev)
    size = torch.empty((N,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.trase_mask_stats(_lib.ptr(m), N, H * W, _lib.ptr(cover), _lib.ptr(size), _dev_index(dev), _stream(dev)),
               "trase_mask_stats")
    return cover, size


@torch.no_grad()
def get_sample_pixel_and_mask(sam_masks, num_sampled_pixels, num_sampled_masks, cover_count=None, rng="cpu"):
    """utils/feature_utils.py:17-26, the non-mask region taken from ``mask_stats`` instead of an int64 N x H x W
    reduction.  ``rng="cpu"`` (default) makes the same draws as the reference -- ``torch.rand`` on the CPU generator, in
    the same order, then copied to the GPU -- which costs ~3.4 ms at 1080p (2 M scalar Mersenne-Twister draws + an 8 MB
    copy); ``rng="cuda"`` draws the same Bernoulli fields from torch's CUDA generator (~0.03 ms)."""
    if rng not in ("cpu", "cuda"):
        raise ValueError("rng must be 'cpu' or 'cuda'")
    if cover_count is None:
        cover_count, _ = mask_stats(sam_masks)
    dev = sam_masks.device
    where = dev if rng == "cuda" else "cpu"
    mask_sample_rate = num_sampled_masks / (sam_masks.shape[0])
    sampled_mask = torch.rand(sam_masks.shape[0], device=where).to(dev) < mask_sample_rate
    pixel_sample_rate = num_sampled_pixels / (sam_masks.shape[-1] * sam_masks.shape[-2])
    sampled_pixel = torch.rand(sam_masks.shape[-2], sam_masks.shape[-1], device=where).to(dev) < pixel_sample_rate
    sampled_pixel = torch.logical_and(sampled_pixel, cover_count != 0)
    # (how many pixels the draw aims at: lets contrastive_head size its buffers without reading the count back from the device)
    # tied to THIS tensor's contents: (target count, number of pixels of the draw, version counter) -- an in-place edit afterwards
    # (`sampled_pixel |= ...`) bumps the version and contrastive_head falls back to counting
    sampled_pixel._trase_expected_count = (int(num_sampled_pixels), int(sampled_pixel.numel()), int(sampled_pixel._version))
    return sampled_pixel, sampled_mask


_COUNT_CHECKS: list = []      # (event, pinned int32[2] copy of the device count, capacity) of sync-free compactions not yet verified
_COUNT_FREE: list = []        # verified (pinned buffer, event) pairs, reused (no pinned allocation / event creation per iteration)


def _poll_count_checks(block: bool = False):
    """A compaction whose true count (count[1]) exceeded the buffer it was given dropped its highest pixels: reported here, by a
    LATER call of the head (or check_sampled_counts()), without making any call wait for the device."""
    keep = []
    for ev, pin, cap in _COUNT_CHECKS:
        if block:
            ev.synchronize()
        if block or ev.query():
            if len(_COUNT_FREE) < 8:
                _COUNT_FREE.append((pin, ev))
            if int(pin[1]) > cap:
                _COUNT_CHECKS[:] = []
                raise RuntimeError(f"trase_amd.feature_head: a sync-free contrastive_head call sampled {int(pin[1])} pixels but its index "
                                   f"buffer held {cap}: the highest pixels of that draw were dropped from its losses.  Pass a boolean "
                                   f"mask that does not come from get_sample_pixel_and_mask (counted on the host) for such draws.")
        else:
            keep.append((ev, pin, cap))
    _COUNT_CHECKS[:] = keep


def check_sampled_counts():
    """Blocking: verify every sync-free contrastive_head call issued so far (see _poll_count_checks)."""
    _poll_count_checks(block=True)


class _PairHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, masks_u8, sampled_mask_u8, n_sampled, mask_size, pix, mode, pth, nth, use_w, with_reg, s_dev=None):
        # s_dev: int32[2] on the device, s_dev[0] = number of valid entries of `pix` (trase_compact_pixels); None: all of them
        lib = _lib.load()
        dev = feats.device
        F, H, W = feats.shape
        S = pix.numel()
        f = feats.detach().float().contiguous()
        nbytes = C.c_size_t()
        _lib.check(lib.trase_pairhead_sizes(S, C.byref(nbytes)), "trase_pairhead_sizes")
        ws = _bytes(nbytes.value, dev)
        out8 = torch.empty(8, device=dev)
        d = _dev_index(dev)
        _lib.check(lib.trase_pairhead_forward_n(_lib.ptr(f), F, H * W, _lib.ptr(masks_u8), masks_u8.shape[0], _lib.ptr(sampled_mask_u8),
                                                int(n_sampled), _lib.ptr(mask_size), _lib.ptr(pix), S, _lib.ptr(s_dev), int(mode), float(pth),
                                                float(nth), int(use_w), _lib.ptr(out8), _lib.ptr(ws), ws.numel(), d, _stream(dev)),
                   "trase_pairhead_forward")
        ctx.s_dev = s_dev
        sims = out8[4:6].clone()
        ctx.mark_non_differentiable(sims)
        ctx.cfg = (F, H, W, S, int(mode), float(pth), float(nth), int(use_w), bool(with_reg))
        if not with_reg:
            ctx.save_for_backward(ws, out8, pix)
            return out8[0], out8[2], sims
        # the regulariser of train.py:281-282 on the same image: its dense gradient is written first in the backward and
        # the S sampled columns are added into it (no zero-fill, no separate accumulation pass)
        _lib.check(lib.trase_featnorm_sizes(H * W, C.byref(nbytes)), "trase_featnorm_sizes")
        ws_r = _bytes(nbytes.value, dev)
        out2 = torch.empty(2, device=dev)
        _lib.check(lib.trase_featnorm_forward(_lib.ptr(f), F, H * W, _lib.ptr(out2), _lib.ptr(ws_r), ws_r.numel(), d, _stream(dev)),
                   "trase_featnorm_forward")
        ctx.save_for_backward(ws, out8, pix, f, out2)
        return out8[0], out8[2], sims, out2[0]

    @staticmethod
    def backward(ctx, g_pos, g_neg, _g_sims, g_reg=None):
        lib = _lib.load()
        F, H, W, S, mode, pth, nth, use_w, with_reg = ctx.cfg
        ws, out8, pix = ctx.saved_tensors[:3]
        dev = ws.device
        d, st = _dev_index(dev), _stream(dev)
        g2 = torch.stack([g_pos.reshape(()), g_neg.reshape(())]).float().contiguous()
        d_feats = torch.empty((F, H, W), device=dev)
        if with_reg:
            f, out2 = ctx.saved_tensors[3:]
            gg = g_reg.reshape(1).float().contiguous()
            _lib.check(lib.trase_featnorm_backward(_lib.ptr(f), F, H * W, _lib.ptr(out2), _lib.ptr(gg), _lib.ptr(d_feats), d, st),
                       "trase_featnorm_backward")
        _lib.check(lib.trase_pairhead_backward_n(F, H * W, _lib.ptr(pix), S, _lib.ptr(ctx.s_dev), mode, pth, nth, use_w, _lib.ptr(out8),
                                                 _lib.ptr(g2), _lib.ptr(ws), ws.numel(), 1 if with_reg else 0, _lib.ptr(d_feats), d, st),
                   "trase_pairhead_backward")
        return (d_feats,) + (None,) * 11


def contrastive_head(rendered_features, sam_masks, sampled_pixel, sampled_mask, mode="soft", positive_th=0.75, negative_th=0.5,
                     use_weights=True, mask_size=None, with_norm_reg=False):
    """(loss_pos, loss_neg, pos_similarity, neg_similarity[, norm_reg]) of train.py:272-296:
    ``positive_pixel_pair_loss[mode](C, C_F, positive_th, weights)``, ``negative_pixel_pair_loss[mode](...)``,
    ``C_F[C == 1].mean()``, ``C_F[C == 0].mean()`` for the matrices the reference derives from ``sam_masks``,
    ``sampled_pixel``, ``sampled_mask`` and the (32, H, W) features.  No synchronisation when ``sampled_pixel`` comes from
    ``get_sample_pixel_and_mask`` (it carries the draw's target count: the indices are compacted and counted on the device); one --
    the number of sampled pixels -- for any other boolean mask (the reference synchronises at every boolean index).  ``with_norm_reg=True`` also returns the regulariser
    ``(1 - rendered_features.norm(dim=0).mean()) ** 2`` (train.py:281-282) of the same image -- valid when the rendered
    features already have the mask resolution, so that train.py:284's ``interpolate`` is the identity -- and shares one
    dense gradient pass with the pair losses."""
    if mode not in _MODES:
        raise ValueError(f"contrastive mode {mode!r} (expected one of {sorted(_MODES)})")
    m = _masks_u8(sam_masks)
    N, H, W = m.shape
    if rendered_features.dim() != 3 or tuple(rendered_features.shape[1:]) != (H, W) or rendered_features.shape[0] != 32:
        raise ValueError(f"rendered_features must be (32, {H}, {W}) -- the mask resolution (train.py:284 interpolates to it)")
    if rendered_features.device != m.device:
        raise ValueError("rendered_features and sam_masks live on different devices")
    if tuple(sampled_pixel.shape) != (H, W) or sampled_mask.numel() != N:
        raise ValueError("sampled_pixel must be [H, W] and sampled_mask [N]")
    dev = m.device
    if mask_size is None:
        _, mask_size = mask_stats(sam_masks)
    sm = (sampled_mask != 0).to(dev).contiguous().view(torch.uint8)
    n_sampled = N if N <= 256 else int(sm.sum())
    if n_sampled > 256:
        raise ValueError(f"{n_sampled} sampled masks (the membership bit sets hold 256)")
    expected = getattr(sampled_pixel, "_trase_expected_count", None)
    if expected is not None:
        # only while the tensor still IS the draw the count describes (same number of pixels, not edited in place since)
        expected = expected[0] if (isinstance(expected, tuple) and expected[1] == H * W
                                   and expected[2] == sampled_pixel._version) else None
    s_dev = None
    if expected is not None and sampled_pixel.dtype == torch.bool:
        if not torch.cuda.is_current_stream_capturing():
            _poll_count_checks()
        # no read-back (round 5): the indices are compacted on the device into a buffer sized for the draw's target plus eight
        # standard deviations of the binomial count; every kernel of the head takes the count from the device.  (A draw that
        # overflowed the buffer -- probability ~ 1e-15 -- would drop its highest pixels; count[1] keeps the true number.)
        cap = int(expected + 8.0 * expected ** 0.5 + 64)
        lib = _lib.load()
        flags = sampled_pixel.contiguous().view(torch.uint8)
        pix = torch.empty(cap, dtype=torch.int32, device=dev)
        s_dev = torch.empty(2, dtype=torch.int32, device=dev)
        nb = C.c_size_t()
        _lib.check(lib.trase_compact_pixels_sizes(H * W, C.byref(nb)), "trase_compact_pixels_sizes")
        cws = _bytes(nb.value, dev)
        _lib.check(lib.trase_compact_pixels(_lib.ptr(flags), H * W, _lib.ptr(pix), cap, _lib.ptr(s_dev), _lib.ptr(cws), cws.numel(),
                                            _dev_index(dev), _stream(dev)), "trase_compact_pixels")
        if len(_COUNT_CHECKS) < 64 and not torch.cuda.is_current_stream_capturing():   # the true count is looked at later, off the critical path
            pin, ev = _COUNT_FREE.pop() if _COUNT_FREE else (torch.empty(2, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
            pin.copy_(s_dev, non_blocking=True)
            ev.record(torch.cuda.current_stream(dev))
            _COUNT_CHECKS.append((ev, pin, cap))
    else:
        pix = torch.nonzero(sampled_pixel.reshape(-1)).reshape(-1).to(torch.int32)       # ascending = boolean-index order (synchronises)
        if pix.numel() == 0:
            z = rendered_features.sum() * 0.0
            nan = torch.full((), float("nan"), device=dev)
            return (z, z, nan, nan, feature_norm_reg(rendered_features)) if with_norm_reg else (z, z, nan, nan)
    res = _PairHead.apply(rendered_features, m, sm, n_sampled, mask_size.contiguous(), pix, _MODES[mode], positive_th, negative_th,
                          1 if use_weights else 0, bool(with_norm_reg), s_dev)
    lp, ln, sims = res[:3]
    return (lp, ln, sims[0], sims[1], res[3]) if with_norm_reg else (lp, ln, sims[0], sims[1])


class _FeatNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats):
        lib = _lib.load()
        dev = feats.device
        f = feats.detach().float().contiguous()
        F = f.shape[0]
        HW = f.numel() // F
        nbytes = C.c_size_t()
        _lib.check(lib.trase_featnorm_sizes(HW, C.byref(nbytes)), "trase_featnorm_sizes")
        ws = _bytes(nbytes.value, dev)
        out2 = torch.empty(2, device=dev)
        _lib.check(lib.trase_featnorm_forward(_lib.ptr(f), F, HW, _lib.ptr(out2), _lib.ptr(ws), ws.numel(), _dev_index(dev), _stream(dev)),
                   "trase_featnorm_forward")
        ctx.save_for_backward(f, out2)
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        f, out2 = ctx.saved_tensors
        dev = f.device
        F = f.shape[0]
        d = torch.empty_like(f)
        gg = g.reshape(1).float().contiguous()
        _lib.check(lib.trase_featnorm_backward(_lib.ptr(f), F, f.numel() // F, _lib.ptr(out2), _lib.ptr(gg), _lib.ptr(d), _dev_index(dev),
                                               _stream(dev)), "trase_featnorm_backward")
        return d


def feature_norm_reg(rendered_features: torch.Tensor) -> torch.Tensor:
    """``(1 - rendered_features.norm(dim=0, p=2).mean()) ** 2`` (train.py:281-282): one reduction pass forward, one
    dense pass backward."""
    if rendered_features.device.type != "cuda":
        raise RuntimeError("trase_amd.feature_head runs on the GPU only (there is no CPU path)")
    if rendered_features.dim() != 3:
        raise ValueError("rendered_features must be (C, H, W)")
    return _FeatNorm.apply(rendered_features)
