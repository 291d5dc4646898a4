"""Photometric loss heads of the training loop on the GPU as two HIP kernels:
``l1_loss`` (utils/loss_utils.py:30-31) and ``ssim`` (utils/loss_utils.py:56-86), combined at train.py:235-238 as
``(1 - lambda) * Ll1 + lambda * (1 - ssim(image, gt))``.

``l1_ssim(img, gt)`` returns both scalars from ONE forward kernel (separable 11-tap window from LDS tiles instead
of the reference's five depthwise 11x11 convolutions) and back-propagates both with ONE backward kernel.
``ssim(img1, img2)`` and ``l1_loss(x, gt)`` keep the reference signatures; when they are called on the same pair of
tensors (as train.py does) the second call reuses the first call's fused result."""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib
from .rasterizer import _bytes, _stream


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt):
        lib = _lib.load()
        dev = img.device
        x = img.detach().float().contiguous()
        y = gt.detach().float().contiguous()
        c, h, w = x.shape
        nbytes = C.c_size_t()
        _lib.check(lib.trase_loss_sizes(c, h, w, C.byref(nbytes)), "trase_loss_sizes")
        ws = _bytes(nbytes.value, dev)
        out2 = torch.empty(2, device=dev)
        d = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(lib.trase_loss_l1_ssim_forward(_lib.ptr(x), _lib.ptr(y), c, h, w, _lib.ptr(out2), _lib.ptr(ws), ws.numel(),
                                                  d, _stream(dev)), "trase_loss_l1_ssim_forward")
        ctx.save_for_backward(x, y, ws)
        return out2[0], out2[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        lib = _lib.load()
        x, y, ws = ctx.saved_tensors
        dev = x.device
        c, h, w = x.shape
        z = torch.zeros((), device=dev)
        g2 = torch.stack([g_l1 if g_l1 is not None else z, g_ssim if g_ssim is not None else z]).float().contiguous()
        d_img = torch.empty_like(x)
        d = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(lib.trase_loss_l1_ssim_backward(_lib.ptr(x), _lib.ptr(y), c, h, w, _lib.ptr(g2), _lib.ptr(ws), ws.numel(),
                                                   _lib.ptr(d_img), d, _stream(dev)), "trase_loss_l1_ssim_backward")
        return d_img, None


def _check(img, gt):
    if img.device.type != "cuda":
        raise RuntimeError("trase_amd.losses runs on the GPU only (there is no CPU path)")
    if img.dim() != 3 or img.shape != gt.shape:
        raise ValueError(f"expected two (C,H,W) images of equal shape, got {tuple(img.shape)} and {tuple(gt.shape)}")


def l1_ssim(img: torch.Tensor, gt: torch.Tensor):
    """(mean |img - gt|, mean SSIM(img, gt)) -- one fused forward, one fused backward (gradient w.r.t. img only,
    the reference's gt is a constant)."""
    _check(img, gt)
    return _L1SSIM.apply(img, gt)


_last = {"img": None, "gt": None, "ver": None, "val": None}


def _fused(img, gt):
    """One fused evaluation shared by l1_loss(x, gt) and ssim(x, gt) when train.py calls them back to back on the same
    tensors.  The cache holds WEAK references and compares object identity + version counters: a new tensor that
    happens to reuse a freed tensor's id() can never hit it."""
    ri, rg = _last["img"], _last["gt"]
    ver = (img._version, gt._version, torch.is_grad_enabled(), img.requires_grad)
    if ri is None or ri() is not img or rg() is not gt or _last["ver"] != ver:
        _last["val"] = l1_ssim(img, gt)
        _last["img"], _last["gt"], _last["ver"] = weakref.ref(img), weakref.ref(gt), ver
    return _last["val"]


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """utils/loss_utils.py:30-31."""
    return _fused(network_output, gt)[0]


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    """utils/loss_utils.py:56-86 for the configuration the training scripts use (window 11, size_average)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("trase_amd.losses.ssim: only window_size=11, size_average=True is compiled in")
    return _fused(img1, img2)[1]
