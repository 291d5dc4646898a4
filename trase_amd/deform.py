"""Deformation MLP of the reference (utils/time_utils.py:60-131 ``DeformNetwork``; wrapper
scene/deform_model.py:26-57 ``DeformModel.step``) on the bf16 matrix cores.

``deform_forward(state_dict, x, t)`` takes the reference network's parameters as they are
(``linear.{i}.weight/bias``, ``gaussian_warp/rotation/scaling.weight/bias``) and returns
``(d_xyz, d_rotation, d_scaling)`` like ``DeformNetwork.forward``.  Forward only: the call sites that
run under ``torch.no_grad()`` (FEATURE state train.py:200-202, style transfer
train_style_transfer_nnfm.py:184-185, render.py:195, gui.py:965)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Mapping, Tuple

import torch

from . import _lib
from .rasterizer import _bytes, _stream


def deform_forward(params: Mapping[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor,
                   is_blender: bool = False, is_6dof: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if x.device.type != "cuda":
        raise RuntimeError("deform_forward runs on the GPU only (there is no CPU path)")
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params.values())):
        raise NotImplementedError("trase_amd.deform: forward-only kernel; call it under torch.no_grad() "
                                  "(the training backward of the MLP is not built yet)")
    lib = _lib.load()
    dev = x.device
    n = x.shape[0]
    keep = []

    def P(name):
        v = params[name].detach()
        if v.device != dev or v.dtype != torch.float32 or not v.is_contiguous():
            v = v.to(dev, torch.float32).contiguous()
        keep.append(v)
        return v

    w = _lib.MlpWeights()
    w.D, w.W, w.xyz_multires, w.t_multires = 8, 256, 10, 10
    w.is_blender, w.is_6dof = int(is_blender), int(is_6dof)
    w.variant = int(os.environ.get("TRASE_MLP_VARIANT", "0"), 0)
    for i in range(8):
        wt, bs = P(f"linear.{i}.weight"), P(f"linear.{i}.bias")
        want = (256, 84) if i == 0 else ((256, 340) if i == 5 else (256, 256))
        if tuple(wt.shape) != want:
            raise ValueError(f"linear.{i}.weight has shape {tuple(wt.shape)}, the compiled network expects {want}")
        w.weight[i], w.bias[i] = wt.data_ptr(), bs.data_ptr()
    w.w_warp, w.b_warp = P("gaussian_warp.weight").data_ptr(), P("gaussian_warp.bias").data_ptr()
    w.w_rotation, w.b_rotation = P("gaussian_rotation.weight").data_ptr(), P("gaussian_rotation.bias").data_ptr()
    w.w_scaling, w.b_scaling = P("gaussian_scaling.weight").data_ptr(), P("gaussian_scaling.bias").data_ptr()

    xs = x.detach().float().contiguous()
    # the reference passes fid.unsqueeze(0).expand(N, -1): a stride-0 view (train.py:196) -- keep it that way
    tt = t.detach().float()
    if tt.dim() == 2 and tt.shape[1] == 1 and tt.shape[0] == n and tt.stride(0) == 0:
        t_stride = 0
    else:
        tt = tt.reshape(n).contiguous()
        t_stride = 1
    d_xyz = torch.empty(n, 3, device=dev)
    d_rot = torch.empty(n, 4, device=dev)
    d_scale = torch.empty(n, 3, device=dev)
    nbytes = C.c_size_t()
    _lib.check(lib.trase_mlp_sizes(C.byref(nbytes)), "trase_mlp_sizes")
    ws = _bytes(nbytes.value, dev)
    d = dev.index if dev.index is not None else torch.cuda.current_device()
    _lib.check(lib.trase_mlp_forward(C.byref(w), _lib.ptr(xs), C.c_void_p(tt.data_ptr()), t_stride, n, _lib.ptr(d_xyz),
                                     _lib.ptr(d_rot), _lib.ptr(d_scale), _lib.ptr(ws), ws.numel(), d, _stream(dev)),
               "trase_mlp_forward")
    return d_xyz, d_rot, d_scale


class DeformNetworkHIP(torch.nn.Module):
    """Wraps a reference-shaped ``DeformNetwork`` (anything whose state_dict has the reference's keys) and
    evaluates ``forward(x, t)`` with the fused kernel when gradients are off."""

    def __init__(self, net: torch.nn.Module):
        super().__init__()
        self.net = net

    def forward(self, x, t):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            return self.net(x, t)          # training: the reference's own PyTorch module
        return deform_forward(dict(self.net.state_dict()), x, t)
