"""Deformation MLP of the reference (utils/time_utils.py:60-131 ``DeformNetwork``; wrapper
scene/deform_model.py:26-57 ``DeformModel.step``) on the bf16 matrix cores.

``deform_forward(state_dict, x, t)`` takes the reference network's parameters as they are
(``linear.{i}.weight/bias``, ``gaussian_warp/rotation/scaling.weight/bias``) and returns
``(d_xyz, d_rotation, d_scaling)`` like ``DeformNetwork.forward``.

* Under ``torch.no_grad()`` (FEATURE state train.py:200-202, style transfer
  train_style_transfer_nnfm.py:184-185, render.py:195, gui.py:965) it is one fused kernel
  (``trase_mlp_forward``).
* With gradients (GAUSSIAN state, train.py:202-204 + ``loss.backward()`` train.py:299) it is an autograd
  function over two C entry points: ``trase_mlp_forward_train`` (the fused forward, keeping bf16 activations
  and the ReLU gates) and ``trase_mlp_backward`` (fused data chain + split-N MFMA GEMMs for every parameter
  gradient).  The reference detaches ``x`` and ``t`` before the call, so no gradient is produced for them.

Numerics: bf16 operands, fp32 accumulation (the north star asks for a bf16 MFMA GEMM here).  The gradients are
the exact gradients of that bf16-evaluated network up to bf16 rounding of the back-propagated signal; against
fp32 autograd of the same parameters they differ by a few percent in relative L2, because ReLU gates whose
pre-activation is within bf16 rounding of zero open or close differently (same effect as torch.autocast)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Mapping, Tuple

import torch

from . import _lib
from .rasterizer import _bytes, _stream

PARAM_KEYS = tuple(k for i in range(8) for k in (f"linear.{i}.weight", f"linear.{i}.bias")) + (
    "gaussian_warp.weight", "gaussian_warp.bias", "gaussian_rotation.weight", "gaussian_rotation.bias",
    "gaussian_scaling.weight", "gaussian_scaling.bias")
_HIDDEN_KEYS = tuple(k for i in range(8) for k in (f"linear.{i}.weight", f"linear.{i}.bias"))
_TAIL_KEYS = ("gaussian_rotation.weight", "gaussian_rotation.bias", "gaussian_scaling.weight", "gaussian_scaling.bias")
# is_6dof (utils/time_utils.py:100-102, :111-118): the translation head is replaced by two 3-vector heads
KEYS_6DOF_W = _HIDDEN_KEYS + ("branch_w.weight", "branch_w.bias") + _TAIL_KEYS
KEYS_6DOF_V = _HIDDEN_KEYS + ("branch_v.weight", "branch_v.bias") + _TAIL_KEYS
TIMENET_KEYS = ("timenet.0.weight", "timenet.0.bias", "timenet.2.weight", "timenet.2.bias")   # is_blender only


def _fill_weights(tensors, dev, keep, is_blender=False, is_6dof=False) -> "_lib.MlpWeights":
    _EMB = 93 if is_blender else 84           # 63 + timenet output 30 | 63 + PE(t) 21  (utils/time_utils.py:70-97)
    def P(i):
        v = tensors[i].detach()
        if v.device != dev or v.dtype != torch.float32 or not v.is_contiguous():
            v = v.to(dev, torch.float32).contiguous()
        keep.append(v)
        return v

    w = _lib.MlpWeights()
    w.D, w.W, w.xyz_multires, w.t_multires = 8, 256, 10, (6 if is_blender else 10)
    w.is_blender, w.is_6dof = int(is_blender), int(is_6dof)
    w.variant = 0
    for i in range(8):
        wt, bs = P(2 * i), P(2 * i + 1)
        want = (256, _EMB) if i == 0 else ((256, _EMB + 256) if i == 5 else (256, 256))
        if tuple(wt.shape) != want:
            raise ValueError(f"linear.{i}.weight has shape {tuple(wt.shape)}, the compiled network expects {want}")
        w.weight[i], w.bias[i] = wt.data_ptr(), bs.data_ptr()
    w.w_warp, w.b_warp = P(16).data_ptr(), P(17).data_ptr()
    w.w_rotation, w.b_rotation = P(18).data_ptr(), P(19).data_ptr()
    w.w_scaling, w.b_scaling = P(20).data_ptr(), P(21).data_ptr()
    return w


def _prep_xt(x, t):
    n = x.shape[0]
    xs = x.detach().float().contiguous()
    # the reference passes fid.unsqueeze(0).expand(N, -1): a stride-0 view (train.py:196) -- keep it that way
    tt = t.detach().float()
    if tt.dim() == 2 and tt.shape[1] == 1 and tt.shape[0] == n and tt.stride(0) == 0:
        t_stride = 0
    else:
        tt = tt.reshape(n).contiguous()
        t_stride = 1
    return xs, tt, t_stride


def _dev_index(dev):
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _time_embedding(params, t, n):
    """is_blender: ``timenet(embed_time_fn(t))`` (utils/time_utils.py:74-80, :107-109) evaluated ONCE -- the reference
    feeds the same time to every row when is_blender (train.py:190-198: ``fid.unsqueeze(0).expand(N, -1)``, no
    ``ast_noise``) and runs the two small Linear layers on all N identical rows.  Returns the (30,) output with autograd
    history to the timenet parameters."""
    if t.dim() == 2 and t.shape[0] == n and t.stride(0) == 0:
        row = t[0:1]
    else:
        tt = t.reshape(n, -1)
        if n > 1 and not bool((tt == tt[0:1]).all()):
            raise NotImplementedError("trase_amd.deform: is_blender expects one time for all rows (train.py:190-198)")
        row = tt[0:1]
    row = row.reshape(1, 1).float()
    freqs = 2.0 ** torch.arange(6, device=row.device, dtype=torch.float32)               # get_embedder(6, 1)
    ang = row * freqs                                                                   # (1, 6)
    emb = torch.cat([row, torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).reshape(1, 12)], dim=-1)   # t, sin, cos, ...
    h = torch.relu(torch.nn.functional.linear(emb, params["timenet.0.weight"], params["timenet.0.bias"]))
    return torch.nn.functional.linear(h, params["timenet.2.weight"], params["timenet.2.bias"]).reshape(30)


# ---- row order of the training pair ("dead rows", trase_amd/csrc/mlp.hip) -------------------------------------------------
# The backward skips every 32-row tile whose cotangents are all zero: the Gaussians a view culls (radii == 0,
# gaussian_renderer/__init__.py:152) send back exactly zero.  Culling is spatially coherent, Gaussian indices are not; so the
# training pair evaluates the rows along a Morton curve of their positions.  The order only groups rows into tiles -- every row
# is computed exactly as before and lands at its own index -- so it may be stale: it is rebuilt when N changes and every
# ROW_ORDER_REFRESH calls (positions drift slowly; densification changes N).
ROW_ORDER_REFRESH = 64
_ROW_ORDER: dict = {"key": None, "age": 0, "perm": None}
_ROW_ORDER_MODE = os.environ.get("TRASE_MLP_ROW_ORDER", "morton")      # "morton" | "none"


def set_row_order(mode: str = "morton"):
    """"morton" (default): rows of the training pair are evaluated along a Morton curve; "none": in index order."""
    global _ROW_ORDER_MODE
    if mode not in ("morton", "none"):
        raise ValueError("row order must be 'morton' or 'none'")
    _ROW_ORDER_MODE = mode
    _ROW_ORDER.update(key=None, age=0, perm=None)


def morton_order(x: torch.Tensor) -> torch.Tensor:
    """int32 permutation that sorts the rows of x (N,3) along a 30-bit Morton curve of their bounding box."""
    with torch.no_grad():
        lo, hi = x.min(0).values, x.max(0).values
        q = ((x - lo) / (hi - lo).clamp_min(1e-20) * 1023.0).to(torch.int64).clamp_(0, 1023)

        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        return torch.argsort(code, stable=True).to(torch.int32)


def _row_order(x: torch.Tensor):
    n = x.shape[0]
    if _ROW_ORDER_MODE == "none" or n < 4096:          # a handful of tiles: nothing to skip
        return None
    key = (n, x.device)
    if _ROW_ORDER["key"] != key or _ROW_ORDER["age"] >= ROW_ORDER_REFRESH or _ROW_ORDER["perm"] is None:
        _ROW_ORDER.update(key=key, age=0, perm=morton_order(x))
    _ROW_ORDER["age"] += 1
    return _ROW_ORDER["perm"]


_LIVE: dict = {"track": False, "count": None}


def track_live_tiles(flag: bool = True):
    """Tests / benches: have every MLP backward leave its live-tile count (one stream-ordered 4-byte copy) for last_live_tiles()."""
    _LIVE["track"] = bool(flag)
    _LIVE["count"] = None


def last_live_tiles() -> int:
    """32-row tiles the most recent MLP backward worked on (the others held only zero cotangents).  Synchronises."""
    if _LIVE["count"] is None:
        raise RuntimeError("no MLP backward has run since track_live_tiles(True)")
    return int(_LIVE["count"].item())


class _DeformMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, is_blender, *params):
        lib = _lib.load()
        dev = x.device
        n = x.shape[0]
        keep = []
        w = _fill_weights(params, dev, keep, is_blender)
        if is_blender:                      # t = the (30,) timenet output, shared by all rows
            xs, tt, t_stride = x.detach().float().contiguous(), t.detach().float().contiguous(), 0
        else:
            xs, tt, t_stride = _prep_xt(x, t)
        d_xyz = torch.empty(n, 3, device=dev)
        d_rot = torch.empty(n, 4, device=dev)
        d_scale = torch.empty(n, 3, device=dev)
        ws_b, saved_b, bwd_b = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _lib.check(lib.trase_mlp_train_sizes(n, C.byref(ws_b), C.byref(saved_b), C.byref(bwd_b)), "trase_mlp_train_sizes")
        ws = _bytes(ws_b.value, dev)
        saved = _bytes(saved_b.value, dev)
        order = _row_order(xs)
        _lib.check(lib.trase_mlp_forward_train_rows(C.byref(w), _lib.ptr(xs), C.c_void_p(tt.data_ptr()), t_stride, n,
                                                    _lib.ptr(order), _lib.ptr(d_xyz), _lib.ptr(d_rot), _lib.ptr(d_scale),
                                                    _lib.ptr(saved), saved.numel(), _lib.ptr(ws), ws.numel(), _dev_index(dev),
                                                    _stream(dev)),
                   "trase_mlp_forward_train_rows")
        ctx.order = order                   # the SAME permutation tensor goes to the backward (the cache may move on)
        ctx.save_for_backward(saved, *params)
        ctx.n = n
        ctx.bwd_bytes = bwd_b.value
        ctx.is_blender = bool(is_blender)
        return d_xyz, d_rot, d_scale

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scale):
        lib = _lib.load()
        saved, *params = ctx.saved_tensors
        dev = saved.device
        n = ctx.n
        need = list(ctx.needs_input_grad[3:])
        need_t = ctx.is_blender and ctx.needs_input_grad[1]
        if n == 0:
            return (None, torch.zeros(30, device=dev) if need_t else None, None,
                    *[torch.zeros_like(p) if nd else None for p, nd in zip(params, need)])
        if need_t:                          # the time block's gradient is assembled from the bias gradients of its two layers
            need[1] = need[11] = True
        keep = []
        w = _fill_weights(params, dev, keep, ctx.is_blender)
        g_xyz, g_rot, g_scale = (None if g is None else g.float().contiguous() for g in (g_xyz, g_rot, g_scale))
        out = [torch.empty(p.shape, dtype=torch.float32, device=dev) if nd else None for p, nd in zip(params, need)]
        gr = _lib.MlpGrads()
        for i in range(8):
            gr.weight[i] = out[2 * i].data_ptr() if out[2 * i] is not None else None
            gr.bias[i] = out[2 * i + 1].data_ptr() if out[2 * i + 1] is not None else None
        gr.w_warp, gr.b_warp, gr.w_rotation, gr.b_rotation, gr.w_scaling, gr.b_scaling = (
            (o.data_ptr() if o is not None else None) for o in out[16:22])
        ws = _bytes(ctx.bwd_bytes, dev)
        _lib.check(lib.trase_mlp_backward_rows(C.byref(w), n, _lib.ptr(ctx.order), _lib.ptr(g_xyz), _lib.ptr(g_rot),
                                               _lib.ptr(g_scale), _lib.ptr(saved), saved.numel(), C.byref(gr), _lib.ptr(ws),
                                               ws.numel(), _dev_index(dev), _stream(dev)), "trase_mlp_backward_rows")
        if _LIVE["track"]:
            if _LIVE["count"] is None or _LIVE["count"].device != dev:
                _LIVE["count"] = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.trase_mlp_live_tiles(_lib.ptr(ws), ws.numel(), n, _lib.ptr(_LIVE["count"]), _stream(dev)),
                       "trase_mlp_live_tiles")
        g_t = None
        if need_t:
            # every row sees the same 30 inputs at columns 63..92 of layer 0 and of the skip layer, so
            # dL/dtemb = sum_rows dZ_0 W_0[:, 63:93] + sum_rows dZ_5 W_5[:, 63:93] = db_0 W_0[:, 63:93] + db_5 W_5[:, 63:93]
            g_t = out[1] @ params[0].detach()[:, 63:93] + out[11] @ params[10].detach()[:, 63:93]
        real = ctx.needs_input_grad[3:]
        return (None, g_t, None, *[o if nd else None for o, nd in zip(out, real)])


def _skew(w):
    """utils/rigid_utils.py:6-23."""
    z = torch.zeros_like(w[:, 0])
    return torch.stack([z, -w[:, 2], w[:, 1], w[:, 2], z, -w[:, 0], -w[:, 1], w[:, 0], z], dim=-1).reshape(-1, 3, 3)


def exp_se3(S: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """Screw axis (N, 6) and magnitude (N, 1) -> homogeneous transforms (N, 4, 4); utils/rigid_utils.py:43-86
    (Rodrigues' formula, Modern Robotics 3.51 / 3.88), same operation order."""
    w, v = S[:, :3], S[:, 3:]
    W = _skew(w)
    eye = torch.eye(3, device=S.device, dtype=S.dtype).unsqueeze(0).expand(W.shape[0], -1, -1)
    W2 = torch.bmm(W, W)
    th = theta.reshape(-1, 1, 1)
    R = eye + torch.sin(th) * W + (1.0 - torch.cos(th)) * W2
    p = torch.bmm(th * eye + (1.0 - torch.cos(th)) * W + (th - torch.sin(th)) * W2, v.unsqueeze(-1))
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=S.device, dtype=S.dtype).expand(R.shape[0], 1, 4)
    return torch.cat([torch.cat([R, p], dim=-1), bottom], dim=1)


def deform_forward(params: Mapping[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor,
                   is_blender: bool = False, is_6dof: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if x.device.type != "cuda":
        raise RuntimeError("deform_forward runs on the GPU only (there is no CPU path)")
    if is_6dof:
        # utils/time_utils.py:111-118.  The fused kernels have one 3-vector head next to rotation and scaling, so the two
        # screw-axis heads take two evaluations of the network (the second one's rotation / scaling outputs are unused and
        # receive no cotangent); autograd adds the two passes' gradients of the shared layers.  Twice the cost of the
        # default network -- is_6dof is an option of the reference (arguments: --is_6dof), off by default.
        pw = {k2: params[k] for k, k2 in zip(KEYS_6DOF_W, PARAM_KEYS)}
        pv = {k2: params[k] for k, k2 in zip(KEYS_6DOF_V, PARAM_KEYS)}
        for k in TIMENET_KEYS:
            if k in params:
                pw[k] = pv[k] = params[k]
        w, rotation, scaling = deform_forward(pw, x, t, is_blender, False)
        v, _, _ = deform_forward(pv, x, t, is_blender, False)
        theta = torch.norm(w, dim=-1, keepdim=True)
        w = w / theta + 1e-5
        v = v / theta + 1e-5
        return exp_se3(torch.cat([w, v], dim=-1), theta), rotation, scaling
    tensors = [params[k] for k in PARAM_KEYS]
    n = x.shape[0]
    if torch.is_grad_enabled() and any(p.requires_grad for p in tensors + ([params[k] for k in TIMENET_KEYS] if is_blender else [])):
        if x.requires_grad or t.requires_grad:
            raise NotImplementedError("trase_amd.deform: x and t are detached inputs in the reference "
                                      "(scene/deform_model.py:34-35 called at train.py:202-204); no gradient is produced for them")
        tin = _time_embedding(params, t.detach(), n) if is_blender else t
        return _DeformMLP.apply(x, tin, bool(is_blender), *tensors)
    lib = _lib.load()
    dev = x.device
    keep = []
    w = _fill_weights(tensors, dev, keep, is_blender, is_6dof)
    if is_blender:
        with torch.no_grad():
            tt = _time_embedding(params, t, n).float().contiguous()
        xs, t_stride = x.detach().float().contiguous(), 0
    else:
        xs, tt, t_stride = _prep_xt(x, t)
    d_xyz = torch.empty(n, 3, device=dev)
    d_rot = torch.empty(n, 4, device=dev)
    d_scale = torch.empty(n, 3, device=dev)
    nbytes = C.c_size_t()
    _lib.check(lib.trase_mlp_sizes(C.byref(nbytes)), "trase_mlp_sizes")
    ws = _bytes(nbytes.value, dev)
    _lib.check(lib.trase_mlp_forward(C.byref(w), _lib.ptr(xs), C.c_void_p(tt.data_ptr()), t_stride, n, _lib.ptr(d_xyz),
                                     _lib.ptr(d_rot), _lib.ptr(d_scale), _lib.ptr(ws), ws.numel(), _dev_index(dev),
                                     _stream(dev)), "trase_mlp_forward")
    return d_xyz, d_rot, d_scale


class DeformNetworkHIP(torch.nn.Module):
    """Wraps a reference-shaped ``DeformNetwork`` (anything whose parameters carry the reference's names) and
    evaluates ``forward(x, t)`` with the fused kernels, with or without gradients."""

    def __init__(self, net: torch.nn.Module):
        super().__init__()
        self.net = net

    def forward(self, x, t):
        params = dict(self.net.named_parameters())
        return deform_forward(params, x, t, is_blender=bool(getattr(self.net, "is_blender", False)),
                              is_6dof=bool(getattr(self.net, "is_6dof", False)))
