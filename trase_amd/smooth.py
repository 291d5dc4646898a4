"""KNN feature smoothing of the FEATURE state as one HIP gather kernel with a
gather-style (atomic-free) backward.

``smoothed_gaussian_features(pc, K, dropout)`` mirrors ``GaussianModel.get_smoothed_gaussian_features``
(scene/gaussian_model.py:79-104): same cache (``pc.feature_smooth_map = {"K", "m"}``), same KNN call
(``pytorch3d.ops.knn_points`` -- the top-level shim of this repository), same ``torch.randperm(K)[:int(K*dropout)]``
neighbour-slot selection (so the host RNG stream advances identically), same (N,1,32) result.  The reference
materialises ``normed[idx[:, sel], 0, :]`` (N x S x 32) and back-propagates through an index_put with atomics; here
``trase_smooth_forward`` / ``trase_smooth_backward`` do the gather-mean and its transpose directly, the transpose
over a reverse adjacency that is built once per KNN map and cached next to it."""
from __future__ import annotations

import torch

from . import _lib
from .rasterizer import _stream


def reverse_adjacency(idx: torch.Tensor):
    """CSR of 'who points at me': rev_src = flat positions i*K+k of ``idx`` sorted stably by target, rev_ptr (P+1)."""
    P, K = idx.shape
    flat = idx.reshape(-1)
    order = torch.sort(flat, stable=True).indices
    counts = torch.bincount(flat, minlength=P)
    rev_ptr = torch.zeros(P + 1, dtype=torch.int32, device=idx.device)
    rev_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return rev_ptr, order.to(torch.int32).contiguous()


class _SmoothFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx, sel, rev_ptr, rev_src):
        lib = _lib.load()
        dev = feats.device
        P, K = idx.shape
        F = feats.shape[-1]
        f2 = feats.detach().reshape(P, F).float().contiguous()
        if sel.device.type == "cpu":
            # (a copy from pageable host memory blocks the host until the stream has drained -- once per iteration, in front of
            # the whole render: through the pinned-memory cache the upload is asynchronous)
            if torch.cuda.is_current_stream_capturing():
                # whole-iteration capture (torch.cuda.graph): no host-to-device copy inside a capture (a pinned allocation, and the
                # event query of the pinned-memory cache behind it, invalidate the capture on ROCm 7) -- the few slot numbers are
                # written by fill kernels with the values as launch arguments.  The draw is FROZEN into the graph.
                sel_dev = torch.empty(int(sel.numel()), dtype=torch.int32, device=dev)
                for j, k in enumerate(sel.tolist()):
                    sel_dev[j:j + 1].fill_(int(k))
            else:
                sel_dev = sel.to(torch.int32).contiguous().pin_memory().to(dev, non_blocking=True)
        else:
            sel_dev = sel.to(device=dev, dtype=torch.int32).contiguous()
        S = int(sel_dev.numel())
        inv_norm = torch.empty(P, device=dev)
        out = torch.empty(P, F, device=dev)
        d = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(lib.trase_smooth_forward(_lib.ptr(f2), _lib.ptr(idx), P, F, K, _lib.ptr(sel_dev), S, _lib.ptr(inv_norm),
                                            _lib.ptr(out), d, _stream(dev)), "trase_smooth_forward")
        ctx.save_for_backward(f2, inv_norm, rev_ptr, rev_src)
        mask = 0
        for k in sel.tolist():
            mask |= 1 << int(k)
        ctx.meta = (P, F, K, S, mask, feats.shape)
        return out.reshape(P, 1, F)

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        f2, inv_norm, rev_ptr, rev_src = ctx.saved_tensors
        P, F, K, S, mask, shape = ctx.meta
        dev = f2.device
        g = g_out.reshape(P, F).float().contiguous()
        g_feat = torch.empty(P, F, device=dev)
        d = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(lib.trase_smooth_backward(_lib.ptr(f2), _lib.ptr(inv_norm), P, F, K, mask, S, _lib.ptr(rev_ptr),
                                             _lib.ptr(rev_src), _lib.ptr(g), _lib.ptr(g_feat), d, _stream(dev)),
                   "trase_smooth_backward")
        return g_feat.reshape(shape), None, None, None, None


def smooth_features(feats: torch.Tensor, idx: torch.Tensor, sel: torch.Tensor, rev=None) -> torch.Tensor:
    """mean over the neighbour slots ``sel`` of the L2-normalised rows ``feats[idx[:, sel]]`` -> (P,1,F)."""
    if feats.device.type != "cuda":
        raise RuntimeError("smooth_features runs on the GPU only (there is no CPU path)")
    if idx.dtype != torch.int64 or idx.dim() != 2 or not idx.is_contiguous():
        idx = idx.to(torch.int64).contiguous()
    if rev is None:
        rev = reverse_adjacency(idx)
    return _SmoothFeatures.apply(feats, idx, sel, rev[0], rev[1])


def smoothed_gaussian_features(pc, K: int = 16, dropout: float = 0.5) -> torch.Tensor:
    """Drop-in for ``pc.get_smoothed_gaussian_features(K, dropout)`` (scene/gaussian_model.py:79-104)."""
    if K <= 1:
        return pc._gaussian_features
    assert dropout < 0 or int(K * dropout) >= 1
    fmap = getattr(pc, "feature_smooth_map", None)
    with torch.no_grad():
        if fmap is None or fmap["K"] != K:
            import pytorch3d.ops
            xyz = pc.get_xyz
            nearest_k_idx = pytorch3d.ops.knn_points(xyz.unsqueeze(0), xyz.unsqueeze(0), K=K).idx.squeeze()
            fmap = {"K": K, "m": nearest_k_idx}
            pc.feature_smooth_map = fmap
        if "rev" not in fmap:
            fmap["rev"] = reverse_adjacency(fmap["m"].to(torch.int64).contiguous())
    if 0 < dropout < 1:
        sel = torch.randperm(K)[: int(K * dropout)]
    else:
        sel = torch.arange(K)
    return smooth_features(pc._gaussian_features, fmap["m"], sel, fmap["rev"])
