"""Densification bookkeeping and densify / prune (second half) for a reference-style
``GaussianModel`` (scene/gaussian_model.py): same attributes, same two Adam optimizers with one parameter per named
group, same results -- one planning pass, one host read of five counters and one gather launch over every parameter
and Adam moment instead of three rounds of boolean-index / cat calls per tensor.

``add_densification_stats(model, viewspace_point_tensor, radii)`` replaces train.py:362-365 (the ``max_radii2D``
update and ``GaussianModel.add_densification_stats``, scene/gaussian_model.py:637-639); it does not synchronise (the
reference's boolean-mask indexing does, three times).

``densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size)`` replaces
``GaussianModel.densify_and_prune`` (scene/gaussian_model.py:617-635).  The model is duck-typed: ``_xyz``,
``_features_dc``, ``_features_rest``, ``_opacity``, ``_scaling``, ``_rotation``, ``_gaussian_features`` (and
``_clusters['id']`` in 'finetuning' mode), ``optimizer`` (dict of torch optimizers whose groups are named as in
training_setup, :253-289), ``xyz_gradient_accum``, ``denom``, ``max_radii2D``, ``percent_dense``,
``feature_smooth_map``.  Under view-parallel DP call ``trase_amd.dp.allreduce_densify_stats`` first and seed every
rank's generator identically: all replicas then take identical decisions and draw identical split samples.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from .rasterizer import _bytes, _stream

_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
         "rotation": "_rotation", "gaussian_feats": "_gaussian_features"}


def _dev_index(dev):
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _f32(t, what):
    if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"trase_amd.densify: {what} must be a contiguous float32 CUDA tensor (there is no CPU path)")
    return t


@torch.no_grad()
def add_densification_stats(model, viewspace_point_tensor: torch.Tensor, radii: torch.Tensor) -> None:
    """train.py:362-365: ``max_radii2D[vis] = max(max_radii2D[vis], radii[vis])`` and
    ``add_densification_stats(viewspace_point_tensor, vis)`` with ``vis = radii > 0``, in place, one launch."""
    g = viewspace_point_tensor.grad
    if g is None:
        raise RuntimeError("add_densification_stats: viewspace_point_tensor has no gradient (call after backward())")
    g = _f32(g, "viewspace_point_tensor.grad")
    P = g.shape[0]
    if g.dim() != 2 or g.shape[1] != 3 or radii.shape[0] != P:
        raise ValueError("expected a [P, 3] screen-space gradient and [P] radii")
    rad = radii if radii.dtype == torch.int32 and radii.is_contiguous() else radii.to(torch.int32).contiguous()
    acc, den, mr = _f32(model.xyz_gradient_accum, "xyz_gradient_accum"), _f32(model.denom, "denom"), _f32(model.max_radii2D, "max_radii2D")
    if acc.numel() != P or den.numel() != P or mr.numel() != P:
        raise ValueError("densification statistics do not match the number of Gaussians")
    lib = _lib.load()
    dev = g.device
    # under the sync-free policy the statistics of a view whose pair buffer overflowed are not taken (device-side guard)
    from . import rasterizer as _r
    gh = _r.current_guard()
    gp = _lib.ptr(gh[0]) if (gh and gh[0].device == dev) else None
    _lib.check(lib.trase_densify_stats_guarded(_lib.ptr(g), _lib.ptr(rad), _lib.ptr(acc), _lib.ptr(den), _lib.ptr(mr), P, gp,
                                               _dev_index(dev), _stream(dev)), "trase_densify_stats_guarded")


def _groups(model):
    """(optimizer, group, name) for every parameter group of both optimizers, in the reference's iteration order
    (scene/gaussian_model.py:473-476)."""
    out = []
    for mode in ("GAUSSIAN", "FEATURE"):
        opt = model.optimizer[mode]
        for group in opt.param_groups:
            if len(group["params"]) != 1:
                raise ValueError("one parameter per group expected (scene/gaussian_model.py:516)")
            out.append((opt, group))
    return out


@torch.no_grad()
def densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size, normal_samples: torch.Tensor | None = None):
    """scene/gaussian_model.py:617-635.  Returns (num_clone, num_split) as python ints (the reference returns 0-d
    tensors; nobody reads them, train.py:370).  ``normal_samples`` ([2M, 3] standard normals, M = number of
    split-selected rows) overrides the draw from torch's CUDA generator -- the tests use it to replay the reference's
    samples.  ``torch.cuda.empty_cache()`` (:630) is not called."""
    lib = _lib.load()
    xyz = _f32(model._xyz.data, "_xyz")
    dev = xyz.device
    d, st = _dev_index(dev), _stream(dev)
    P = xyz.shape[0]
    scaling, rotation, opacity = _f32(model._scaling.data, "_scaling"), _f32(model._rotation.data, "_rotation"), _f32(model._opacity.data, "_opacity")
    acc, den = _f32(model.xyz_gradient_accum, "xyz_gradient_accum"), _f32(model.denom, "denom")
    if acc.numel() != P or den.numel() != P or scaling.shape != (P, 3) or rotation.shape != (P, 4) or opacity.numel() != P:
        raise ValueError("model tensors do not agree on the number of Gaussians")
    nbytes = C.c_size_t()
    _lib.check(lib.trase_densify_sizes(P, C.byref(nbytes)), "trase_densify_sizes")
    ws = _bytes(nbytes.value, dev)
    counts = torch.empty(5, dtype=torch.int32, device=dev)
    # thresholds as the reference's comparisons see them: python doubles rounded to float32 by type promotion
    _lib.check(lib.trase_densify_plan(_lib.ptr(acc), _lib.ptr(den), _lib.ptr(scaling), _lib.ptr(opacity), P, float(max_grad),
                                      float(model.percent_dense * extent), float(min_opacity), 1 if max_screen_size else 0,
                                      float(0.1 * extent), _lib.ptr(counts), _lib.ptr(ws), ws.numel(), d, st),
               "trase_densify_plan")
    n_orig, n_clone_kept, n_child, n_split, n_clone = counts.tolist()          # the one synchronisation
    new_P = n_orig + n_clone_kept + 2 * n_child
    z = None
    if n_split > 0:
        if normal_samples is None:
            z = torch.randn((2 * n_split, 3), device=dev)                       # torch.normal(mean=0, std) = randn * std
        else:
            z = _f32(normal_samples, "normal_samples")
            if z.shape != (2 * n_split, 3):
                raise ValueError(f"normal_samples must be [{2 * n_split}, 3]")
    # gather table: every parameter, and both Adam moments where a state exists
    src, dst, rows, zero_new, replaced = [], [], [], [], []
    for opt, group in _groups(model):
        p = group["params"][0]
        pd = _f32(p.data, f"parameter '{group['name']}'")
        if pd.shape[0] != P:
            raise ValueError(f"parameter '{group['name']}' has {pd.shape[0]} rows, expected {P}")
        new_p = torch.empty((new_P,) + tuple(pd.shape[1:]), device=dev)
        row = pd.numel() // P
        src.append(pd); dst.append(new_p); rows.append(row); zero_new.append(0)
        state = opt.state.get(p, None)
        new_m = new_v = None
        if state is not None and "exp_avg" in state:
            new_m, new_v = torch.empty_like(new_p), torch.empty_like(new_p)
            src += [_f32(state["exp_avg"], "exp_avg"), _f32(state["exp_avg_sq"], "exp_avg_sq")]
            dst += [new_m, new_v]; rows += [row, row]; zero_new += [1, 1]
        replaced.append((opt, group, p, state, new_p, new_m, new_v))
    by_name = {g["name"]: np_ for (_, g, _, _, np_, _, _) in replaced}
    if "xyz" not in by_name or "scaling" not in by_name:
        raise ValueError("the optimizers hold no 'xyz' / 'scaling' group (scene/gaussian_model.py:253-266)")
    if new_P > 0:
        n = len(src)
        for i in range(0, n, 32):
            j = min(i + 32, n)
            k = j - i
            S = (C.c_void_p * k)(*[t.data_ptr() for t in src[i:j]])
            D = (C.c_void_p * k)(*[t.data_ptr() for t in dst[i:j]])
            R = (C.c_int32 * k)(*rows[i:j])
            Z = (C.c_int32 * k)(*zero_new[i:j])
            last = j == n
            _lib.check(lib.trase_densify_apply(k, S, D, R, Z, P, new_P, _lib.ptr(xyz), _lib.ptr(scaling), _lib.ptr(rotation),
                                               _lib.ptr(z) if (last and z is not None) else None, _lib.ptr(by_name["xyz"]),
                                               _lib.ptr(by_name["scaling"]), _lib.ptr(ws), ws.numel(), d, st),
                       "trase_densify_apply")
    # optimizer surgery as scene/gaussian_model.py:478-487: new nn.Parameter, state carried over under the new key
    for opt, group, p, state, new_p, new_m, new_v in replaced:
        param = nn.Parameter(new_p.requires_grad_(True))
        if state is not None:
            if new_m is not None:
                state["exp_avg"], state["exp_avg_sq"] = new_m, new_v
            del opt.state[p]
            opt.state[param] = state
        group["params"][0] = param
        name = group["name"]
        if name in _ATTR:
            setattr(model, _ATTR[name], param)
        elif name == "cls":
            model._clusters["id"] = param
    model.xyz_gradient_accum = torch.zeros((new_P, 1), device=dev)
    model.denom = torch.zeros((new_P, 1), device=dev)
    model.max_radii2D = torch.zeros((new_P,), device=dev)
    model.feature_smooth_map = None
    return n_clone, n_split
