"""Multi-tensor Adam (first half): a drop-in for the ``torch.optim.Adam(l, lr=0.0, eps=1e-15)``
the reference builds over per-parameter groups (scene/gaussian_model.py:253-300, deform: scene/deform_model.py:38-47)
and steps at train.py:376-389.

``FusedAdam`` keeps torch.optim.Adam's param_groups and per-parameter state layout (``step``, ``exp_avg``,
``exp_avg_sq``), so the reference's densification code that edits the optimizer state in place
(scene/gaussian_model.py:472-534 ``replace_tensor_to_optimizer`` / ``cat_tensors_to_optimizer`` / ``_prune_optimizer``)
and ``state_dict()`` checkpoints keep working; ``step()`` is one HIP launch over every tensor that has a gradient
(``trase_adam_step``) instead of ~10 element-wise PyTorch kernels per tensor.  Under view-parallel DP every replica
executes the same arithmetic on the same all-reduced gradients, so replicas stay bit-identical."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .rasterizer import _stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, guard="auto", only=None):
        """only: restrict this call to the given parameters (by identity) -- the two halves of a phased gradient exchange
        (trase_amd.dp.FlatGradBucket.allreduce_phased) step ``only=first`` as soon as phase A has landed and the rest before the
        next render(); every parameter keeps its own moments and step counter, so two partial steps equal one whole step.
        guard="auto": under the rasterizer's sync-free policy the step is issued behind a device-side guard on the most
        recent forward's overflow flag (``trase_amd.rasterizer.current_guard``): if that forward overflowed its pair
        buffer, parameters and moments stay bit-identical and the step counters are rolled back when the overflow is
        reported.  guard=None: unconditional."""
        from . import rasterizer as _r
        g_handle = _r.current_guard() if guard == "auto" else guard
        guard_geom, on_overflow = g_handle if g_handle else (None, None)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # one launch per (betas, eps, device) combination: the reference uses a single one
        batches = {}
        stepped = []                             # (group index, parameter index) of what this step touches
        only_ids = None if only is None else {id(p) for p in only}
        for gi, group in enumerate(self.param_groups):
            for pi, p in enumerate(group["params"]):
                if p.grad is None or (only_ids is not None and id(p) not in only_ids):
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous float32 CUDA tensors (there is no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)          # host scalar, like torch.optim.Adam(capturable=False)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad
                if not g.is_contiguous() or g.dtype != torch.float32:
                    g = g.float().contiguous()
                key = (group["betas"], group["eps"], p.device)
                batches.setdefault(key, []).append((p, g, st, group["lr"]))
                stepped.append((gi, pi))
        for (betas, eps, dev), items in batches.items():
            d = dev.index if dev.index is not None else torch.cuda.current_device()
            for i in range(0, len(items), 16):
                chunk = items[i:i + 16]
                n = len(chunk)
                ptrs = lambda f: (C.c_void_p * n)(*[f(it) for it in chunk])
                P = ptrs(lambda it: it[0].data_ptr()); G = ptrs(lambda it: it[1].data_ptr())
                M = ptrs(lambda it: it[2]["exp_avg"].data_ptr()); V = ptrs(lambda it: it[2]["exp_avg_sq"].data_ptr())
                N = (C.c_int64 * n)(*[it[0].numel() for it in chunk])
                LR = (C.c_float * n)(*[float(it[3]) for it in chunk])
                ST = (C.c_int64 * n)(*[int(it[2]["step"]) for it in chunk])
                gp = _lib.ptr(guard_geom) if (guard_geom is not None and guard_geom.device == dev) else None
                _lib.check(lib.trase_adam_step_guarded(n, P, G, M, V, N, LR, ST, C.c_double(betas[0]), C.c_double(betas[1]),
                                                       float(eps), gp, d, _stream(dev)), "trase_adam_step_guarded")
        if on_overflow is not None:
            def undo():                  # the device skipped this step: the bias corrections must not count it
                # looked up NOW, by position: densify / prune may have replaced the parameters and their state dicts between
                # this step and the report of the overflow
                for gi_, pi_ in stepped:
                    if gi_ < len(self.param_groups) and pi_ < len(self.param_groups[gi_]["params"]):
                        st_ = self.state.get(self.param_groups[gi_]["params"][pi_])
                        if st_ and "step" in st_ and float(st_["step"]) > 0:
                            st_["step"] -= 1
            on_overflow(undo)
        return loss
