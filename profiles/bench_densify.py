"""Densification bookkeeping + densify / prune (second half) at the headline size (300k Gaussians):
(a) the per-iteration statistics update (train.py:362-365) fused vs the reference's boolean-mask statements;
(b) densify_and_prune (scene/gaussian_model.py:617-635): one plan + one gather launch vs the reference's three rounds
    of boolean-index / cat calls per tensor, restated here in PyTorch.  Prints one JSON line.

    python profiles/bench_densify.py [--points 300000] [--iters 10]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.densify import add_densification_stats, densify_and_prune  # noqa: E402
from trase_amd.optim import FusedAdam  # noqa: E402

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "gaussian_feats"]
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation", "gaussian_feats": "_gaussian_features"}
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,), "gaussian_feats": (1, 32)}


def make(P, extent, pd, opt_cls, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    m = SimpleNamespace(percent_dense=pd, feature_smooth_map=None, mode="from_scratch")
    vals = {n: rn(P, *SHAPES[n]) for n in NAMES}
    vals["opacity"] = vals["opacity"] * 3 - 2
    vals["scaling"] = torch.log(torch.tensor(pd * extent)) + vals["scaling"] * 1.2
    for n in NAMES:
        setattr(m, ATTR[n], nn.Parameter(vals[n].contiguous()))
    grp = lambda names: [{"params": [getattr(m, ATTR[n])], "lr": 1e-3, "name": n} for n in names]
    m.optimizer = {"GAUSSIAN": opt_cls(grp(NAMES[:6]), lr=0.0, eps=1e-15), "FEATURE": opt_cls(grp(NAMES[6:]), lr=0.0, eps=1e-15)}
    for mode in m.optimizer:
        for gg in m.optimizer[mode].param_groups:
            p = gg["params"][0]
            m.optimizer[mode].state[p] = {"step": torch.tensor(5.0), "exp_avg": rn(*p.shape), "exp_avg_sq": rn(*p.shape).abs()}
    m.denom = torch.randint(0, 4, (P, 1), device="cuda", generator=g).float()
    m.xyz_gradient_accum = torch.rand(P, 1, device="cuda", generator=g) * 0.0003 * m.denom      # ~10 % selected
    m.max_radii2D = torch.rand(P, device="cuda", generator=g) * 40
    return m


# ---- the reference's formulation, restated (scene/gaussian_model.py:472-635) -------------------------------------------
def ref_prune(m, mask):
    keep = ~mask
    for mode in m.optimizer:
        opt = m.optimizer[mode]
        for gg in opt.param_groups:
            p = gg["params"][0]
            st = opt.state.get(p, None)
            if st is not None:
                st["exp_avg"] = st["exp_avg"][keep]
                st["exp_avg_sq"] = st["exp_avg_sq"][keep]
                del opt.state[p]
                gg["params"][0] = nn.Parameter(p[keep].requires_grad_(True))
                opt.state[gg["params"][0]] = st
            else:
                gg["params"][0] = nn.Parameter(p[keep].requires_grad_(True))
            setattr(m, ATTR[gg["name"]], gg["params"][0])
    m.xyz_gradient_accum = m.xyz_gradient_accum[keep]
    m.denom = m.denom[keep]
    m.max_radii2D = m.max_radii2D[keep]


def ref_cat(m, new):
    for mode in m.optimizer:
        opt = m.optimizer[mode]
        for gg in opt.param_groups:
            p = gg["params"][0]
            ext = new[gg["name"]]
            st = opt.state.get(p, None)
            if st is not None:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del opt.state[p]
                gg["params"][0] = nn.Parameter(torch.cat((p, ext), dim=0).requires_grad_(True))
                opt.state[gg["params"][0]] = st
            else:
                gg["params"][0] = nn.Parameter(torch.cat((p, ext), dim=0).requires_grad_(True))
            setattr(m, ATTR[gg["name"]], gg["params"][0])
    n = m._xyz.shape[0]
    m.xyz_gradient_accum = torch.zeros((n, 1), device="cuda")
    m.denom = torch.zeros((n, 1), device="cuda")
    m.max_radii2D = torch.zeros((n,), device="cuda")


def build_rotation(r):
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device="cuda")
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


@torch.no_grad()
def ref_densify_and_prune(m, max_grad, min_opacity, extent, max_screen_size):
    grads = m.xyz_gradient_accum / m.denom
    grads[grads.isnan()] = 0.0
    sel = torch.where(torch.norm(grads, dim=-1) >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(torch.exp(m._scaling), dim=1).values <= m.percent_dense * extent)
    ref_cat(m, {n: getattr(m, ATTR[n])[sel] for n in NAMES})
    n0 = m._xyz.shape[0]
    padded = torch.zeros((n0), device="cuda")
    padded[:grads.shape[0]] = grads.squeeze()
    sel = torch.where(padded >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(torch.exp(m._scaling), dim=1).values > m.percent_dense * extent)
    stds = torch.exp(m._scaling[sel]).repeat(2, 1)
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device="cuda"), std=stds)
    rots = build_rotation(m._rotation[sel]).repeat(2, 1, 1)
    new = {n: getattr(m, ATTR[n])[sel].repeat(2, *([1] * (getattr(m, ATTR[n]).dim() - 1))) for n in NAMES}
    new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + m._xyz[sel].repeat(2, 1)
    new["scaling"] = torch.log(torch.exp(m._scaling[sel]).repeat(2, 1) / (0.8 * 2))
    ref_cat(m, new)
    ref_prune(m, torch.cat((sel, torch.zeros(2 * sel.sum(), device="cuda", dtype=bool))))
    mask = (torch.sigmoid(m._opacity) < min_opacity).squeeze()
    if max_screen_size:
        mask = torch.logical_or(torch.logical_or(mask, m.max_radii2D > max_screen_size), torch.exp(m._scaling).max(dim=1).values > 0.1 * extent)
    ref_prune(m, mask)
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=300_000)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    P, extent, pd = a.points, 5.2, 0.01
    out = {"points": P}
    # (a) statistics
    vp = torch.zeros(P, 3, device="cuda", requires_grad=True)
    vp.grad = torch.randn(P, 3, device="cuda") * 1e-3
    radii = (torch.randint(1, 60, (P,), device="cuda", dtype=torch.int32) * (torch.rand(P, device="cuda") < 0.6).int()).contiguous()
    m = make(P, extent, pd, FusedAdam)

    def stats_ref():
        vis = radii > 0
        m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis])
        m.xyz_gradient_accum[vis] += torch.norm(vp.grad[vis, :2], dim=-1, keepdim=True)
        m.denom[vis] += 1
    for name, fn in (("stats_torch_ms", stats_ref), ("stats_hip_ms", lambda: add_densification_stats(m, vp, radii))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize(); out[name] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
    # (b) densify_and_prune: fresh model per repetition (setup outside the timed region)
    for name, fn, cls in (("densify_torch_ms", ref_densify_and_prune, torch.optim.Adam), ("densify_hip_ms", densify_and_prune, FusedAdam)):
        ts = []
        for it in range(a.iters + 2):
            mm = make(P, extent, pd, cls, seed=it)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn(mm, 0.0002, 0.005, extent, 20)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            rows = mm._xyz.shape[0]
        out[name] = round(sum(ts[2:]) / len(ts[2:]), 3)
        out[name.replace("_ms", "_rows_after")] = rows
    out["stats_speedup"] = round(out["stats_torch_ms"] / out["stats_hip_ms"], 1)
    out["densify_speedup"] = round(out["densify_torch_ms"] / out["densify_hip_ms"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
