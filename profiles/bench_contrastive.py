#!/usr/bin/env python
"""Secondary measurement: the FEATURE-state pixel-pair losses in 'soft' mode (utils/loss_utils.py:304-349, train.py:290-291)
on S x S matrices, S = 5000 (num_sampled_pixels, arguments/__init__.py:126), forward + backward w.r.t. C_F --
fused HIP kernels vs the reference's composition restated in PyTorch (which synchronises on torch.nonzero twice)."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from trase_amd.losses import (pixel_mask_correspondence_loss_soft_hard_positive as soft_pos,
                              pixel_mask_correspondence_loss_soft_negative as soft_neg)


def ref(C, C_F, pth, nth, w):
    n = C_F.shape[0]
    diag = torch.eye(n, dtype=torch.bool, device=C_F.device)
    tot = 0
    for neg in (False, True):
        cond = torch.logical_and(C_F > nth, C == 0) if neg else torch.logical_and(C_F < pth, C == 1)
        m = torch.triu(torch.logical_and(torch.any(cond, dim=0), ~diag), diagonal=0)
        npair = torch.nonzero(m).shape[0]
        m = torch.logical_and(m, C == (0 if neg else 1))
        tot = tot + ((w[m] * torch.relu(C_F[m])).sum() if neg else (-w[m] * C_F[m]).sum()) / npair
    return tot


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    memb = (torch.rand(50, S, device=dev) < 0.1).float()
    C = (memb.t() @ memb != 0).float()
    f = torch.nn.functional.normalize(torch.randn(S, 32, device=dev) + memb.t() @ torch.randn(50, 32, device=dev), dim=-1)
    CF = (f @ f.t()).requires_grad_(True)
    w = 1.0 + 9.0 * torch.rand(S, S, device=dev)

    def run_ref():
        CF.grad = None
        ref(C, CF, 0.75, 0.5, w).backward()

    def run_hip():
        CF.grad = None
        (soft_pos(C, CF, 0.75, w) + soft_neg(C, CF, 0.5, w)).backward()

    run_ref(); g0 = CF.grad.clone()
    run_hip(); g1 = CF.grad.clone()
    print(json.dumps({"S": S, "hip_fwd_bwd_ms": round(timed(run_hip), 4), "torch_fwd_bwd_ms": round(timed(run_ref), 4),
                      "max_rel_grad_diff": float((g1 - g0).abs().max() / g0.abs().max())}))


if __name__ == "__main__":
    main()
