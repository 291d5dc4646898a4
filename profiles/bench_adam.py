#!/usr/bin/env python
"""Secondary measurement: one optimizer step over the seven Gaussian parameter tensors (300k Gaussians, 91 floats each;
scene/gaussian_model.py:253-300) -- FusedAdam (one HIP launch) vs torch.optim.Adam (default foreach implementation)."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.optim import FusedAdam


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    shapes = [(n, 3), (n, 1, 3), (n, 15, 3), (n, 1), (n, 3), (n, 4), (n, 1, 32)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3, 2.5e-3]
    a = [torch.randn(*sh, device="cuda").requires_grad_(True) for sh in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    for pa, pb in zip(a, b):
        pa.grad = torch.randn_like(pa); pb.grad = pa.grad.clone()
    mk = lambda ps: [{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)]
    ref = torch.optim.Adam(mk(a), lr=0.0, eps=1e-15)
    opt = FusedAdam(mk(b), lr=0.0, eps=1e-15)
    print(json.dumps({"n": n, "floats": sum(p.numel() for p in a), "fused_adam_ms": round(timed(opt.step), 4),
                      "torch_adam_ms": round(timed(ref.step), 4)}))


if __name__ == "__main__":
    main()
