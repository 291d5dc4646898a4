#!/usr/bin/env python
"""Secondary measurement: the fused NNFM style loss (trase_amd.losses.loss_nnfm_style, nnfm.hip) forward + backward
vs the reference formulation (utils/loss_utils.py:223-228: normalise, N1 x N2 matmul, amin, mean) in PyTorch fp32,
at the conv4_1 size of a 1280x960 frame (config 5: 120 x 160 = 19 200 positions, 512 channels) and of a 1080p frame
(135 x 240 = 32 400 positions; the reference's cosine matrix is 4.2 GB there)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.losses import loss_nnfm_style  # noqa: E402
from trase_amd import rasterizer as R  # noqa: E402


def ref(f1, f2):
    a = f1 / torch.linalg.norm(f1, dim=0)
    b = f2 / torch.linalg.norm(f2, dim=0)
    return torch.mean(torch.amin(1.0 - torch.matmul(a.T, b), dim=1))


def timed(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for name, n in (("config5_1280x960", 120 * 160), ("1080p", 135 * 240)):
        g = torch.Generator(device="cpu").manual_seed(0)
        f1 = torch.relu(torch.randn(512, n, generator=g) + 0.3).to(dev).requires_grad_(True)
        f2 = torch.relu(torch.randn(512, n, generator=g) + 0.3).to(dev)

        def hip():
            f1.grad = None
            loss_nnfm_style(f1, f2).backward()

        def torch_ref():
            f1.grad = None
            ref(f1, f2).backward()

        R.profile_enable(1)
        hip(); torch.cuda.synchronize()
        kern = {k: round(v["ms"], 4) for k, v in R.profile_report().items() if k.startswith("nnfm")}
        R.profile_enable(0)
        rec = {"positions": n, "hip_ms": round(timed(hip), 3), "kernels_ms": kern,
               "gemm_tflops_bf16": round(2.0 * n * n * 512 / (kern.get("nnfm_match", 1e9) * 1e-3) / 1e12, 1)}
        try:
            torch.cuda.reset_peak_memory_stats()
            rec["torch_fp32_ms"] = round(timed(torch_ref, iters=3), 3)
            rec["torch_peak_bytes"] = int(torch.cuda.max_memory_allocated())
            rec["loss_hip_vs_torch"] = [float(loss_nnfm_style(f1, f2)), float(ref(f1, f2))]
        except RuntimeError as e:                      # the N x N matrix may simply not fit next to its gradient
            rec["torch_fp32_ms"] = None
            rec["torch_error"] = str(e)[:120]
        out[name] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
