#!/usr/bin/env python
"""Secondary measurement: the photometric loss of train.py:235-238, (1-l)*L1 + l*(1 - SSIM), forward + backward on a
(3,1080,1920) image -- fused HIP kernels (trase_amd.losses) vs the reference's PyTorch composition
(utils/loss_utils.py:30-86: five depthwise 11x11 conv2d + autograd)."""
import sys, os, time, json, math
import torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.losses import l1_ssim, photometric_loss
from trase_amd import rasterizer as R


def ref_ssim(x, y, win):
    conv = lambda t: Fn.conv2d(t, win, padding=5, groups=x.shape[0])
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()


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
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    x = torch.rand(3, 1080, 1920, device=dev, requires_grad=True)
    y = (x.detach() + 0.1 * torch.randn(3, 1080, 1920, device=dev)).clamp(0, 1)
    g = torch.tensor([math.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    win = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous().to(dev)

    def ref():
        x.grad = None
        (0.8 * (x - y).abs().mean() + 0.2 * (1 - ref_ssim(x, y, win))).backward()

    def hip():
        x.grad = None
        l1, ss = l1_ssim(x, y)
        (0.8 * l1 + 0.2 * (1 - ss)).backward()

    def hip_one_node():
        x.grad = None
        photometric_loss(x, y, 0.2).backward()

    ref(); g_ref = x.grad.clone()
    hip(); g_hip = x.grad.clone()
    R.profile_enable(1)
    for _ in range(5):
        hip()
    prof = R.profile_report(); R.profile_enable(0)
    print(json.dumps({"shape": [3, 1080, 1920], "hip_fwd_bwd_ms": round(timed(hip), 4), "hip_one_node_fwd_bwd_ms": round(timed(hip_one_node), 4),
                      "hip_fwd_bwd_ms_again": round(timed(hip), 4), "hip_one_node_fwd_bwd_ms_again": round(timed(hip_one_node), 4), "torch_fwd_bwd_ms": round(timed(ref), 4),
                      "kernels_ms": {k: round(v["ms"], 4) for k, v in prof.items()},
                      "max_rel_grad_diff": float((g_hip - g_ref).abs().max() / g_ref.abs().max())}))


if __name__ == "__main__":
    main()
