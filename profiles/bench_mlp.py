#!/usr/bin/env python
"""Secondary measurement (reported separately from views/s): the fused bf16-MFMA deformation MLP
forward at N Gaussians vs the same network in PyTorch fp32 (what the reference runs).
Algorithmic flops: 2 * 504 320 MAC/Gaussian = 1.009 MFLOP/Gaussian."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.deform import deform_forward
from trase_amd import rasterizer as R


class RefNet(torch.nn.Module):       # same layer shapes as utils/time_utils.py:60-104
    def __init__(self):
        super().__init__()
        self.linear = torch.nn.ModuleList([torch.nn.Linear(84, 256)] + [torch.nn.Linear(340 if i == 4 else 256, 256) for i in range(7)])
        self.gaussian_warp = torch.nn.Linear(256, 3)
        self.gaussian_rotation = torch.nn.Linear(256, 4)
        self.gaussian_scaling = torch.nn.Linear(256, 3)

    @staticmethod
    def pe(v, nf):
        out = [v]
        for f in range(nf):
            out += [torch.sin(v * 2.0 ** f), torch.cos(v * 2.0 ** f)]
        return torch.cat(out, -1)

    def forward(self, x, t):
        e = torch.cat([self.pe(x, 10), self.pe(t, 10)], -1)
        h = e
        for i, l in enumerate(self.linear):
            h = torch.relu(l(h))
            if i == 4:
                h = torch.cat([e, h], -1)
        return self.gaussian_warp(h), self.gaussian_rotation(h), self.gaussian_scaling(h)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = RefNet().to(dev)
    x = (torch.rand(n, 3, device=dev) * 2 - 1) * 1.3
    t = torch.tensor([[0.4]], device=dev).expand(n, -1)
    params = dict(net.state_dict())
    with torch.no_grad():
        for _ in range(3):
            a = deform_forward(params, x, t)
            b = net(x, t.contiguous())
        torch.cuda.synchronize()
        err = max((u - v).abs().max().item() for u, v in zip(a, b))
        R.profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(10):
            deform_forward(params, x, t)
        torch.cuda.synchronize()
        t_hip = (time.perf_counter() - t0) / 10
        prof = R.profile_report(); R.profile_enable(0)
        t0 = time.perf_counter()
        for _ in range(10):
            net(x, t.contiguous())
        torch.cuda.synchronize()
        t_ref = (time.perf_counter() - t0) / 10
    flops = 2 * 504320 * n
    k_ms = prof["mlp_fwd"]["ms"]
    print(json.dumps({"n": n, "mlp_fwd_kernel_ms": round(k_ms, 4), "mlp_call_ms": round(t_hip * 1e3, 4),
                      "tflops_kernel": round(flops / (k_ms * 1e-3) / 1e12, 2), "peak_bf16_tflops": 2500.0,
                      "frac_mfma_peak": round(flops / (k_ms * 1e-3) / 2.5e15, 4),
                      "torch_fp32_ms": round(t_ref * 1e3, 4), "max_abs_diff_vs_fp32": err}))


if __name__ == "__main__":
    main()
