#!/usr/bin/env python
"""Secondary measurement (reported separately from views/s): the fused bf16-MFMA deformation MLP at N
Gaussians -- inference forward and training forward+backward -- vs the same network in PyTorch fp32
(what the reference runs).  Algorithmic flops: forward 2 * 504 320 MAC/Gaussian = 1.009 MFLOP/Gaussian
(SURVEY.md 8d); backward = data chain (7 x 256 x 256 + 10 x 256 MAC) + parameter GEMMs (= forward MACs)."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.deform import deform_forward
from trase_amd.synthetic import SynthDeformNetwork
from trase_amd import rasterizer as R


def timed(fn, iters=10):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = SynthDeformNetwork().to(dev)
    x = (torch.rand(n, 3, device=dev) * 2 - 1) * 1.3
    t = torch.tensor([[0.4]], device=dev).expand(n, -1)
    tc = t.contiguous()
    params = dict(net.state_dict())
    live = dict(net.named_parameters())
    g = [torch.randn(n, c, device=dev) for c in (3, 4, 3)]

    def train_hip():
        net.zero_grad(set_to_none=True)
        torch.autograd.backward(deform_forward(live, x, t), g)

    def train_ref():
        net.zero_grad(set_to_none=True)
        torch.autograd.backward(net(x, tc), g)

    with torch.no_grad():
        a = deform_forward(params, x, t)
        b = net(x, tc)
        err = max((u - v).abs().max().item() for u, v in zip(a, b))
        t_hip = timed(lambda: deform_forward(params, x, t))
        t_ref = timed(lambda: net(x, tc))
        R.profile_enable(1)
        for _ in range(5):
            deform_forward(params, x, t)
        prof = R.profile_report(); R.profile_enable(0)
    tt_hip = timed(train_hip)
    tt_ref = timed(train_ref)
    R.profile_enable(1)
    for _ in range(5):
        train_hip()
    tprof = R.profile_report(); R.profile_enable(0)
    # the same step when the view culls a quarter of the Gaussians (exactly-zero cotangents; here the slab |y| > 0.975 of the
    # +-1.3 box -- the S4 orbit views cull 25 %, oracle-measured): the backward skips the dead 32-row tiles
    from trase_amd import deform as D
    g_all = g
    keep = (x[:, 1].abs() <= 0.975)[:, None].float()
    g = [v * keep for v in g_all]
    D.track_live_tiles(True)
    culled = {}
    for mode in ("morton", "none"):
        D.set_row_order(mode)
        ms = timed(train_hip)
        R.profile_enable(1)
        for _ in range(5):
            train_hip()
        cp = R.profile_report(); R.profile_enable(0)
        culled[mode] = {"train_step_ms": round(ms * 1e3, 4), "live_tiles": D.last_live_tiles(), "tiles": (n + 31) // 32,
                        "kernels_ms": {k: round(v["ms"], 4) for k, v in cp.items()}}
    D.set_row_order("morton"); D.track_live_tiles(False)
    g = g_all
    train_ref()
    want = {k: p.grad.clone() for k, p in live.items()}
    train_hip()
    gerr = max(float((p.grad - want[k]).abs().max() / want[k].abs().max()) for k, p in live.items())
    flops = 2 * 504320 * n
    bwd_data_flops = 2 * (7 * 256 * 256 + 16 * 256) * n
    k_ms = prof["mlp_fwd"]["ms"]
    out = {"n": n, "mlp_fwd_kernel_ms": round(k_ms, 4), "mlp_call_ms": round(t_hip * 1e3, 4),
           "tflops_kernel": round(flops / (k_ms * 1e-3) / 1e12, 2), "peak_bf16_tflops": 2500.0,
           "frac_mfma_peak": round(flops / (k_ms * 1e-3) / 2.5e15, 4),
           "torch_fp32_ms": round(t_ref * 1e3, 4), "max_abs_diff_vs_fp32": err,
           "train_step_ms": round(tt_hip * 1e3, 4), "torch_fp32_train_step_ms": round(tt_ref * 1e3, 4),
           "train_kernels_ms": {k: round(v["ms"], 4) for k, v in tprof.items()},
           "bwd_data_tflops": round(bwd_data_flops / (tprof["mlp_bwd_data"]["ms"] * 1e-3) / 1e12, 2),
           "max_rel_grad_diff_vs_fp32": gerr, "live_rows_frac_culled_case": round(float(keep.mean()), 4),
           "culled_quarter": culled}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
