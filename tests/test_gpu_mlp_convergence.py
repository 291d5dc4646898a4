"""Does training THROUGH the bf16 deformation MLP behave like training through the reference's fp32 network?  (VERDICT r5 item 2 /
weak 2.)  Per step the hidden-layer weight gradients of the MFMA path are up to ~12 % (relative L2) away from fp32 autograd of the
same parameters -- ReLU gates within bf16 rounding of zero flip, and dZ crosses the layers in bf16 -- and until this test the claim
that this does not matter rested on that per-step distance alone.

The loop is the reference's GAUSSIAN state after the warm-up (train.py:195-243, :299, :376-389) in miniature: six views of a scene
whose ground truth MOVES with time (a smooth, time-dependent displacement field the network has to learn), 320 iterations,
densification off, fixed seeds, Adam on the Gaussian parameters and on the network.  It runs twice from identical initial state --
once with the torch fp32 ``SynthDeformNetwork`` (the reference's composition, autograd), once with ``DeformNetworkHIP`` (bf16 MFMA
forward + backward) -- everything else (fused render, photometric loss, FusedAdam) identical.  The yardstick is a FAMILY: the
loop amplifies rounding-size noise (two fp32 runs whose initial network weights differ by 1e-6 relative noise end 3-10 % apart in
final loss and 11-15 % in learnt motion after 320 iterations; VERDICT's 2 % / 5 % are tighter than fp32 is to itself), so five fp32
runs (base + four perturbed) span the range training "like fp32" lands in, and three bf16 runs (base + two perturbed) must fall
inside it.  Probe of what the network has learnt: mean |d_xyz| over the six view times, evaluated every 10 iterations with the SAME
fp32 forward for every run (it measures the learnt function, not the evaluating kernel).  Asserted, for the final
photometric loss (mean of the last 24 iterations = four passes over the views) and the final probe: the families' means agree within
twice the standard error of their difference (+ 2 %), no bf16 run is an outlier of the fp32 family (3 sigma + 5 %); and every bf16
probe trajectory is no further from the fp32 base run than 1.5 x the furthest fp32 sibling.  All curves go
to gpurun_out/mlp_convergence.json (copied to profiles/r6_mlp_convergence.json)."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train(mode: str, iters: int = 320):
    use_hip = mode.startswith("bf16")
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
    if "perturbed" in mode:
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
    fam32 = [_train(m) for m in ("fp32", "fp32_perturbed", "fp32_perturbed_b", "fp32_perturbed_cc", "fp32_perturbed_ddd")]
    fam16 = [_train(m) for m in ("bf16", "bf16_perturbed", "bf16_perturbed_b")]
    tail = 24
    fin = lambda run: sum(run[0][-tail:]) / tail
    L32, L16 = [fin(r) for r in fam32], [fin(r) for r in fam16]
    P32, P16 = [r[1][-1] for r in fam32], [r[1][-1] for r in fam16]
    l32, d32, ext = fam32[0]
    head = sum(l32[:6]) / 6
    floor = 0.01 * d32[-1]
    traj = lambda d: max(abs(a - b) / a for a, b in zip(d32, d) if a > floor)
    T32, T16 = [traj(r[1]) for r in fam32[1:]], [traj(r[1]) for r in fam16]
    rec = {"iterations": len(l32), "views": 6, "gaussians": 8000, "image": [320, 192], "scene_extent": ext,
           "loss_first6_mean_fp32": head, "final_loss_fp32_family": L32, "final_loss_bf16_family": L16,
           "probe": "mean |d_xyz| over the six view times, fp32 forward, every 10 iterations",
           "final_probe_fp32_family": P32, "final_probe_bf16_family": P16,
           "probe_traj_max_rel_dist_from_fp32_base": {"fp32_family": T32, "bf16_family": T16},
           "probe_fp32": d32, "probe_bf16": fam16[0][1],
           "loss_curve_fp32": [round(v, 6) for v in l32], "loss_curve_bf16": [round(v, 6) for v in fam16[0][0]]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mlp_convergence.json"), "w") as f:
        json.dump(rec, f)
    print(json.dumps({k: v for k, v in rec.items() if not k.startswith("loss_curve") and not k.startswith("probe_")}))
    assert all(math.isfinite(v) for r in fam32 + fam16 for v in r[0])
    assert max(L32) < 0.8 * head, f"the fp32 runs did not learn ({head} -> {L32}): the test scene is not a test"
    assert d32[-1] > 5 * d32[0] and d32[-1] > 1e-3 * ext, f"the network did not move anything: {d32[0]} -> {d32[-1]}"
    # the two families' means agree within twice the standard error of their difference (+ 2 % of the fp32 mean), and no bf16 run is
    # further from the fp32 mean than three fp32 standard deviations (+ 5 %) -- final loss and learnt motion (mean of the last five
    # probe samples)
    import statistics as st
    P32 = [sum(r[1][-5:]) / 5 for r in fam32]
    P16 = [sum(r[1][-5:]) / 5 for r in fam16]
    rec["final_probe_fp32_family"], rec["final_probe_bf16_family"] = P32, P16
    for name, a32, a16 in (("final photometric loss", L32, L16), ("final mean |d_xyz|", P32, P16)):
        m32, m16, s32, s16 = st.mean(a32), st.mean(a16), st.stdev(a32), st.stdev(a16)
        se = math.sqrt(s32 * s32 / len(a32) + s16 * s16 / len(a16))
        assert abs(m16 - m32) <= 2 * se + 0.02 * m32, f"{name}: bf16 mean {m16:.6g} vs fp32 mean {m32:.6g} (standard error of the difference {se:.3g})"
        assert all(abs(v - m32) <= 3 * s32 + 0.05 * m32 for v in a16), f"{name}: a bf16 run of {a16} is an outlier of the fp32 family {a32}"
    # and their trajectories no further from the fp32 base run than 1.5 x the furthest fp32 sibling (5 % at least)
    bar = max(0.05, 1.5 * max(T32))
    assert max(T16) <= bar, f"probe trajectories: bf16 runs are {T16} from the fp32 base, its fp32 siblings {T32} (bar {bar:.3f})"
