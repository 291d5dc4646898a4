"""Combination sweep (round 5, after a GPU memory fault in a combination no test had touched: tile-row strips through the operator
entry point's backward): the operator entry point and the fused render() under random combinations of tile-row strip / capacity
policy / launch-graph replay / feature width / cotangent scope / variant bits (32-bit depth keys, packed-FP32 forward or backward,
slot lists), each against the plain call (synchronising policy, no replay, default kernels) of the same scene.  Different kernels
sum in different orders: 2e-4 of every tensor's scale (the image-only MFMA backward against the packed-FP32 one reaches 1.1e-4 on a
one-Gaussian scene); what this sweep is for is the crash or the garbage a wrong combination produces.  1 000 further combinations
were run once with scratch seeds (profiles/r5_ab_experiments.txt)."""
import contextlib
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

VAR = {"depth32": 0x400000, "valu_fwd": 0x2000, "valu_bwd": 0x40, "slot_lists": 0x100000}
H, W = 112, 176          # 7 x 11 tiles


def _op_call(scene, cam, feat, rows, cot, gi, gf):
    from tests import test_gpu_parity as T
    from tests.util import settings_for
    from trase_amd import rasterizer as R
    act = scene.activated()
    if feat == 0:
        act["sh_objs"] = None
    elif feat == 16:
        act["sh_objs"] = act["sh_objs"][..., :16].contiguous()
    st = settings_for(cam)
    with (R.tile_rows(*rows) if rows else contextlib.nullcontext()):
        out, leaves = T._gpu_call(act, st)
        outs, cots = [out[0]], [gi]
        if cot == "both" and feat:
            outs.append(out[2]); cots.append(gf[:feat])
        torch.autograd.backward(outs, cots)
    res = [out[0].detach().clone(), out[3].detach().clone(), out[1].clone().float()]
    res += [leaves[k].grad.clone() for k in ("means3D", "opacities", "scales", "rotations", "shs", "means2D")]
    if feat and cot == "both":
        res.append(leaves["sh_objs"].grad.clone())
    return res


def _fused_call(scene, cam, rows, cot, gi, gf):
    from gaussian_renderer import render
    from trase_amd import rasterizer as R
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe
    dev = gi.device
    pc = SynthGaussianModel(scene.to(dev))
    bg = torch.zeros(3, device=dev)
    with (R.tile_rows(*rows) if rows else contextlib.nullcontext()):
        out = render(cam.to(dev), pc, SynthPipe(), bg, 0.0, 0.0, 0.0)
        outs, cots = [out["render"]], [gi]
        if cot == "both":
            outs.append(out["render_gaussian_features"]); cots.append(gf)
        torch.autograd.backward(outs, cots)
    res = [out["render"].detach().clone(), out["depth"].detach().clone(), out["radii"].clone().float()]
    res += [p.grad.clone() if p.grad is not None else None for p in pc.parameters()]
    return res


def _close(a, b, tag):
    for i, (x, y) in enumerate(zip(a, b)):
        if x is None or y is None:
            assert x is None and y is None, (tag, i)
            continue
        s = float(y.abs().max())
        d = float((x - y).abs().max())
        assert d <= 2e-4 * max(s, 1e-12) + 2e-6, f"{tag}: tensor {i} differs by {d} (scale {s})"


@pytest.mark.parametrize("seed", [1, 4])
def test_random_combinations_agree_with_the_plain_call(seed):
    from trase_amd import rasterizer as R
    from trase_amd.synthetic import make_scene, orbit_camera
    rng = random.Random(seed)
    dev = torch.device("cuda", 0)
    for it in range(40):
        n = rng.choice([1, 50, 700, 3000])
        sd = rng.randrange(100)
        scene = make_scene(n, feat_dim=32, seed=sd, scale_mult=rng.choice([0.5, 0.9, 1.5]))
        cam = orbit_camera(W, H, angle=rng.random() * 3.0)
        entry = rng.choice(["op", "op", "fused"])
        feat = 32 if entry == "fused" else rng.choice([0, 16, 32])
        rows = rng.choice([None, None, (0, 3), (2, 5), (6, 7), (3, 4)])
        cot = "image" if feat == 0 else rng.choice(["both", "image"])
        sync = rng.choice([True, False])
        graph = rng.choice([False, True, "auto"])
        var = 0
        for k, b in VAR.items():
            if rng.random() < 0.25:
                var |= b
        g = torch.Generator().manual_seed(sd)
        gi = torch.randn(3, H, W, generator=g).to(dev)
        gf = torch.randn(32, H, W, generator=g).to(dev)
        tag = f"[{it}] entry={entry} n={n} seed={sd} feat={feat} rows={rows} cot={cot} sync={sync} graph={graph} var={hex(var)}"
        print(tag, flush=True)              # (-s: a GPU fault kills the process; the last line names the combination)
        call = (lambda: _op_call(scene, cam, feat, rows, cot, gi, gf)) if entry == "op" else (lambda: _fused_call(scene, cam, rows, cot, gi, gf))
        R.set_sync(True); R.set_graph(False); R.set_variant(0)
        base = call()
        cap = max(R.last_status()[2], 1)
        try:
            R.set_variant(var)
            R.set_graph(graph)
            if not sync:
                R.set_sync(False, capacity=2 * cap + 1024)
            got = call()
            if graph:
                got = call()                # the second call of an identical record may replay
            if not sync:
                R.check_overflow()
            torch.cuda.synchronize()
            _close(got, base, tag)
        finally:
            R.set_sync(True); R.set_graph("auto"); R.set_variant(0)


def _close_r(a, b, tag, tol=5e-4):
    for i, (x, y) in enumerate(zip(a, b)):
        if x is None or y is None:
            z = x if x is not None else y
            assert z is None or float(z.abs().max()) == 0.0, (tag, i, x is None, y is None)
            continue
        assert x.shape == y.shape, (tag, i, x.shape, y.shape)
        if x.numel() == 0:
            continue
        s = float(y.abs().max())
        d = float((x - y).abs().max())
        assert d <= tol * max(s, 1e-12) + 2e-6, f"{tag}: tensor {i} differs by {d} (scale {s})"


@pytest.mark.parametrize("seed", [3, 8])
def test_random_render_option_combinations_agree_with_the_plain_call(seed):
    """The fused render() under random combinations of: deformation tensors, forward scope (image), backward scope (features),
    tile-row strips with and without sparse strip gradients, the gradient sink with a chunked tail, lineage switches (feature
    background, normalised depth, depth gradient), feature normalisation, variant bits, capacity policy, launch-graph replay and
    two views per launch sequence -- against the plain call.  Round 5: this sweep found gradients silently DOUBLED when the sink
    was set while ``.grad`` still held the bucket's slices, and the features-only scope ignored under the packed-FP32 backward.
    Tolerance 5e-4 of scale (packed-FP32 against MFMA backward with a normalised-depth gradient on a one-Gaussian scene: 3.2e-4)."""
    from gaussian_renderer import render
    from trase_amd import rasterizer as R
    from trase_amd import renderer as RR
    from trase_amd.dp import FlatGradBucket
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    rng = random.Random(seed)
    dev = torch.device("cuda", 0)
    close = _close_r
    for it in range(40):
        n = rng.choice([1, 60, 800, 3000])
        sd = rng.randrange(100)
        scene = make_scene(n, feat_dim=32, seed=sd, scale_mult=rng.choice([0.5, 0.9, 1.5])).to(dev)
        cams = [orbit_camera(W, H, angle=rng.random() * 3.0).to(dev) for _ in range(2)]
        deform = rng.random() < 0.5
        fscope = rng.choice(["all", "all", "image"])
        bscope = "all" if fscope == "image" else rng.choice(["all", "all", "features"])
        rows = rng.choice([None, None, None, (0, 3), (2, 5), (6, 7)])
        sparse = bool(rows) and rng.random() < 0.5
        chunks = 1 if (rows or rng.random() < 0.6) else rng.choice([2, 3, 5])
        pair = (not rows) and chunks == 1 and rng.random() < 0.25
        norm = rng.random() < 0.5
        lin = dict(feats_bg=rng.choice([None, 0.3]), depth_normalised=rng.random() < 0.3, depth_grad=rng.random() < 0.3)
        cot = "image" if fscope == "image" else rng.choice(["both", "both", "image"])
        if bscope == "features":
            cot = "both"
        sync = False if pair else rng.choice([True, False])
        graph = rng.choice([False, True, "auto"])
        var = 0
        for k, b in VAR.items():
            if rng.random() < 0.2:
                var |= b
        g = torch.Generator().manual_seed(sd)
        gi = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(2)]
        gf = [torch.randn(32, H, W, generator=g).to(dev) for _ in range(2)]
        gd = [torch.randn(1, H, W, generator=g).to(dev) for _ in range(2)]
        dd = [0.01 * torch.randn(n, c, generator=g).to(dev) for c in (3, 4, 3)]
        tag = (f"[{it}] n={n} seed={sd} deform={deform} fscope={fscope} bscope={bscope} rows={rows} sparse={sparse} chunks={chunks} pair={pair} "
               f"norm={norm} lin={lin} cot={cot} sync={sync} graph={graph} var={hex(var)}")
        print(tag, flush=True)

        def run(plain):
            pc = SynthGaussianModel(scene)
            bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
            d = [t.clone().requires_grad_(True) for t in dd] if deform else [0.0, 0.0, 0.0]
            views = cams if pair else cams[:1]
            bucket = None
            if not plain and chunks > 1:
                bucket = FlatGradBucket(pc.parameters())
                RR.set_grad_sink(**bucket.overlapped(chunks))
            try:
                RR.set_forward_scope("all" if plain else fscope)
                with (R.tile_rows(*rows) if rows else contextlib.nullcontext()):
                    if pair and not plain:
                        outs = RR.render_views(views, pc, SynthPipe(), bg, *d, norm_gaussian_features=norm)
                    else:
                        outs = [render(c, pc, SynthPipe(), bg, *d, norm_gaussian_features=norm) for c in views]
                    ts, cs = [], []
                    for k, o in enumerate(outs):
                        ts.append(o["render"]); cs.append(gi[k])
                        if cot == "both":
                            ts.append(o["render_gaussian_features"]); cs.append(gf[k])
                        if lin["depth_grad"]:
                            ts.append(o["depth"]); cs.append(gd[k].reshape(o["depth"].shape))
                    torch.autograd.backward(ts, cs)
            finally:
                RR.set_forward_scope("all")
                RR.set_grad_sink(None)
            res = []
            for o in outs:
                res += [o["render"].detach().clone(), o["depth"].detach().clone(), o["radii"].clone().float()]
                if (plain or fscope == "all"):
                    res.append(o["render_gaussian_features"].detach().clone())
            grads = [p.grad.clone() if p.grad is not None else None for p in pc.parameters()]
            dgr = [t.grad.clone() if (deform and t.grad is not None) else None for t in (d if deform else [])]
            vsp = [o["viewspace_points"].grad.clone() if o["viewspace_points"].grad is not None else None for o in outs]
            return res, grads, dgr, vsp, pc

        R.set_sync(True); R.set_graph(False); R.set_variant(0); R.set_lineage(**lin); RR.set_backward_scope("all"); R.set_sparse_strip_grads(False)
        base_var = R._Policy.variant
        b_res, b_grads, b_dgr, b_vsp, b_pc = run(True)
        cap = max(R.last_status()[2], 1)
        try:
            R.set_variant(base_var | var)
            if bscope == "features":
                RR.set_backward_scope("features")
            R.set_sparse_strip_grads(sparse)
            R.set_graph(graph)
            if not sync:
                R.set_sync(False, capacity=2 * cap + 1024)
            got = run(False)
            if graph:
                got = run(False)
            if not sync:
                R.check_overflow()
            torch.cuda.synchronize()
            g_res, g_grads, g_dgr, g_vsp, g_pc = got
            if fscope == "image":        # the plain call also returned the feature map: drop it from the comparison
                keep = [x for k, x in enumerate(b_res) if k % 4 != 3]
                close(g_res, keep, tag + " outputs")
            else:
                close(g_res, b_res, tag + " outputs")
            if bscope == "features":
                names = [nm for nm, _ in zip(["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "gfeat"], g_grads)]
                gi_ = len(g_grads) - 1     # the Gaussian features are the last parameter of the synthetic model
                close([g_grads[gi_]], [b_grads[gi_]], tag + " feature grad")
                for k, x in enumerate(g_grads[:gi_]):
                    assert x is None or float(x.abs().max()) == 0.0, f"{tag}: parameter {k} has a gradient under the features-only scope"
            else:
                close(g_grads, b_grads, tag + " grads")
                close(g_dgr, b_dgr, tag + " deformation grads")
                close(g_vsp, b_vsp, tag + " viewspace grads")
        finally:
            R.set_sync(True); R.set_graph("auto"); R.set_variant(0); R.set_lineage(); RR.set_backward_scope("all"); R.set_sparse_strip_grads(False)


def test_degenerate_gaussians_neither_fault_nor_hang():
    """NaN / inf / 1e30 positions, zero, negative, NaN and 1e6 scales, zero and NaN quaternions, opacity 0 / 1 / NaN, NaN colours and
    features, Gaussians at and behind the camera -- on 2 %, 20 % or all of the Gaussians, under both capacity policies: the call
    completes (NaNs may come out; the reference's kernels tolerate such inputs too), radii are never negative and the binned pair
    count stays within Gaussians x sub-tiles.  360 cases of this sweep ran once with scratch seeds."""
    from tests import test_gpu_parity as T
    from tests.util import settings_for, small_case
    from trase_amd import rasterizer as R
    rng = random.Random(2)
    Hh, Ww = 96, 160
    kinds = ["nan_pos", "inf_pos", "zero_scale", "huge_scale", "zero_quat", "op0", "op1", "nan_op", "nan_feat", "at_camera", "behind",
             "nan_scale", "neg_scale", "nan_quat", "huge_pos", "nan_sh"]
    for it in range(30):
        n = rng.choice([1, 40, 600, 2500])
        act, cam = small_case(n=n, w=Ww, h=Hh, feat=32, seed=rng.randrange(100))
        frac = rng.choice([0.02, 0.2, 1.0])
        chosen = rng.sample(kinds, rng.choice([1, 2, 4]))
        g = torch.Generator().manual_seed(it)
        cc = cam.camera_center.reshape(1, 3)
        for kind in chosen:
            m = torch.rand(n, generator=g) < frac
            if kind == "nan_pos": act["means3D"][m] = float("nan")
            if kind == "inf_pos": act["means3D"][m] = float("inf")
            if kind == "huge_pos": act["means3D"][m] = 1e30
            if kind == "zero_scale": act["scales"][m] = 0.0
            if kind == "huge_scale": act["scales"][m] = 1e6
            if kind == "nan_scale": act["scales"][m] = float("nan")
            if kind == "neg_scale": act["scales"][m] = -act["scales"][m]
            if kind == "zero_quat": act["rotations"][m] = 0.0
            if kind == "nan_quat": act["rotations"][m] = float("nan")
            if kind == "op0": act["opacities"][m] = 0.0
            if kind == "op1": act["opacities"][m] = 1.0
            if kind == "nan_op": act["opacities"][m] = float("nan")
            if kind == "nan_feat": act["sh_objs"][m] = float("nan")
            if kind == "nan_sh": act["shs"][m] = float("nan")
            if kind == "at_camera": act["means3D"][m] = cc.expand(int(m.sum()), 3)
            if kind == "behind": act["means3D"][m] = cc + (cc - act["means3D"][m])
        sync = rng.choice([True, False])
        print(f"[{it}] n={n} frac={frac} kinds={chosen} sync={sync}", flush=True)
        st = settings_for(cam)
        gi = torch.randn(3, Hh, Ww, generator=g).cuda()
        gf = torch.randn(32, Hh, Ww, generator=g).cuda()
        R.set_sync(True)
        try:
            if not sync:
                T._gpu_call(act, st, need_grad=False)
                R.set_sync(False, capacity=2 * max(R.last_status()[2], 1) + 1024)
            out, leaves = T._gpu_call(act, st)
            torch.autograd.backward([out[0], out[2]], [gi, gf])
            if not sync:
                R.check_overflow()
            torch.cuda.synchronize()
            assert int(out[1].min()) >= 0
            if sync:
                assert 0 <= R.last_status()[2] <= n * ((Hh // 8 + 1) * (Ww // 8 + 1))
        finally:
            R.set_sync(True)
