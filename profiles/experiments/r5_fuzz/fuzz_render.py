"""Combination fuzz of the fused render(): deformation tensors, forward / backward scope, strips (+ sparse strip gradients), gradient
sink with chunked tail, lineage switches, feature normalisation, variant bits, policy, graph replay, two views per launch sequence --
each against the plain call of the same scene.  Prints every combination before running it."""
import sys, os, contextlib, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trase_amd import rasterizer as R
from trase_amd import renderer as RR
from trase_amd.dp import FlatGradBucket
from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
from gaussian_renderer import render

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
dev = torch.device("cuda", 0)
VAR = {"depth32": 0x400000, "valu_fwd": 0x2000, "valu_bwd": 0x40, "slot_lists": 0x100000}
H, W = 112, 176


def close(a, b, tag, tol=2e-4):
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


bad = 0
for it in range(count):
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
    if os.environ.get("FUZZ_ONLY") and it != int(os.environ["FUZZ_ONLY"]):
        continue

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

    R.set_sync(True); R.set_graph(False); R.set_variant(int(os.environ.get("FUZZ_VAR_BOTH", "0"), 0)); R.set_lineage(**lin); RR.set_backward_scope("all"); R.set_sparse_strip_grads(False)
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
    except AssertionError as e:
        bad += 1
        print("   MISMATCH", e, flush=True)
        if os.environ.get("FUZZ_ONLY"):
            for k, (x, y) in enumerate(zip(g_grads, b_grads)):
                if x is not None and y is not None and x.numel() <= 64:
                    print("    grad", k, x.flatten().tolist(), y.flatten().tolist(), flush=True)
    finally:
        R.set_sync(True); R.set_graph("auto"); R.set_variant(0); R.set_lineage(); RR.set_backward_scope("all"); R.set_sparse_strip_grads(False)
print("done; mismatches:", bad)
