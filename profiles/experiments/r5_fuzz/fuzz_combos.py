"""Combination fuzz: the operator entry point and the fused render() under random combinations of strip / policy / graph replay /
feature width / cotangent scope / variant bits, against a plain baseline of the same call.  Prints every combination BEFORE running
it (a GPU fault kills the process: the last line names the culprit)."""
import sys, os, contextlib, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import settings_for
from tests import test_gpu_parity as T
from trase_amd import rasterizer as R
from trase_amd import renderer as RR
from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
from gaussian_renderer import render

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
dev = torch.device("cuda", 0)
VAR = {"depth32": 0x400000, "valu_fwd": 0x2000, "valu_bwd": 0x40, "slot_lists": 0x100000}
H, W = 112, 176          # 7 x 11 tiles


def op_call(scene, cam, feat, rows, cot, gi, gf):
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


def fused_call(scene, cam, rows, cot, gi, gf):
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


def close(a, b, tag):
    for i, (x, y) in enumerate(zip(a, b)):
        if x is None or y is None:
            assert x is None and y is None, (tag, i)
            continue
        s = float(y.abs().max())
        d = float((x - y).abs().max())
        assert d <= 5e-5 * max(s, 1e-12) + 2e-6, f"{tag}: tensor {i} differs by {d} (scale {s})"


bad = 0
for it in range(count):
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
    print(f"[{it}] entry={entry} n={n} seed={sd} feat={feat} rows={rows} cot={cot} sync={sync} graph={graph} var={hex(var)}", flush=True)
    call = (lambda: op_call(scene, cam, feat, rows, cot, gi, gf)) if entry == "op" else (lambda: fused_call(scene, cam, rows, cot, gi, gf))
    # baseline: sync policy, no graph, default variant
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
            got = call()                    # the second call of an identical record may replay
        if not sync:
            R.check_overflow()
        torch.cuda.synchronize()
        close(got, base, f"combo {it}")
    except AssertionError as e:
        bad += 1
        print("   MISMATCH", e, flush=True)
    finally:
        R.set_sync(True); R.set_graph("auto"); R.set_variant(0)
print("done; mismatches:", bad)
