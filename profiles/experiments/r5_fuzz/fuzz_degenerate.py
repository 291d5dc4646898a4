"""Crash-safety with degenerate Gaussians (the reference's kernels tolerate them -- NaNs may come out, faults and hangs may not):
NaN / inf positions, zero and huge scales, zero quaternions, opacity 0 / 1 / NaN, NaN features, Gaussians at and behind the camera."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import settings_for, small_case
from tests import test_gpu_parity as T
from trase_amd import rasterizer as R
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = random.Random(seed)
H, W = 96, 160
kinds = ["nan_pos", "inf_pos", "zero_scale", "huge_scale", "zero_quat", "op0", "op1", "nan_op", "nan_feat", "at_camera", "behind", "nan_scale", "neg_scale", "nan_quat", "huge_pos", "nan_sh"]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    n = rng.choice([1, 40, 600, 2500])
    act, cam = small_case(n=n, w=W, h=H, feat=32, seed=rng.randrange(100))
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
    gi = torch.randn(3, H, W, generator=g).cuda(); gf = torch.randn(32, H, W, generator=g).cuda()
    R.set_sync(True)
    try:
        if not sync:
            out, leaves = T._gpu_call(act, st, need_grad=False)
            R.set_sync(False, capacity=2 * max(R.last_status()[2], 1) + 1024)
        out, leaves = T._gpu_call(act, st)
        torch.autograd.backward([out[0], out[2]], [gi, gf])
        if not sync:
            try:
                R.check_overflow()
            except RuntimeError as e:
                print("   overflow reported:", str(e)[:100])
        torch.cuda.synchronize()
        radii = out[1]
        if int(radii.min()) < 0:
            print("   NEGATIVE radius", int(radii.min()), flush=True)
        st_ = R.last_status() if sync else None
        if st_ is not None:
            assert 0 <= st_[2] <= n * ((H // 8 + 1) * (W // 8 + 1)), st_
    except (RuntimeError, ValueError) as e:
        print("   raised:", type(e).__name__, str(e)[:160], flush=True)
    finally:
        R.set_sync(True)
print("done")
