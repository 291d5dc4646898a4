"""Parity at the sizes that matter.

1. A seeded random sweep of small scenes (100+ configurations: Gaussian count, image shape incl. ragged
   and one-tile-high images, splat size from sub-pixel to screen-filling) -- forward maps and every gradient
   against the float64 oracle.  It includes the three configurations that exceeded the old whole-tile
   fragility budget in round 1.
2. Every BASELINE single-GPU workload (S2, S3, S4, and one rank's share of S5) at FULL size against the
   oracle on a seeded sample of ~1 % of the 16x16 tiles (plus the ragged corner tiles): maps at 1e-4 abs on
   those tiles, per-Gaussian gradients (all N rows) for a cotangent restricted to them.

The oracle sees the device's own float32 depth keys / radii / centres only to settle decisions that are
ambiguous between float32 and float64 (oracle/raster_oracle.py `device_view`); what remains excluded is the
per-pixel "gate within rounding of its threshold" set, budgeted at <= 1 % per scene and < 0.5 % overall."""
import math
import os
import random

import pytest
import torch

from tests import test_gpu_parity as T
from tests.util import settings_for, small_case

pytestmark = pytest.mark.gpu

ROUND1_FAILURES = [(3500, 160, 136, 9650, 0.3), (2000, 64, 136, 6788, 0.6), (3500, 64, 17, 3268, 3.0)]


def _sweep_configs(count=None, seed=None):
    # TRASE_SWEEP_COUNT / TRASE_SWEEP_SEED widen the sweep without editing the test (a 450-configuration run with seed 4242:
    # 0 failures, 0.15 % fragile pixels -- round 2)
    count = int(os.environ.get("TRASE_SWEEP_COUNT", 104)) if count is None else count
    seed = int(os.environ.get("TRASE_SWEEP_SEED", 20260928)) if seed is None else seed
    rnd = random.Random(seed)
    cfgs = list(ROUND1_FAILURES)
    while len(cfgs) < count:
        cfgs.append((rnd.choice([50, 333, 777, 2000, 3500]), rnd.choice([33, 64, 100, 160, 250]),
                     rnd.choice([17, 64, 90, 136]), rnd.randrange(10_000), rnd.choice([0.3, 0.6, 1.0, 1.8, 3.0])))
    return cfgs


def _one(n, w, h, seed, scale, feat=32):
    act, cam = small_case(n=n, w=w, h=h, feat=feat, seed=seed, scale_mult=scale, d_rot=0.05)
    st = settings_for(cam, bg=(0.1, 0.25, 0.4))
    g, gl = T._gpu_call(act, st)
    o, ol = T._oracle_call(act, st, gpu=g)
    n_frag, n_reg = T._check_maps(g, o)
    gen = torch.Generator().manual_seed(seed)
    gi = T._masked(torch.randn(3, h, w, generator=gen), o)
    gf = T._masked(torch.randn(feat, h, w, generator=gen), o)
    (o.image * gi.double()).sum().add((o.feats * gf.double()).sum()).backward()
    torch.autograd.backward([g[0], g[2]], [gi.cuda(), gf.cuda()])
    T._check_grads(gl, ol, o, ["means3D", "means2D", "opacities", "scales", "rotations", "shs", "sh_objs"])
    return n_frag, n_reg, int(o.frag_gauss.sum())


def test_random_parity_sweep():
    torch.set_num_threads(max(1, min(32, (torch.get_num_threads() or 1))))
    tot_frag = tot_pix = tot_g = 0
    failures = []
    for cfg in _sweep_configs():
        try:
            a, b, c = _one(*cfg)
            tot_frag, tot_pix, tot_g = tot_frag + a, tot_pix + b, tot_g + c
        except AssertionError as e:       # collect, so that one run reports every failing configuration
            failures.append((cfg, str(e)[:300]))
    assert not failures, f"{len(failures)} of {len(_sweep_configs())} configurations fail: {failures[:6]}"
    share = tot_frag / max(tot_pix, 1)
    print(f"sweep: {len(_sweep_configs())} configurations, fragile pixels {tot_frag} of {tot_pix} ({100 * share:.3f} %), "
          f"unresolved Gaussians {tot_g}")
    assert share < 0.005, f"fragile-pixel share of the sweep {share:.4f} >= 0.5 %"


def _sample_tiles(w, h, share, seed):
    gx, gy = (w + 15) // 16, (h + 15) // 16
    rnd = random.Random(seed)
    k = max(8, int(round(share * gx * gy)))
    tiles = {(0, 0), (gx - 1, 0), (0, gy - 1), (gx - 1, gy - 1), (gx // 2, gy // 2)}     # ragged right / bottom edges included
    while len(tiles) < k + 5:
        tiles.add((rnd.randrange(gx), rnd.randrange(gy)))
    return sorted(tiles)


FULL_SIZE = [
    ("S2 NeRF-DS size", 150_000, 480, 270, 0.05),
    ("S4 headline", 300_000, 1920, 1080, 0.01),
    ("S3 Neu3D size", 1_000_000, 1352, 1014, 0.01),
    ("S5 Immersive size (one rank's share)", 2_500_000, 1280, 960, 0.006),
]


@pytest.mark.parametrize("name,n,w,h,share", FULL_SIZE, ids=[c[0].split()[0] for c in FULL_SIZE])
def test_fullsize_sampled_tile_parity(name, n, w, h, share):
    """Full-size forward + backward on the GPU; the float64 oracle composites a seeded sample of tiles."""
    act, cam = small_case(n=n, w=w, h=h, feat=32, seed=0, scale_mult=0.27, angle=0.3)
    st = settings_for(cam, bg=(0.1, 0.25, 0.4))
    tiles = _sample_tiles(w, h, share, seed=n)
    g, gl = T._gpu_call(act, st)
    o, ol = T._oracle_call(act, st, gpu=g, tiles=tiles)
    n_frag, n_reg = T._check_maps(g, o, frag_budget=0.005)
    assert n_reg >= 0.9 * len(tiles) * 256 * 0.5
    gen = torch.Generator().manual_seed(n)
    gi = T._masked(torch.randn(3, h, w, generator=gen), o)
    gf = T._masked(torch.randn(32, h, w, generator=gen), o)
    (o.image * gi.double()).sum().add((o.feats * gf.double()).sum()).backward()
    torch.autograd.backward([g[0], g[2]], [gi.cuda(), gf.cuda()])
    T._check_grads(gl, ol, o, ["means3D", "means2D", "opacities", "scales", "rotations", "shs", "sh_objs"])
    touched = int((ol["opacities"].grad.abs().reshape(-1) > 0).sum())
    print(f"{name}: {len(tiles)} tiles, {n_reg} pixels compared, {n_frag} fragile ({100 * n_frag / n_reg:.3f} %), "
          f"{o.pairs_done} (tile,Gaussian) pairs composited of R = {o.num_rendered}, {touched} Gaussians with gradient")
    assert touched > 50
