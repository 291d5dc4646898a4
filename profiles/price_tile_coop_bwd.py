#!/usr/bin/env python
"""Pricing of VERDICT r5 item 6 (tile-cooperative backward): how many gradient rows would remain if the four 8x8 sub-tiles of a 16x16
tile merged the rows of the Gaussians they share before the store?  Counts, for S4 orbit views, the (sub-tile, Gaussian) pairs R_sub
(= rows written today) and the distinct (16x16 tile, Gaussian) pairs R_tile among them (= rows after a perfect in-tile merge), from the
forward's own lists.  python profiles/price_tile_coop_bwd.py > profiles/r6_tile_coop_pricing.json"""
import json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd import rasterizer as R
from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
from gaussian_renderer import render
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda", 0)
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev), requires_grad=False)
pipe, bg = SynthPipe(), torch.zeros(3, device=dev)
R.set_variant(R.VARIANT_SLOT_LISTS * 0)
gx8, gy8 = (W + 7) // 8, (H + 7) // 8
gx16 = (W + 15) // 16
T = gx8 * gy8
rows = []
for k in range(0, 16, 4):
    cam = orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev)
    R.set_sync(True)
    o = render(cam, pc, pipe, bg, 0.0, 0.0, 0.0)          # (grad mode on: the backward-only per-pixel state is stored)
    img_ws = o["render"].grad_fn.saved_tensors[13]
    wh = (4 * W * H + 255) // 256 * 256
    n_contrib = img_ws[wh: wh + 4 * W * H].view(torch.int32).reshape(H, W).to(torch.int64)
    cap = int(R._Policy.last_capacity)
    hdr = R._Policy.last_geom[:256].view(torch.int32)
    jb = int(hdr[3])
    assert jb > 0, "packed list values expected (id << jb | pair index)"
    binb = R._Policy.last_bin
    a = (4 * cap + 255) // 256 * 256
    plist = binb[a: a + 4 * cap].view(torch.int32).to(torch.int64) & 0xffffffff       # the sorted list values (BinBuf::pair_slot)
    rng = binb[2 * a: 2 * a + 8 * T].view(torch.int32).reshape(T, 2).to(torch.int64)
    lens = (rng[:, 1] - rng[:, 0]).clamp_min(0)
    r_sub = int(lens.sum())
    sub = torch.repeat_interleave(torch.arange(T, device=dev), lens)
    # entries of sub-tile t are plist[rng[t,0] : rng[t,1]]; lists are laid out in sub-tile order, so a running index works when starts are cumulative
    idx = torch.repeat_interleave(rng[:, 0], lens) + (torch.arange(r_sub, device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens))
    gid = plist[idx] >> jb
    local = torch.arange(r_sub, device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
    # walked entries only (rows are written for those): up to the sub-tile's last contributor of any of its 64 pixels
    pad = torch.zeros(gy8 * 8, gx8 * 8, dtype=torch.int64, device=dev); pad[:H, :W] = n_contrib
    last = pad.reshape(gy8, 8, gx8, 8).amax(dim=(1, 3)).reshape(-1)
    keep = local < torch.repeat_interleave(last, lens)
    r_binned = r_sub
    sub, gid = sub[keep], gid[keep]
    r_sub = int(keep.sum())
    sy, sx = sub // gx8, sub % gx8
    tile = (sy // 2) * gx16 + sx // 2
    key = tile * (1 << 20) + gid
    r_tile = int(torch.unique(key).numel())
    # how the shared Gaussians split: number of sub-tiles (1..4) a (tile, Gaussian) pair has
    _, cnt = torch.unique(key, return_counts=True)
    hist = [int((cnt == c).sum()) for c in (1, 2, 3, 4)]
    rows.append({"view": k, "binned_pairs": r_binned, "R_sub": r_sub, "R_tile": r_tile, "ratio": round(r_tile / r_sub, 4), "pairs_with_1_2_3_4_subtiles": hist})
r_sub = sum(r["R_sub"] for r in rows) / len(rows); r_tile = sum(r["R_tile"] for r in rows) / len(rows)
row_bytes = (F + 12) * 4
out = {"workload": f"{N} Gaussians {W}x{H} F={F}, orbit views 0, 4, 8, 12", "views": rows, "R_sub_mean": r_sub, "R_tile_mean": r_tile,
       "rows_after_perfect_merge_over_rows_today": round(r_tile / r_sub, 4),
       "row_bytes": row_bytes, "render_bwd_write_bytes_saved": int((r_sub - r_tile) * row_bytes),
       "reduce_rows_read_bytes_saved": int((r_sub - r_tile) * row_bytes)}
print(json.dumps(out))
