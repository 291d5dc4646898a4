"""FEATURE-state loss head (train.py:251-296) at 1080p, 100 SAM masks, ~5000 sampled pixels, ~50 sampled masks:
the reference's composition restated in PyTorch (S x S matrices; its pair losses replaced by this repo's fused ones, so
the comparison is conservative) vs trase_amd.feature_head (no S x S matrix).  Prints one JSON line.

    python profiles/bench_feature_head.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trase_amd.feature_head import contrastive_head, feature_norm_reg, get_sample_pixel_and_mask, mask_stats  # noqa: E402
from trase_amd.losses import negative_pixel_pair_loss, positive_pixel_pair_loss  # noqa: E402


def ref_sample(sam_masks, nsp=5000, nsm=50):                 # utils/feature_utils.py:17-26
    sampled_mask = torch.rand(sam_masks.shape[0]).cuda() < nsm / sam_masks.shape[0]
    rate = nsp / (sam_masks.shape[-1] * sam_masks.shape[-2])
    sampled_pixel = torch.rand(sam_masks.shape[-2], sam_masks.shape[-1]).cuda() < rate
    non_mask_region = sam_masks.sum(dim=0) == 0
    return torch.logical_and(sampled_pixel, ~non_mask_region), sampled_mask


def ref_weights(sam_masks, sampled_pixel):                    # utils/feature_utils.py:28-38
    per_pixel_mask_size = sam_masks * sam_masks.sum(-1).sum(-1)[:, None, None]
    m = per_pixel_mask_size.sum(dim=0) / (sam_masks.sum(dim=0) + 1e-9)
    m = m[sampled_pixel]
    pp = m.unsqueeze(0) * m.unsqueeze(1)
    mx = pp.max()
    pp[pp == 0] = 1e10
    w = torch.clamp(mx / pp, 1.0, None)
    return (w - w.min()) / (w.max() - w.min()) * 9. + 1.


def ref_cmat(sam_masks, sampled_pixel, sampled_mask):         # utils/feature_utils.py:40-49
    v = sam_masks[:, sampled_pixel][sampled_mask, :]
    c = torch.einsum("nh,nj->hj", v.float(), v.float())
    c[c != 0] = 1
    return c


def ref_cf(feats, sampled_pixel):                             # utils/feature_utils.py:51-57
    f = torch.nn.functional.normalize(feats[:, sampled_pixel].permute([1, 0]), dim=-1, p=2)
    return torch.einsum("hc,jc->hj", f, f)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t) / n * 1e3, 3)


def main():
    torch.manual_seed(0)
    N, H, W, F = 100, 1080, 1920, 32
    g = torch.Generator().manual_seed(0)
    sam = torch.zeros(N, H, W, dtype=torch.bool, device="cuda")
    for n in range(N):
        y0, x0 = int(torch.randint(0, H - 50, (1,), generator=g)), int(torch.randint(0, W - 50, (1,), generator=g))
        h, w = int(torch.randint(40, 500, (1,), generator=g)), int(torch.randint(40, 700, (1,), generator=g))
        sam[n, y0:y0 + h, x0:x0 + w] = True
    feat = torch.randn(F, H, W, device="cuda", requires_grad=True)
    out = {"masks": N, "H": H, "W": W}
    sp, sm = ref_sample(sam)
    out["S"], out["sampled_masks"] = int(sp.sum()), int(sm.sum())

    def ref_iter():
        feat.grad = None
        sp_, sm_ = ref_sample(sam)
        C = ref_cmat(sam, sp_, sm_)
        reg = (1 - feat.norm(dim=0, p=2).mean()) ** 2
        CF = ref_cf(feat, sp_)
        Wt = ref_weights(sam, sp_)
        loss = positive_pixel_pair_loss["soft"](C=C, C_F=CF, positive_th=0.75, weights=Wt) + \
            negative_pixel_pair_loss["soft"](C=C, C_F=CF, negative_th=0.5, weights=Wt) + reg
        with torch.no_grad():
            ps, ns = CF[C == 1].mean(), CF[C == 0].mean()
        loss.backward()

    def hip_iter(rng):
        feat.grad = None
        cover, size = mask_stats(sam)
        sp_, sm_ = get_sample_pixel_and_mask(sam, 5000, 50, cover_count=cover, rng=rng)
        lp, ln, ps, ns, reg = contrastive_head(feat, sam, sp_, sm_, "soft", 0.75, 0.5, mask_size=size, with_norm_reg=True)
        (lp + ln + reg).backward()

    out["torch_head_ms"] = timed(ref_iter)
    out["hip_head_cpu_rng_ms"] = timed(lambda: hip_iter("cpu"))
    out["hip_head_ms"] = timed(lambda: hip_iter("cuda"))
    # pieces of the fused head
    cover, size = mask_stats(sam)
    out["mask_stats_ms"] = timed(lambda: mask_stats(sam))
    out["sampler_cpu_rng_ms"] = timed(lambda: get_sample_pixel_and_mask(sam, 5000, 50, cover_count=cover))
    out["sampler_cuda_rng_ms"] = timed(lambda: get_sample_pixel_and_mask(sam, 5000, 50, cover_count=cover, rng="cuda"))

    def pair_only():
        feat.grad = None
        lp, ln, _, _ = contrastive_head(feat, sam, sp, sm, "soft", 0.75, 0.5, mask_size=size)
        (lp + ln).backward()
    out["pair_head_fwd_bwd_ms"] = timed(pair_only)

    def reg_only():
        feat.grad = None
        feature_norm_reg(feat).backward()
    out["norm_reg_fwd_bwd_ms"] = timed(reg_only)
    out["speedup"] = round(out["torch_head_ms"] / out["hip_head_ms"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
