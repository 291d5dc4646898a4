"""TEST INFRASTRUCTURE ONLY -- torch restatement of the FEATURE-state loss head (train.py:251-296), materialising the
S x S matrices exactly as the reference's helpers do:
  utils/feature_utils.py:17-26 sampler, :28-38 weights, :40-49 C, :51-57 C_F; utils/loss_utils.py:275-406 pair losses.

Pinned: tests/test_feature_head_oracle.py checks every function against tests/golden/feature_head.npz, produced by the
imported reference (tests/golden/make_golden.py, G8).  ``dtype`` selects float32 (the reference's arithmetic) or float64
(a tighter yardstick for the HIP path at sizes where float32 summation order matters).
"""
import torch


def sample_pixel_and_mask(sam_masks, num_sampled_pixels, num_sampled_masks):
    """utils/feature_utils.py:17-26 (draws on the CPU generator, like the reference)."""
    sampled_mask = torch.rand(sam_masks.shape[0]) < num_sampled_masks / sam_masks.shape[0]
    rate = num_sampled_pixels / (sam_masks.shape[-1] * sam_masks.shape[-2])
    sampled_pixel = torch.rand(sam_masks.shape[-2], sam_masks.shape[-1]) < rate
    non_mask_region = sam_masks.sum(dim=0) == 0
    return torch.logical_and(sampled_pixel, ~non_mask_region.cpu()), sampled_mask


def pixel_weights(sam_masks, sampled_pixel):
    """utils/feature_utils.py:28-38 (float32 like the reference: the int64 sums are divided as float32)."""
    per_pixel_mask_size = sam_masks * sam_masks.sum(-1).sum(-1)[:, None, None]
    mean_size = per_pixel_mask_size.sum(dim=0) / (sam_masks.sum(dim=0) + 1e-9)
    mean_size = mean_size[sampled_pixel]
    pp = mean_size.unsqueeze(0) * mean_size.unsqueeze(1)
    mx = pp.max()
    pp[pp == 0] = 1e10
    w = torch.clamp(mx / pp, 1.0, None)
    return (w - w.min()) / (w.max() - w.min()) * 9. + 1.


def correspondence_matrix(sam_masks, sampled_pixel, sampled_mask):
    """utils/feature_utils.py:40-49."""
    v = sam_masks[:, sampled_pixel][sampled_mask, :].float()
    c = torch.einsum("nh,nj->hj", v, v)
    c[c != 0] = 1
    return c


def feature_matrix(rendered_features, sampled_pixel, dtype=torch.float32):
    """utils/feature_utils.py:51-57."""
    f = rendered_features[:, sampled_pixel].permute([1, 0]).to(dtype)
    f = torch.nn.functional.normalize(f, dim=-1, p=2)
    return torch.einsum("hc,jc->hj", f, f)


def pair_losses(C, C_F, pth, nth, weights, mode):
    """utils/loss_utils.py:275-406: (positive_pixel_pair_loss[mode], negative_pixel_pair_loss[mode])."""
    n = C_F.shape[0]
    diag = torch.eye(n, dtype=torch.bool, device=C_F.device)
    w = torch.ones_like(C_F) if weights is None else weights.to(C_F.dtype)
    out = []
    for neg in (False, True):
        cv = 0 if neg else 1
        if mode == "hard":
            m = torch.triu(((C_F > nth) if neg else (C_F < pth)) & (C == cv) & (~diag), diagonal=0)
            if int(m.sum()) == 0:
                out.append(C_F.sum() * 0.0)
                continue
            out.append((w[m] * torch.relu(C_F[m])).mean() if neg else (-w[m] * C_F[m]).mean())
            continue
        if mode == "all":
            cond = C == cv
        else:
            cond = torch.logical_and(C_F > nth, C == 0) if neg else torch.logical_and(C_F < pth, C == 1)
        m = torch.triu(torch.logical_and(torch.any(cond, dim=0), ~diag), diagonal=0)
        npair = int(torch.nonzero(m).shape[0])
        m = torch.logical_and(m, C == cv)
        if int(m.sum()) == 0:
            out.append(C_F.sum() * 0.0)
        elif neg:
            out.append((w[m] * torch.relu(C_F[m])).sum() / npair)
        else:
            out.append((-w[m] * C_F[m]).sum() / npair)
    return out


def head(rendered_features, sam_masks, sampled_pixel, sampled_mask, mode="soft", pth=0.75, nth=0.5, use_weights=True,
         dtype=torch.float32):
    """(loss_pos, loss_neg, pos_similarity, neg_similarity) as train.py:272-296 composes them."""
    C = correspondence_matrix(sam_masks, sampled_pixel, sampled_mask)
    C_F = feature_matrix(rendered_features, sampled_pixel, dtype)
    W = pixel_weights(sam_masks, sampled_pixel) if use_weights else None
    lp, ln = pair_losses(C, C_F, pth, nth, W, mode)
    with torch.no_grad():
        ps, ns = C_F[C == 1].mean(), C_F[C == 0].mean()
    return lp, ln, ps, ns


def feature_norm_reg(rendered_features):
    """train.py:281-282."""
    return (1 - rendered_features.norm(dim=0, p=2).mean()) ** 2
