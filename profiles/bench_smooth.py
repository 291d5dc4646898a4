#!/usr/bin/env python
"""Secondary measurement: KNN feature smoothing (row A7, scene/gaussian_model.py:79-104) at N Gaussians, K = 16,
8 selected slots -- fused HIP gather + reverse-adjacency backward vs the reference's PyTorch composition
(normalize -> index -> mean, index_put backward), forward + backward."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d.ops import knn_points
from trase_amd.smooth import smooth_features, reverse_adjacency


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    xyz = (torch.rand(n, 3, device=dev) * 2 - 1) * 1.3
    feats = torch.randn(n, 1, 32, device=dev, requires_grad=True)
    t0 = time.perf_counter()
    idx = knn_points(xyz.unsqueeze(0), xyz.unsqueeze(0), K=16).idx.squeeze()
    torch.cuda.synchronize()
    t_knn = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    rev = reverse_adjacency(idx)
    torch.cuda.synchronize()
    t_rev = (time.perf_counter() - t0) * 1e3
    sel = torch.randperm(16)[:8]
    seld = sel.to(dev)
    w = torch.randn(n, 1, 32, device=dev)

    def ref():
        feats.grad = None
        normed = torch.nn.functional.normalize(feats, dim=-1, p=2)
        (normed[idx[:, seld], 0, :].mean(dim=1).unsqueeze(1) * w).sum().backward()

    def hip():
        feats.grad = None
        (smooth_features(feats, idx, sel, rev) * w).sum().backward()

    ref(); g_ref = feats.grad.clone()
    hip(); g_hip = feats.grad.clone()
    print(json.dumps({"n": n, "K": 16, "S": 8, "hip_fwd_bwd_ms": round(timed(hip), 4), "torch_fwd_bwd_ms": round(timed(ref), 4),
                      "knn_once_ms": round(t_knn, 2), "reverse_adjacency_once_ms": round(t_rev, 2),
                      "max_rel_grad_diff": float((g_hip - g_ref).abs().max() / g_ref.abs().max())}))


if __name__ == "__main__":
    main()
