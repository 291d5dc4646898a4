"""Edge cases of the spatial-hash KNN (distCUDA2, knn_points) against scipy's KD-tree: tiny clouds, duplicates, degenerate
dimensionality, extreme scales, clusters far apart.  Prints each case first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial import cKDTree
from simple_knn._C import distCUDA2
from pytorch3d.ops import knn_points
rng = np.random.default_rng(3)
bad = 0
def clouds():
    for n in (4, 5, 7, 17, 64, 65, 1000, 4097):
        yield f"uniform n={n}", rng.uniform(-1, 1, size=(n, 3))
    yield "line", np.stack([np.linspace(0, 1, 3000), np.zeros(3000), np.zeros(3000)], 1)
    yield "plane", np.concatenate([rng.uniform(size=(5000, 2)), np.zeros((5000, 1))], 1)
    yield "duplicates (every point four times)", np.repeat(rng.uniform(size=(800, 3)), 4, axis=0)
    yield "all identical", np.ones((300, 3)) * 0.37
    yield "huge scale", rng.uniform(-1, 1, size=(3000, 3)) * 1e6
    yield "tiny scale", rng.uniform(-1, 1, size=(3000, 3)) * 1e-6
    yield "offset 1e4 + unit cloud", rng.uniform(-1, 1, size=(3000, 3)) + 1e4
    yield "two clusters 1e5 apart", np.concatenate([rng.normal(size=(2000, 3)) * 0.01, rng.normal(size=(2000, 3)) * 0.01 + 1e5])
    yield "one outlier", np.concatenate([rng.uniform(size=(5000, 3)), [[1e3, 1e3, 1e3]]])
for name, pts in clouds():
    pts = pts.astype(np.float32)
    n = len(pts)
    print(name, flush=True)
    t = torch.from_numpy(pts).cuda()
    tree = cKDTree(pts.astype(np.float64))
    if n >= 4:
        d, _ = tree.query(pts.astype(np.float64), k=4)
        want = (d[:, 1:] ** 2).mean(axis=1)
        got = distCUDA2(t).cpu().numpy()
        scale = max(float(want.max()), 1e-30)
        if not np.allclose(got, want, rtol=5e-4, atol=1e-6 * scale + 1e-30):
            bad += 1; print("   distCUDA2 MISMATCH max abs", float(np.abs(got - want).max()), "scale", scale, flush=True)
    for K in (1, 3, 16):
        if n < K:
            continue
        out = knn_points(t.unsqueeze(0), t.unsqueeze(0), K=K)
        d, i = tree.query(pts.astype(np.float64), k=K)
        d = d.reshape(n, K)
        got = out.dists[0].cpu().numpy()
        scale = max(float((d ** 2).max()), 1e-30)
        if not np.allclose(got, d ** 2, rtol=5e-4, atol=1e-6 * scale + 1e-30):
            bad += 1; print(f"   knn_points K={K} MISMATCH max abs", float(np.abs(got - d ** 2).max()), "scale", scale, flush=True)
        idx = out.idx[0].cpu().numpy()
        if idx.min() < 0 or idx.max() >= n:
            bad += 1; print(f"   knn_points K={K} index out of range", idx.min(), idx.max(), flush=True)
print("done; problems:", bad)
