"""TEST INFRASTRUCTURE ONLY -- numpy restatement of GaussianModel.densify_and_prune (scene/gaussian_model.py:617-635)
and of the per-iteration densification statistics (train.py:362-365, scene/gaussian_model.py:637-639).

Pinned: tests/test_densify_oracle.py checks it against tests/golden/densify.npz, produced by running the reference's
own GaussianModel class on the CPU (tests/golden/make_golden.py, G7).  It follows the reference's three rounds
literally (clone-append -> split-append + prune parents -> final prune) on plain arrays, so that the single row map of
trase_amd/csrc/densify.hip is checked against an independent formulation.
"""
import numpy as np

F = np.float32


def add_densification_stats(accum, denom, max_radii2D, viewspace_grad, radii):
    """train.py:362-365; scene/gaussian_model.py:637-639.  Arrays are modified in place."""
    vis = radii > 0
    max_radii2D[vis] = np.maximum(max_radii2D[vis], radii[vis].astype(F))
    g = viewspace_grad[vis, :2].astype(F)
    accum[vis, 0] += np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1], dtype=F)
    denom[vis, 0] += F(1.0)


def _sigmoid(x):
    return (F(1.0) / (F(1.0) + np.exp(-x, dtype=F))).astype(F)


def _build_rotation(r):
    """utils/general_utils.py:122-143."""
    norm = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3], dtype=F)
    q = (r / norm[:, None]).astype(F)
    R = np.zeros((q.shape[0], 3, 3), dtype=F)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_prune(params, moments, accum, denom, percent_dense, extent, max_grad, min_opacity, max_screen_size, z):
    """params: {name: array [P, ...]} with the reference's group names (xyz, f_dc, f_rest, opacity, scaling, rotation,
    gaussian_feats, ...); moments: {name: (exp_avg, exp_avg_sq)} for the groups that have Adam state; z: [2M, 3]
    standard normals consumed by the split (ignored rows never read).  Returns (params, moments, num_clone, num_split)."""
    params = {k: v.astype(F, copy=True) for k, v in params.items()}
    moments = {k: (m.astype(F, copy=True), v.astype(F, copy=True)) for k, (m, v) in moments.items()}

    def cat(new):                                   # cat_tensors_to_optimizer :511-534
        for k in params:
            params[k] = np.concatenate([params[k], new[k]], axis=0)
            if k in moments:
                m, v = moments[k]
                moments[k] = (np.concatenate([m, np.zeros_like(new[k])], 0), np.concatenate([v, np.zeros_like(new[k])], 0))

    def prune(mask):                                # prune_points :491-509
        keep = ~mask
        for k in params:
            params[k] = params[k][keep]
            if k in moments:
                m, v = moments[k]
                moments[k] = (m[keep], v[keep])

    with np.errstate(invalid="ignore", divide="ignore"):
        grads = (accum.astype(F) / denom.astype(F)).reshape(-1)
    grads[np.isnan(grads)] = 0.0                    # :618-619
    dense = F(percent_dense * extent)
    # densify_and_clone :594-615
    smax = np.exp(params["scaling"], dtype=F).max(axis=1)
    sel = (np.abs(grads) >= F(max_grad)) & (smax <= dense)
    num_clone = int(sel.sum())
    cat({k: v[sel] for k, v in params.items()})
    # densify_and_split :563-592 (statistics were re-created as zeros, the padded gradient covers the original rows)
    n_now = params["xyz"].shape[0]
    padded = np.zeros(n_now, dtype=F)
    padded[:grads.shape[0]] = grads
    smax = np.exp(params["scaling"], dtype=F).max(axis=1)
    sel = (padded >= F(max_grad)) & (smax > dense)
    M = int(sel.sum())
    stds = np.tile(np.exp(params["scaling"][sel], dtype=F), (2, 1))
    samples = (z[:2 * M].astype(F) * stds).astype(F)
    rots = np.tile(_build_rotation(params["rotation"][sel]), (2, 1, 1))
    new = {k: np.tile(v[sel], (2,) + (1,) * (v.ndim - 1)) for k, v in params.items()}
    new["xyz"] = (np.einsum("nij,nj->ni", rots, samples).astype(F) + np.tile(params["xyz"][sel], (2, 1))).astype(F)
    new["scaling"] = np.log(np.tile(np.exp(params["scaling"][sel], dtype=F), (2, 1)) / F(0.8 * 2), dtype=F)
    cat(new)
    prune(np.concatenate([sel, np.zeros(2 * M, dtype=bool)]))
    # final prune :624-628; max_radii2D was zeroed by densification_postfix (:553-555), so its term never fires
    mask = _sigmoid(params["opacity"]).reshape(-1) < F(min_opacity)
    if max_screen_size:
        mask = mask | (np.exp(params["scaling"], dtype=F).max(axis=1) > F(0.1 * extent))
    prune(mask)
    return params, moments, num_clone, M
