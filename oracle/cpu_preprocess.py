"""TEST / BENCH INFRASTRUCTURE ONLY -- fp32 PyTorch-CPU restatement of the reference's *Python* preprocess, the CPU
baseline leg BASELINE.json's north_star names ("the reference's PyTorch-CPU preprocess timed on the box's host
cores").  Never imported by the product.

What it restates, in the reference's own op order (so that timing it is timing the same work):
  * activations + deformation add + feature L2-normalisation -- gaussian_renderer/__init__.py:82-121,
    scene/gaussian_model.py:43-51,183-205
  * SH -> RGB on the Python route (pipe.convert_SHs_python) -- gaussian_renderer/__init__.py:103-108,
    utils/sh_utils.py:57-112
  * Sigma = (R S)(R S)^T stripped to 6 floats (pipe.compute_cov3D_python) -- scene/gaussian_model.py:37-41,216-217,
    utils/general_utils.py:108-154 (build_rotation normalises the quaternion)
  * projection of the centres by full_proj_transform -- render.py:247-251

Pinned by tests/test_cpu_preprocess.py against tests/golden/render_prep.npz (arguments the imported reference's
render() handed to the rasterizer for the shs_python / cov_python / plain call patterns) and sh_eval.npz / cov3d.npz.
"""
from __future__ import annotations

import torch

from .raster_oracle import SH_C0, SH_C1, SH_C2, SH_C3


def eval_sh_channel_major(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """utils/sh_utils.py:57-112 for sh (N,3,16) (= get_features.transpose(1,2)), dirs (N,3) -> (N,3)."""
    res = SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - SH_C1 * y * sh[..., 1] + SH_C1 * z * sh[..., 2] - SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[..., 4] + SH_C2[1] * yz * sh[..., 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + SH_C2[3] * xz * sh[..., 7] + SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + SH_C3[1] * xy * z * sh[..., 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + SH_C3[5] * z * (xx - yy) * sh[..., 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """utils/general_utils.py:121-144 (normalises)."""
    q = r / torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])[:, None]
    R = torch.zeros((q.size(0), 3, 3), dtype=r.dtype)
    r_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r_ * z)
    R[:, 0, 2] = 2 * (x * z + r_ * y)
    R[:, 1, 0] = 2 * (x * y + r_ * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r_ * x)
    R[:, 2, 0] = 2 * (x * z - r_ * y)
    R[:, 2, 1] = 2 * (y * z + r_ * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def covariance6(scaling_act: torch.Tensor, scaling_modifier: float, rotation_raw: torch.Tensor) -> torch.Tensor:
    """scene/gaussian_model.py:37-41 + utils/general_utils.py:108-119,146-156."""
    s = scaling_modifier * scaling_act
    L = torch.zeros((s.shape[0], 3, 3), dtype=s.dtype)
    L[:, 0, 0], L[:, 1, 1], L[:, 2, 2] = s[:, 0], s[:, 1], s[:, 2]
    L = build_rotation(rotation_raw) @ L
    S = L @ L.transpose(1, 2)
    out = torch.zeros((s.shape[0], 6), dtype=s.dtype)
    out[:, 0], out[:, 1], out[:, 2] = S[:, 0, 0], S[:, 0, 1], S[:, 0, 2]
    out[:, 3], out[:, 4], out[:, 5] = S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]
    return out


def reference_cpu_preprocess(xyz, features_dc, features_rest, opacity, scaling, rotation, gaussian_features,
                             d_xyz, d_rotation, d_scaling, full_proj_transform, camera_center, image_width, image_height,
                             sh_degree=3, scaling_modifier=1.0, norm_features=True):
    """One view's per-Gaussian preprocess on the reference's Python routes.  Raw (pre-activation) parameters in,
    dict of the tensors the rasterizer would be handed + the projected centres out."""
    means3D = xyz + d_xyz                                                     # gaussian_renderer/__init__.py:82
    opac = torch.sigmoid(opacity)                                             # :85
    scales = torch.exp(scaling) + d_scaling                                   # :95
    rotations = torch.nn.functional.normalize(rotation) + d_rotation          # :96
    cov6 = covariance6(torch.exp(scaling), scaling_modifier, rotation)        # :94 -> scene/gaussian_model.py:216-217
    feats = torch.cat((features_dc, features_rest), dim=1)                    # scene/gaussian_model.py:194-197
    shs_view = feats.transpose(1, 2).view(-1, 3, 16)                          # gaussian_renderer/__init__.py:104
    dir_pp = xyz - camera_center.repeat(feats.shape[0], 1)                    # :105
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)             # :106
    sh2rgb = eval_sh_channel_major(sh_degree, shs_view, dir_pp_normalized)    # :107
    colors = torch.clamp_min(sh2rgb + 0.5, 0.0)                               # :108
    sh_objs = gaussian_features
    if norm_features:
        sh_objs = sh_objs / (sh_objs.norm(dim=2, keepdim=True) + 1e-9)        # :120-121
    cur_pts = torch.cat([means3D, torch.ones_like(means3D[..., :1])], dim=-1)  # render.py:247
    p2d = cur_pts @ full_proj_transform                                       # render.py:249
    p2d = p2d[..., :2] / p2d[..., -1:]
    p2d = (p2d + 1) / 2 * torch.tensor([image_width, image_height], dtype=p2d.dtype)
    return dict(means3D=means3D, opacities=opac, scales=scales, rotations=rotations, cov3D_precomp=cov6,
                colors_precomp=colors, sh_objs=sh_objs, pts2d=p2d)
