"""Real spherical-harmonics colour evaluation in PyTorch ops -- the ``pipe.convert_SHs_python`` branch of the
reference's render() (gaussian_renderer/__init__.py:103-108 -> utils/sh_utils.py:57-112), degrees 0..3.  The default
path evaluates the same polynomial inside the per-Gaussian HIP kernel (trase_amd/csrc/gs_math.h); this is the
opt-in Python route the reference offers, kept so that a pipeline configured with --convert_SHs_python runs."""
from __future__ import annotations

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh (..., C, (deg_max+1)^2) channel-major as the reference passes it (``get_features.transpose(1,2)``),
    dirs (..., 3) unit vectors -> (..., C).  Same term order as utils/sh_utils.py:74-100."""
    if not 0 <= deg <= 3:
        raise ValueError(f"eval_sh: degree {deg} outside 0..3")
    if sh.shape[-1] < (deg + 1) ** 2:
        raise ValueError("eval_sh: not enough coefficients for the requested degree")
    res = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                       + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                       + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def sh_colors_python(pc, camera_center: torch.Tensor) -> torch.Tensor:
    """colors_precomp of gaussian_renderer/__init__.py:104-108 (note: the direction uses pc.get_xyz, not the deformed
    means -- as the reference does)."""
    feats = pc.get_features
    shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
    dir_pp = pc.get_xyz - camera_center.reshape(1, 3)
    dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
