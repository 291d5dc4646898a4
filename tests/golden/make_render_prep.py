"""Generates render_prep.npz: what the REFERENCE's own render() (gaussian_renderer/__init__.py:37-155) feeds the
rasterizer operator.  The reference module is imported in the build container with a *recording* stand-in for
``diff_gaussian_rasterization`` (the CUDA extension is absent) and run on a seeded CPU ``GaussianModel`` of the
reference's own class (scene/gaussian_model.py), once per call pattern:

  plain            tensor d_xyz / d_rotation / d_scaling                    (train.py:204,210 after warm-up)
  float0           d_* = 0.0 Python floats                                  (train.py:192-193 warm-up, render.py:306)
  sixdof           is_6dof with (N,4,4) transforms                          (gaussian_renderer/__init__.py:75-80)
  sixdof_float     is_6dof with a non-tensor d_xyz                          (:76-77)
  mask             boolean subset                                           (:123-135)
  override         override_color                                           (:112-113)
  nonorm           norm_gaussian_features=False                             (:120-121 skipped)
  smooth           is_smooth_gaussian_features=True, K=16, dropout 0.5      (:116-118, scene/gaussian_model.py:79-104)
  shs_python       pipe.convert_SHs_python                                  (:103-108)
  cov_python       pipe.compute_cov3D_python                                (:93-94, scene/gaussian_model.py:216-217)
  modifier         scaling_modifier = 0.7

Only data is written (inputs and the recorded operator arguments).  Runs only where /root/reference exists:

    python tests/golden/make_render_prep.py
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, no_cuda_kwarg  # noqa: E402

KW = ("means3D", "means2D", "shs", "sh_objs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")


class Recorder:
    last = None

    class Settings:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class Rasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, **kw):
            assert set(kw) == set(KW), sorted(kw)
            Recorder.last = (self.rs, {k: (None if v is None else v.detach().clone()) for k, v in kw.items()})
            n = kw["means3D"].shape[0]
            H, W = self.rs.image_height, self.rs.image_width
            return torch.zeros(3, H, W), torch.ones(n, dtype=torch.int32), torch.zeros(32, H, W), torch.zeros(1, H, W)


def knn_points_cpu(p1, p2, K=1, **_):
    d = torch.cdist(p1[0].double(), p2[0].double())
    dist, idx = d.topk(K, largest=False)
    return types.SimpleNamespace(dists=(dist ** 2).float().unsqueeze(0), idx=idx.unsqueeze(0), knn=None)


def main():
    import_reference()
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizationSettings = Recorder.Settings
    sys.modules["diff_gaussian_rasterization"].GaussianRasterizer = Recorder.Rasterizer
    sys.modules["pytorch3d.ops"].knn_points = knn_points_cpu
    from gaussian_renderer import render
    from scene.gaussian_model import GaussianModel
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix

    torch.manual_seed(11)
    n, F = 96, 32
    gm = GaussianModel(3)
    gm.active_sh_degree = 3
    P = lambda t: torch.nn.Parameter(t.float().contiguous())
    gm._xyz = P((torch.rand(n, 3) * 2 - 1) * 1.2)
    gm._features_dc = P(torch.randn(n, 1, 3) * 0.8)
    gm._features_rest = P(torch.randn(n, 15, 3) * 0.1)
    gm._scaling = P(torch.randn(n, 3) * 0.4 - 2.5)
    gm._rotation = P(torch.randn(n, 4))
    gm._opacity = P(torch.randn(n, 1) * 2)
    gm._gaussian_features = P(torch.randn(n, 1, F))

    ang = 0.35
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]], dtype=np.float64)
    T = np.array([0.05, -0.1, 4.0])
    W, H, fovx, fovy = 112, 80, 0.8, 0.6
    wvt = torch.tensor(getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    cam = types.SimpleNamespace(FoVx=fovx, FoVy=fovy, image_height=H, image_width=W, world_view_transform=wvt,
                                full_proj_transform=full, camera_center=wvt.inverse()[3, :3])
    bg = torch.tensor([0.1, 0.2, 0.3])
    d_xyz, d_rot, d_scale = torch.randn(n, 3) * 0.05, torch.randn(n, 4) * 0.05, torch.randn(n, 3) * 0.01
    # rigid transforms for is_6dof: small rotation about a random axis + translation
    ax = torch.nn.functional.normalize(torch.randn(n, 3), dim=1)
    th = torch.randn(n) * 0.1
    Kx = torch.zeros(n, 3, 3)
    Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0], Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    Rm = torch.eye(3)[None] + torch.sin(th)[:, None, None] * Kx + (1 - torch.cos(th))[:, None, None] * (Kx @ Kx)
    T44 = torch.eye(4).repeat(n, 1, 1)
    T44[:, :3, :3] = Rm
    T44[:, :3, 3] = torch.randn(n, 3) * 0.05
    mask = torch.rand(n) < 0.6
    override = torch.rand(n, 3)

    pipe = lambda **k: types.SimpleNamespace(**{"debug": False, "compute_cov3D_python": False, "convert_SHs_python": False, **k})
    cases = {
        "plain": dict(args=(d_xyz, d_rot, d_scale)),
        "float0": dict(args=(0.0, 0.0, 0.0)),
        "sixdof": dict(args=(T44, d_rot, d_scale), kw=dict(is_6dof=True)),
        "sixdof_float": dict(args=(0.0, d_rot, d_scale), kw=dict(is_6dof=True)),
        "mask": dict(args=(d_xyz, d_rot, d_scale), kw=dict(mask=mask)),
        "override": dict(args=(d_xyz, d_rot, d_scale), kw=dict(override_color=override)),
        "nonorm": dict(args=(d_xyz, d_rot, d_scale), kw=dict(norm_gaussian_features=False)),
        "smooth": dict(args=(d_xyz, d_rot, d_scale), kw=dict(is_smooth_gaussian_features=True, smooth_K=16), seed=123),
        "shs_python": dict(args=(d_xyz, d_rot, d_scale), pipe=pipe(convert_SHs_python=True)),
        "cov_python": dict(args=(d_xyz, d_rot, d_scale), pipe=pipe(compute_cov3D_python=True)),
        "modifier": dict(args=(d_xyz, d_rot, d_scale), kw=dict(scaling_modifier=0.7)),
    }
    out = {"names": np.array(sorted(cases))}
    for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_gaussian_features"):
        out["pc" + k] = getattr(gm, k).detach().numpy()
    out.update(world_view_transform=wvt.numpy(), full_proj_transform=full.numpy(), camera_center=cam.camera_center.numpy(),
               FoVx=fovx, FoVy=fovy, W=W, H=H, bg=bg.numpy(), d_xyz=d_xyz.numpy(), d_rotation=d_rot.numpy(),
               d_scaling=d_scale.numpy(), T44=T44.numpy(), mask=mask.numpy(), override_color=override.numpy())
    for name, c in cases.items():
        gm.feature_smooth_map = None
        if "seed" in c:
            torch.manual_seed(c["seed"])          # torch.randperm(K) inside get_smoothed_gaussian_features
            out[f"{name}__seed"] = c["seed"]
        with no_cuda_kwarg():
            res = render(cam, gm, c.get("pipe", pipe()), bg, *c["args"], **c.get("kw", {}))
        rs, kw = Recorder.last
        assert set(res) == {"render", "viewspace_points", "visibility_filter", "radii", "render_gaussian_features", "depth"}
        for k, v in kw.items():
            out[f"{name}__{k}"] = np.zeros(0, np.float32) if v is None else v.numpy()
            out[f"{name}__{k}__none"] = v is None
        for k in ("image_height", "image_width", "tanfovx", "tanfovy", "scale_modifier", "sh_degree", "prefiltered", "debug"):
            out[f"{name}__rs_{k}"] = getattr(rs, k)
        for k in ("bg", "viewmatrix", "projmatrix", "campos"):
            out[f"{name}__rs_{k}"] = getattr(rs, k).numpy()
        if name == "smooth":
            out["smooth__knn_idx"] = gm.feature_smooth_map["m"].numpy()
        print(name, {k: (None if v is None else tuple(v.shape)) for k, v in kw.items()})
    np.savez_compressed(os.path.join(HERE, "render_prep.npz"), **out)
    print("wrote render_prep.npz", os.path.getsize(os.path.join(HERE, "render_prep.npz")), "bytes")


if __name__ == "__main__":
    main()
