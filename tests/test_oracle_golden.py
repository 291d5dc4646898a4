"""Pins the oracle's restated sub-steps (and the product's host-side camera helpers) against the
golden vectors generated from the imported reference (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import torch

from oracle import raster_oracle as ro
from trase_amd import synthetic

G = os.path.join(os.path.dirname(__file__), "golden")


def test_sh_colour_matches_reference_eval_sh():
    d = np.load(os.path.join(G, "sh_eval.npz"))
    shs = torch.from_numpy(d["shs"]).double()
    xyz = torch.from_numpy(d["xyz"]).double()
    campos = torch.from_numpy(d["campos"]).double()
    dirs = xyz - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    for deg in range(4):
        raw = ro.eval_sh_colors(deg, shs, dirs)
        np.testing.assert_allclose(raw.numpy(), d[f"raw_deg{deg}"], rtol=1e-5, atol=2e-6)
        rgb = torch.clamp_min(raw + 0.5, 0.0)
        np.testing.assert_allclose(rgb.numpy(), d[f"rgb_deg{deg}"], rtol=1e-5, atol=2e-6)


def test_cov3d_matches_reference_build_scaling_rotation():
    d = np.load(os.path.join(G, "cov3d.npz"))
    cov = ro.cov3d_from_scale_rot(torch.from_numpy(d["scales"]).double(), torch.from_numpy(d["rotations"]).double(),
                                  float(d["modifier"]))
    np.testing.assert_allclose(cov.numpy(), d["cov6"], rtol=2e-5, atol=1e-7)


def test_camera_helpers_match_reference():
    d = np.load(os.path.join(G, "camera.npz"))
    R, T = torch.from_numpy(d["R"]), torch.from_numpy(d["T"])
    wvt = synthetic.world2view(R, T).transpose(0, 1)
    proj = synthetic.projection_matrix(0.01, 100.0, float(d["fovx"]), float(d["fovy"])).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    np.testing.assert_allclose(wvt.numpy(), d["world_view_transform"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(proj.numpy(), d["projection_matrix"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(full.numpy(), d["full_proj_transform"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(wvt.inverse()[3, :3].numpy(), d["camera_center"], rtol=1e-5, atol=1e-6)


def test_oracle_projection_uses_reference_conventions():
    """A point straight ahead of the golden camera lands in the image centre with depth == view z."""
    d = np.load(os.path.join(G, "camera.npz"))
    wvt = torch.from_numpy(d["world_view_transform"])
    full = torch.from_numpy(d["full_proj_transform"])
    center = torch.from_numpy(d["camera_center"])
    w2c = wvt.T.double()
    p_cam = torch.tensor([0.0, 0.0, 3.0, 1.0], dtype=torch.float64)
    p_world = (torch.linalg.inv(w2c) @ p_cam)[:3]

    class S:
        image_width, image_height = 64, 48
        tanfovx, tanfovy = math.tan(float(d["fovx"]) / 2), math.tan(float(d["fovy"]) / 2)
        viewmatrix, projmatrix, campos = wvt, full, center
        scale_modifier, sh_degree = 1.0, 0
    g = ro.preprocess(S, p_world[None], None, torch.ones(1, 3, dtype=torch.float64), torch.ones(1, 1, dtype=torch.float64),
                      torch.full((1, 3), 0.05, dtype=torch.float64), torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64), None)
    assert bool(g.valid[0])
    np.testing.assert_allclose(g.depth.numpy(), [3.0], rtol=1e-6)
    np.testing.assert_allclose(g.xy.numpy(), [[31.5, 23.5]], atol=1e-4)
