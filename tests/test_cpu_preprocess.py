"""The CPU-baseline restatement (oracle/cpu_preprocess.py: the reference's PyTorch-CPU preprocess) against what the
imported reference's own render() handed to the rasterizer (tests/golden/render_prep.npz) -- pins the leg that
bench.py times as `cpu_baseline`."""
import os

import numpy as np
import torch

from oracle.cpu_preprocess import reference_cpu_preprocess

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_prep.npz"))
T = lambda k: torch.from_numpy(np.asarray(G[k]))


def _run(**kw):
    return reference_cpu_preprocess(T("pc_xyz"), T("pc_features_dc"), T("pc_features_rest"), T("pc_opacity"), T("pc_scaling"),
                                    T("pc_rotation"), T("pc_gaussian_features"), T("d_xyz"), T("d_rotation"), T("d_scaling"),
                                    T("full_proj_transform"), T("camera_center"), int(G["W"]), int(G["H"]), **kw)


def test_cpu_preprocess_matches_reference_render_arguments():
    out = _run()
    for key, case in (("means3D", "plain"), ("opacities", "plain"), ("scales", "plain"), ("rotations", "plain"),
                      ("sh_objs", "plain"), ("colors_precomp", "shs_python"), ("cov3D_precomp", "cov_python")):
        want = T(f"{case}__{key}")
        assert out[key].shape == want.shape, key
        np.testing.assert_allclose(out[key].numpy(), want.numpy(), rtol=2e-6, atol=2e-7, err_msg=key)
    assert torch.isfinite(out["pts2d"]).all()


def test_cpu_preprocess_modifier_and_raw_features():
    out = _run(norm_features=False)
    np.testing.assert_allclose(out["sh_objs"].numpy(), G["nonorm__sh_objs"], rtol=0, atol=0)
