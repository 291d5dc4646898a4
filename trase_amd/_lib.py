"""ctypes binding of the C-ABI library (include/trase_rast.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  If
the shared object is missing or fails to load, importing a rasterizer entry
point raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first: the library binds to torch's libamdhip64

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRASE_RAST_LIB") or os.path.join(_HERE, "lib", "libtrase_rast.so")   # override: A/B builds

c_float_p = C.POINTER(C.c_float)


class RastSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_void_p), ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("sh_degree", C.c_int32), ("campos", C.c_void_p),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("device", C.c_int32), ("variant", C.c_int32),
        ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32),
        ("feat_bg", C.c_float), ("reserved0", C.c_int32),
    ]


class RastInputs(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("M", C.c_int32), ("F", C.c_int32),
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("sh_objs", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
        ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
    ]


class RastOutputs(C.Structure):
    _fields_ = [("image", C.c_void_p), ("radii", C.c_void_p), ("feats", C.c_void_p), ("depth", C.c_void_p)]


class RastWorkspace(C.Structure):
    _fields_ = [
        ("geom", C.c_void_p), ("geom_bytes", C.c_size_t),
        ("bin", C.c_void_p), ("bin_bytes", C.c_size_t),
        ("img", C.c_void_p), ("img_bytes", C.c_size_t),
        ("pre", C.c_void_p), ("pre_bytes", C.c_size_t),
        ("tmp", C.c_void_p), ("tmp_bytes", C.c_size_t),
        ("capacity", C.c_int64),
    ]


class RastSizes(C.Structure):
    _fields_ = [("geom_bytes", C.c_size_t), ("bin_bytes", C.c_size_t), ("img_bytes", C.c_size_t),
                ("pre_bytes", C.c_size_t), ("tmp_bytes", C.c_size_t), ("bwd_tmp_bytes", C.c_size_t)]


class RastGrads(C.Structure):
    _fields_ = [
        ("dL_dimage", C.c_void_p), ("dL_dfeats", C.c_void_p), ("dL_ddepth", C.c_void_p),
        ("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
        ("dL_dsh_objs", C.c_void_p), ("dL_dcolors", C.c_void_p), ("dL_dopacities", C.c_void_p),
        ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p),
    ]


class RastRawInputs(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("F", C.c_int32), ("norm_features", C.c_int32),
        ("xyz", C.c_void_p), ("d_xyz", C.c_void_p), ("features_dc", C.c_void_p), ("features_rest", C.c_void_p),
        ("opacity", C.c_void_p), ("scaling", C.c_void_p), ("d_scaling", C.c_void_p), ("rotation", C.c_void_p),
        ("d_rotation", C.c_void_p), ("gaussian_features", C.c_void_p), ("featn", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("mask", C.c_void_p), ("d_xyz_se3", C.c_void_p),
        ("sh_dir_undeformed", C.c_int32), ("reserved1", C.c_int32),
    ]


class RastRawGrads(C.Structure):
    _fields_ = [
        ("dL_dimage", C.c_void_p), ("dL_dfeats", C.c_void_p), ("dL_ddepth", C.c_void_p),
        ("dL_dxyz", C.c_void_p), ("dL_dd_xyz", C.c_void_p), ("dL_dmeans2D", C.c_void_p),
        ("dL_dfeatures_dc", C.c_void_p), ("dL_dfeatures_rest", C.c_void_p), ("dL_dopacity", C.c_void_p),
        ("dL_dscaling", C.c_void_p), ("dL_dd_scaling", C.c_void_p), ("dL_drotation", C.c_void_p),
        ("dL_dd_rotation", C.c_void_p), ("dL_dgaussian_features", C.c_void_p),
        ("dL_dcolors_precomp", C.c_void_p), ("dL_dd_xyz_se3", C.c_void_p),
    ]


class MlpWeights(C.Structure):
    _fields_ = [
        ("D", C.c_int32), ("W", C.c_int32), ("xyz_multires", C.c_int32), ("t_multires", C.c_int32),
        ("is_blender", C.c_int32), ("is_6dof", C.c_int32), ("variant", C.c_int32), ("reserved", C.c_int32),
        ("weight", C.c_void_p * 8), ("bias", C.c_void_p * 8),
        ("w_warp", C.c_void_p), ("b_warp", C.c_void_p), ("w_rotation", C.c_void_p), ("b_rotation", C.c_void_p),
        ("w_scaling", C.c_void_p), ("b_scaling", C.c_void_p),
    ]


class MlpGrads(C.Structure):
    _fields_ = [
        ("weight", C.c_void_p * 8), ("bias", C.c_void_p * 8),
        ("w_warp", C.c_void_p), ("b_warp", C.c_void_p), ("w_rotation", C.c_void_p), ("b_rotation", C.c_void_p),
        ("w_scaling", C.c_void_p), ("b_scaling", C.c_void_p),
    ]


# every symbol include/trase_rast.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("trase_rast_sizes", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(RastSizes)]),
    ("trase_rast_preprocess", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastInputs), C.POINTER(RastOutputs),
                                        C.POINTER(RastWorkspace), C.c_void_p]),
    ("trase_rast_status", C.c_int, [C.POINTER(RastWorkspace), C.POINTER(C.c_int64 * 3), C.c_void_p]),
    ("trase_rast_geom_layout", C.c_int, [C.c_int32, C.POINTER(C.c_int64 * 6)]),
    ("trase_rast_render", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastInputs), C.POINTER(RastOutputs),
                                    C.POINTER(RastWorkspace), C.c_void_p]),
    ("trase_rast_forward", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastInputs), C.POINTER(RastOutputs),
                                     C.POINTER(RastWorkspace), C.c_void_p]),
    ("trase_rast_backward", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastInputs), C.POINTER(RastOutputs),
                                      C.POINTER(RastWorkspace), C.POINTER(RastGrads), C.c_void_p]),
    ("trase_rast_preprocess_raw", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastOutputs),
                                            C.POINTER(RastWorkspace), C.c_void_p]),
    ("trase_rast_forward_raw", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastOutputs),
                                         C.POINTER(RastWorkspace), C.c_void_p]),
    ("trase_rast_pair_sizes", C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    ("trase_rast_forward_raw_pair", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastOutputs),
                                              C.POINTER(RastWorkspace), C.POINTER(RastSettings), C.POINTER(RastRawInputs),
                                              C.POINTER(RastOutputs), C.POINTER(RastWorkspace), C.c_void_p, C.c_size_t, C.c_void_p]),
    ("trase_rast_graph_mode", C.c_int, [C.c_int]),
    ("trase_rast_graph_stats", C.c_int, [C.POINTER(C.c_int64 * 4)]),
    ("trase_rast_render_raw", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastOutputs),
                                        C.POINTER(RastWorkspace), C.c_void_p]),
    ("trase_rast_backward_raw", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastOutputs),
                                          C.POINTER(RastWorkspace), C.POINTER(RastRawGrads), C.c_void_p]),
    ("trase_rast_backward_raw_compose", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastOutputs),
                                                  C.POINTER(RastWorkspace), C.POINTER(RastRawGrads), C.c_void_p]),
    ("trase_rast_zero_live_rows", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastWorkspace),
                                            C.POINTER(RastRawGrads), C.c_void_p]),
    ("trase_rast_backward_raw_gaussians", C.c_int, [C.POINTER(RastSettings), C.POINTER(RastRawInputs), C.POINTER(RastOutputs),
                                                    C.POINTER(RastWorkspace), C.POINTER(RastRawGrads), C.c_int32, C.c_int32,
                                                    C.c_void_p]),
    ("trase_knn_sizes", C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    ("trase_knn_dist2", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_knn_points", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_mlp_sizes", C.c_int, [C.POINTER(C.c_size_t)]),
    ("trase_mlp_forward", C.c_int, [C.POINTER(MlpWeights), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_mlp_train_sizes", C.c_int, [C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("trase_mlp_forward_train", C.c_int, [C.POINTER(MlpWeights), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_mlp_backward", C.c_int, [C.POINTER(MlpWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.POINTER(MlpGrads), C.c_void_p, C.c_size_t, C.c_int32,
                                     C.c_void_p]),
    ("trase_mlp_forward_train_rows", C.c_int, [C.POINTER(MlpWeights), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                               C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_mlp_backward_rows", C.c_int, [C.POINTER(MlpWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.POINTER(MlpGrads), C.c_void_p, C.c_size_t, C.c_int32,
                                          C.c_void_p]),
    ("trase_mlp_live_tiles", C.c_int, [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_void_p]),
    ("trase_smooth_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_smooth_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_loss_sizes", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    ("trase_loss_l1_ssim_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_loss_l1_ssim_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_loss_photometric_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                                 C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_loss_photometric_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_contrastive_sizes", C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    ("trase_contrastive_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_contrastive_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_densify_stats", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("trase_densify_stats_guarded", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                              C.c_int32, C.c_void_p]),
    ("trase_densify_sizes", C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    ("trase_densify_plan", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float,
                                     C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_densify_apply", C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_mask_stats", C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_pairhead_sizes", C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    ("trase_pairhead_forward", C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_pairhead_backward", C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_compact_pixels_sizes", C.c_int, [C.c_int64, C.POINTER(C.c_size_t)]),
    ("trase_compact_pixels", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32,
                                       C.c_void_p]),
    ("trase_pairhead_forward_n", C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_pairhead_backward_n", C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_int32,
                                            C.c_void_p]),
    ("trase_featnorm_sizes", C.c_int, [C.c_int64, C.POINTER(C.c_size_t)]),
    ("trase_featnorm_forward", C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    ("trase_featnorm_backward", C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_nnfm_sizes", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    ("trase_nnfm_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_int32, C.c_void_p]),
    ("trase_nnfm_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_adam_step", C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_double, C.c_double, C.c_float, C.c_int32, C.c_void_p]),
    ("trase_adam_step_guarded", C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_double, C.c_double, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    ("trase_prof_enable", C.c_int, [C.c_int]),
    ("trase_prof_report", C.c_int, [C.c_char_p, C.c_size_t]),
    ("trase_selftest", C.c_int, [C.c_int32, C.c_void_p, C.c_char_p, C.c_size_t]),
    ("trase_last_error", C.c_char_p, []),
    ("trase_version", C.c_char_p, []),
]

_lib = None


def load() -> C.CDLL:
    """Load (once) and return the library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"trase_amd: HIP library not built: {LIB_PATH} is missing. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C trase_amd/csrc`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        if os.environ.get("TRASE_RAST_LIB") and os.environ.get("TRASE_RAST_LIB_AB") and not hasattr(lib, name):
            continue                # an OLDER build loaded for a same-box A/B (both variables set): entry points added since are absent
        fn = getattr(lib, name)     # AttributeError here == header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return load().trase_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        msg = last_error()
        if rc == -1:
            # the reference wrapper raises plain Exception for bad argument combinations
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
