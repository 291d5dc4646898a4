"""``pytorch3d.ops.knn_points`` as imported at scene/gaussian_model.py:32 and utils/loss_utils.py:26."""
from trase_amd.rasterizer import knn_points  # noqa: F401

__all__ = ["knn_points"]
