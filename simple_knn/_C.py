"""``simple_knn._C.distCUDA2`` backed by the HIP spatial-hash KNN (trase_amd/csrc/knn.hip)."""
from trase_amd.rasterizer import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
