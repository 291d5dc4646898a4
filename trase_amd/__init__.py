"""trase_amd -- MI355X-native (gfx950) differentiable Gaussian rasterizer path of yunjinli/TRASE.

Only what the hot path needs: ``csrc/`` (HIP kernels + C ABI, include/trase_rast.h),
``rasterizer`` (host-side mirror of the reference's operator interface),
``synthetic`` (seeded scenes/cameras for tests and bench).
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, distCUDA2,  # noqa: F401
                         rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "distCUDA2"]
