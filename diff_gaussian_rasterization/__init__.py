"""Drop-in module name the reference imports (gaussian_renderer/__init__.py:21,
gui_standalone.py:59): ``from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer``.  Thin re-export of the
MI355X implementation in trase_amd."""
from trase_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                  rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
