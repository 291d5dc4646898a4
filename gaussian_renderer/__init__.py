"""Drop-in for the reference's ``gaussian_renderer`` package (``from gaussian_renderer import render`` at
train.py:24, train_style_transfer_nnfm.py:24, render.py:25, gui.py:23): put this repository BEFORE the
reference checkout on PYTHONPATH and the training / rendering scripts run unmodified on the fused
MI355X path."""
from trase_amd.renderer import render  # noqa: F401

try:   # render.py:30 imports GaussianModel through this module; available when run inside the reference tree
    from scene.gaussian_model import GaussianModel  # noqa: F401
except Exception:   # pragma: no cover
    pass

__all__ = ["render"]
