"""Drop-in import name for GauStudio (gaustudio/renderers/base.py:7 does
`from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`).

Nothing lives here: the operator is gaustudio_amd (hand-written HIP for MI355X behind the
reference's Python interface); this package only re-exports it under the reference's module name,
including the `_C` submodule with its three entry points.
"""
from gaustudio_amd import _C  # noqa: F401
from gaustudio_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _RasterizeGaussians,
    rasterize_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_RasterizeGaussians", "_C"]
