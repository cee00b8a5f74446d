"""gaustudio_amd: MI355X-native (gfx950) differentiable 3D-Gaussian rasterizer behind GauStudio's
`gaustudio_diff_gaussian_rasterization` operator interface.  See DESIGN.md."""
from .rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _RasterizeGaussians,
    rasterize_gaussians,
)
from .options import options  # noqa: F401,E402  (per-call options: tile band, fast_exp, kernel A/B switches)

__version__ = "0.1.0"
