"""Fused parameter activations in front of the operator (NEW interface, not in the reference).

gaustudio's VanillaRenderer gathers the point cloud through `get_attribute` (gaustudio/renderers/vanilla_renderer.py:
28-51 -> models/vanilla_sg.py:58-63,103-106): exp(_scale), sigmoid(_opacity), F.normalize(_rot) and
torch.cat((_f_dc, _f_rest)) materialise four activated copies (~240 B per Gaussian) on every render, and autograd
runs four more kernels backwards.  `FusedGaussianRasterizer` takes the RAW attributes instead; the activations
and their chain rules run inside preprocess_fwd / preprocess_bwd, and the SH coefficients are read from the two
tensors they are stored in.

    rasterizer = FusedGaussianRasterizer(raster_settings)              # same settings tuple
    color, radii, depth, median, opacity = rasterizer(
        means3D=pcd._xyz, means2D=screenspace_points, raw_opacities=pcd._opacity,
        f_dc=pcd._f_dc, f_rest=pcd._f_rest, raw_scales=pcd._scale, raw_rotations=pcd._rot)

Gradients arrive on the raw tensors.  Results equal the unfused path (activations by torch, then
GaussianRasterizer) up to the last-bit differences between torch's and the kernel's exp / sqrt.
"""
import torch
import torch.nn as nn

from . import _C
from .options import clear_grad_mode as _clear_grad_mode, for_forward as _options_for_forward, note_grad_mode as _note_grad_mode
from .rasterizer import _image_grads, _run_guarded

ACT_OPACITY_SIGMOID = 1
ACT_SCALE_EXP = 2
ACT_ROT_NORMALIZE = 4
ACT_VANILLA = ACT_OPACITY_SIGMOID | ACT_SCALE_EXP | ACT_ROT_NORMALIZE     # VanillaPointCloud.default_conf['activations']


class _RasterizeGaussiansRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, f_dc, f_rest, raw_opacities, raw_scales, raw_rotations, raster_settings, act):
        rs = raster_settings
        n = _C.native()
        opts = [int(v) for v in _options_for_forward(any(ctx.needs_input_grad))]       # per-call options, kept with the graph (options.py)
        call = (rs.bg, means3D, f_dc, f_rest, raw_opacities, raw_scales, raw_rotations, float(rs.scale_modifier), int(act),
                rs.viewmatrix, rs.projmatrix, float(rs.tanfovx), float(rs.tanfovy), int(rs.image_height),
                int(rs.image_width), int(rs.sh_degree), rs.campos, bool(rs.prefiltered), bool(rs.debug), opts)
        (num_rendered, color, depth, median, opacity, radii, geom, binning, img) = _run_guarded(
            n.rasterize_gaussians_raw, call, rs.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.raster_settings, ctx.num_rendered, ctx.act, ctx.gsr_options = rs, num_rendered, int(act), opts
        ctx.save_for_backward(means3D, f_dc, f_rest, raw_scales, raw_rotations, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)      # as in rasterizer._RasterizeGaussians
        ctx.set_materialize_grads(False)
        return color, radii, depth, median, opacity

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_median, g_opacity):
        rs = ctx.raster_settings
        means3D, f_dc, f_rest, raw_scales, raw_rotations, radii, geom, binning, img = ctx.saved_tensors
        g_color, g_depth, g_median, g_opacity = _image_grads(g_color, g_depth, g_median, g_opacity)
        n = _C.native()
        call = (rs.bg, means3D, radii, f_dc, f_rest, raw_scales, raw_rotations, float(rs.scale_modifier), ctx.act,
                float(rs.tanfovx), float(rs.tanfovy), g_color, g_depth, g_median, g_opacity, int(rs.sh_degree), geom,
                int(ctx.num_rendered), binning, img, bool(rs.debug), ctx.gsr_options, int(rs.image_height), int(rs.image_width))
        (g_means2D, g_op, g_means3D, g_fdc, g_frest, g_scales, g_rot) = _run_guarded(
            n.rasterize_gaussians_raw_backward, call, rs.debug, "snapshot_bw.dump",
            "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        return g_means3D, g_means2D, g_fdc, g_frest, g_op, g_scales, g_rot, None, None


class FusedGaussianRasterizer(nn.Module):
    def __init__(self, raster_settings, activations=ACT_VANILLA):
        super().__init__()
        self.raster_settings = raster_settings
        self.activations = int(activations)

    def forward(self, means3D, means2D, raw_opacities, f_dc, f_rest, raw_scales, raw_rotations):
        P = means3D.shape[0]
        f_dc = f_dc.reshape(P, 1, 3)
        f_rest = f_rest.reshape(P, -1, 3)
        _note_grad_mode(torch.is_grad_enabled())
        try:
            return _RasterizeGaussiansRaw.apply(means3D, means2D, f_dc, f_rest, raw_opacities.reshape(P, 1), raw_scales,
                                                raw_rotations, self.raster_settings, self.activations)
        finally:
            _clear_grad_mode()
