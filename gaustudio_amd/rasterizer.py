"""Host-side mirror of the reference's operator interface
($RAST/gaustudio_diff_gaussian_rasterization/__init__.py): the 12-field settings tuple (:160-172),
the nn.Module front door (:174-223) and the autograd Function (:44-158), so that
gaustudio/renderers/base.py:7,23-49 works against the MI355X op unchanged.

Everything numeric happens in gaustudio_amd._C -> libgsrast.so (hand-written HIP, gfx950).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C
from .options import clear_grad_mode as _clear_grad_mode, for_forward as _options_for_forward, note_grad_mode as _note_grad_mode


class GaussianRasterizationSettings(NamedTuple):
    # field order is part of the contract (positional construction works in the reference)
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _to_cpu_snapshot(values):
    """Debug mode keeps a host copy of every argument so a failing call can be replayed
    (reference: cpu_deep_copy_tuple, __init__.py:17-19)."""
    return tuple(v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v for v in values)


def _run_guarded(fn, args, debug, dump_name, banner):
    if not debug:
        return fn(*args)
    snapshot = _to_cpu_snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(snapshot, dump_name)
        print(banner)
        raise


def _image_grads(g_color, g_depth, g_median, g_opacity):
    """Upstream gradients autograd did not produce (outputs the loss did not use: set_materialize_grads(False) hands over None)
    are passed on as ABSENT -- the reference's empty-tensor convention -- and the library reads nothing for them: no zero
    planes are materialised (the reference's kernels read all four, backward.cu:476-483, and autograd fills zeros for it:
    41 MB per 1080p step for a colour-only loss)."""
    e = _absent()
    return tuple(e if g is None else g for g in (g_color, g_depth, g_median, g_opacity))


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        # per-call options of the calling thread (gaustudio_amd/options.py; all -1 = process defaults unless a
        # `with options(...)` block is active): they stay with the graph, the backward below runs with the same ones
        opts = _options_for_forward(any(ctx.needs_input_grad))
        # argument order of _C.rasterize_gaussians (rasterize_points.h:18-38), then the options
        call = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, opts)
        (num_rendered, color, depth, median_depth, final_opacity, radii, geom_buf, binning_buf,
         img_buf) = _run_guarded(
            _C.rasterize_gaussians, call, rs.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.raster_settings = rs
        ctx.gsr_options = opts
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf,
                              binning_buf, img_buf)
        # radii is an int tensor: no gradient, and no zero tensor materialised for it before every backward (autograd
        # would fill P ints per call); outputs the loss did not use arrive as None and are replaced below
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, depth, median_depth, final_opacity

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_median_depth, grad_final_opacity):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf,
         img_buf) = ctx.saved_tensors
        grad_out_color, grad_depth, grad_median_depth, grad_final_opacity = _image_grads(
            grad_out_color, grad_depth, grad_median_depth, grad_final_opacity)
        # argument order of _C.rasterize_gaussians_backward (rasterize_points.h:40-65)
        call = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth,
                grad_median_depth, grad_final_opacity, sh, rs.sh_degree, rs.campos, geom_buf, ctx.num_rendered,
                binning_buf, img_buf, rs.debug, ctx.gsr_options, int(rs.image_height), int(rs.image_width))
        (g_means2D, g_colors, g_opacities, g_means3D, g_cov3D, g_sh, g_scales, g_rotations) = _run_guarded(
            _C.rasterize_gaussians_backward, call, rs.debug, "snapshot_bw.dump",
            "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        # back to the input order of forward(); the settings tuple gets no gradient
        if g_cov3D.numel() == 0:      # covariances built from scale / rotation: the library did not write their gradient
            g_cov3D = None
        return (g_means3D, g_means2D, g_sh, g_colors, g_opacities, g_scales, g_rotations, g_cov3D, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    _note_grad_mode(torch.is_grad_enabled())      # torch.no_grad(): a forward-only call (nothing kept for a backward)
    try:
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings)
    finally:
        _clear_grad_mode()                        # (the note is one-shot; a forward that raised early must not leave it behind)


def _absent():
    # the reference's "not provided" marker: an empty CPU float tensor whose data_ptr is null
    return torch.Tensor([])


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: Gaussians passing the near-plane test for this camera (__init__.py:179-188)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        have_sh, have_col = shs is not None, colors_precomp is not None
        if have_sh == have_col:
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr_any = scales is not None or rotations is not None
        have_sr_all = scales is not None and rotations is not None
        have_cov = cov3D_precomp is not None
        if (not have_sr_all and not have_cov) or (have_sr_any and have_cov):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(
            means3D, means2D,
            shs if have_sh else _absent(),
            colors_precomp if have_col else _absent(),
            opacities,
            scales if scales is not None else _absent(),
            rotations if rotations is not None else _absent(),
            cov3D_precomp if have_cov else _absent(),
            self.raster_settings)
