"""`_C`: the operator layer the reference exposes through pybind
($RAST/ext.cpp:15-19 -> $RAST/rasterize_points.cu:35-231), re-created over the C ABI of libgsrast.so
(include/gsrast.h).  Same three entry points, same argument order, same return tuples:

    rasterize_gaussians(...)           -> (num_rendered, color, depth, median, opacity, radii, geomBuffer, binningBuffer, imgBuffer)
    rasterize_gaussians_backward(...)  -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]

This module is glue only: shape checks, output allocation through torch's caching allocator, the
current HIP stream.  All compute happens in hand-written HIP kernels inside libgsrast.so.  There is
no CPU or PyTorch fallback: if the library is missing, or the tensors are not on a ROCm device, the
calls raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsrast.so")
_lib = None

_ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


def lib():
    """Loads libgsrast.so (built by `make -C gaustudio_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} not found: the HIP extension is not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C gaustudio_amd/csrc`). "
                "gaustudio_amd has no CPU fallback.")
        L = ctypes.CDLL(_LIB_PATH)
        L.gsr_last_error.restype = ctypes.c_char_p
        L.gsr_backward_scratch_bytes.restype = ctypes.c_size_t
        L.gsr_backward_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.gsr_abi_version.restype = ctypes.c_int
        if L.gsr_abi_version() != 2:
            raise ImportError("libgsrast.so ABI version mismatch")
        _lib = L
    return _lib


def _err(L, rc):
    msg = L.gsr_last_error()
    msg = msg.decode() if msg else ""
    return RuntimeError(f"{msg} [gsrast rc={rc}]")


def _ptr(t):
    """Device (or host) address of a tensor, NULL for the reference's "absent" convention: an empty
    tensor (`torch.Tensor([])`, $RAST/.../__init__.py:200-210) has a null data pointer."""
    if t is None or t.numel() == 0:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t, name):
    if t is None:
        return None
    if t.numel() and t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on '{t.device}': gaustudio_amd runs on ROCm devices only (hand-written HIP kernels, "
            "no CPU fallback)")


class _Buf:
    """Allocator callback target: the opaque byte tensors the reference resizes through
    resizeFunctional (rasterize_points.cu:27-33)."""

    def __init__(self, device):
        self.device = device
        self.t = torch.empty(0, dtype=torch.uint8, device=device)
        self.cb = _ALLOC_FN(self._alloc)

    def _alloc(self, _ctx, n):
        try:
            self.t = torch.empty(int(n), dtype=torch.uint8, device=self.device)
            return self.t.data_ptr()
        except Exception:   # pragma: no cover - OOM surfaces as GSR_ERR_ALLOC
            return 0


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                        campos, prefiltered, debug):
    """RasterizeGaussiansCUDA, rasterize_points.cu:35-121."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:57-59
    L = lib()
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(image_height), int(image_width)
    means3D = _f32c(means3D, "means3D")
    colors = _f32c(colors, "colors_precomp"); opacity = _f32c(opacity, "opacities")
    scales = _f32c(scales, "scales"); rotations = _f32c(rotations, "rotations")
    cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp"); sh = _f32c(sh, "sh")
    viewmatrix = _f32c(viewmatrix, "viewmatrix"); projmatrix = _f32c(projmatrix, "projmatrix")
    campos = _f32c(campos, "campos"); background = _f32c(background, "bg")
    if colors.numel() and (colors.ndimension() != 2 or colors.size(1) != 3):
        raise RuntimeError("colors_precomp must have dimensions (num_points, 3)")   # NUM_CHANNELS == 3, config.h:15
    for t, name in ((colors, "colors_precomp"), (opacity, "opacities"), (scales, "scales"), (rotations, "rotations"),
                    (cov3D_precomp, "cov3D_precomp"), (sh, "sh")):
        if t.numel():
            _require_device(t, name)

    with torch.cuda.device(dev):
        fo = dict(dtype=torch.float32, device=dev)
        out_color = torch.empty((3, H, W), **fo)
        out_depth = torch.empty((1, H, W), **fo)
        out_median = torch.empty((3, H, W), **fo)
        out_opacity = torch.empty((1, H, W), **fo)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom, binning, img = _Buf(dev), _Buf(dev), _Buf(dev)
        M = sh.size(1) if sh.numel() != 0 and sh.size(0) != 0 else 0     # rasterize_points.cu:86-90
        rc = L.gsr_forward(geom.cb, None, binning.cb, None, img.cb, None,
                           ctypes.c_int(P), ctypes.c_int(int(degree)), ctypes.c_int(M), _ptr(background),
                           ctypes.c_int(W), ctypes.c_int(H), _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity),
                           _ptr(scales), ctypes.c_float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                           _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), ctypes.c_float(tan_fovx),
                           ctypes.c_float(tan_fovy), ctypes.c_int(bool(prefiltered)), _ptr(out_color),
                           _ptr(out_depth), _ptr(out_median), _ptr(out_opacity), _ptr(radii),
                           ctypes.c_int(bool(debug)), _stream(dev))
    if rc < 0:
        raise _err(L, rc)
    return rc, out_color, out_depth, out_median, out_opacity, radii, geom.t, binning.t, img.t


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_median_depth, dL_dout_final_opacity, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA, rasterize_points.cu:123-210."""
    L = lib()
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 and sh.size(0) != 0 else 0
    means3D = _f32c(means3D, "means3D"); colors = _f32c(colors, "colors_precomp")
    scales = _f32c(scales, "scales"); rotations = _f32c(rotations, "rotations")
    cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp"); sh = _f32c(sh, "sh")
    background = _f32c(background, "bg")
    g_color = _f32c(dL_dout_color, "dL_dout_color"); g_depth = _f32c(dL_dout_depth, "dL_dout_depth")
    g_median = _f32c(dL_dout_median_depth, "dL_dout_median_depth")
    g_op = _f32c(dL_dout_final_opacity, "dL_dout_final_opacity")
    radii = radii.contiguous()
    with torch.cuda.device(dev):
        fo = dict(dtype=torch.float32, device=dev)
        dL_dmeans3D = torch.empty((P, 3), **fo)
        dL_dmeans2D = torch.empty((P, 3), **fo)
        dL_dcolors = torch.empty((P, 3), **fo)
        dL_dopacity = torch.empty((P, 1), **fo)
        dL_dcov3D = torch.empty((P, 6), **fo)
        dL_dsh = torch.empty((P, M, 3), **fo)
        dL_dscales = torch.empty((P, 3), **fo)
        dL_drotations = torch.empty((P, 4), **fo)
        if P != 0:
            scratch = torch.empty(L.gsr_backward_scratch_bytes(ctypes.c_int(P), ctypes.c_int(int(R))), dtype=torch.uint8,
                                  device=dev)
            rc = L.gsr_backward(ctypes.c_int(P), ctypes.c_int(int(degree)), ctypes.c_int(M), ctypes.c_int(int(R)),
                                _ptr(background), ctypes.c_int(W), ctypes.c_int(H), _ptr(means3D), _ptr(sh),
                                _ptr(colors), _ptr(scales), ctypes.c_float(scale_modifier), _ptr(rotations),
                                _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos),
                                ctypes.c_float(tan_fovx), ctypes.c_float(tan_fovy), _ptr(radii),
                                _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(g_color),
                                _ptr(g_depth), _ptr(g_median), _ptr(g_op), _ptr(dL_dmeans2D), _ptr(dL_dopacity),
                                _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh),
                                _ptr(dL_dscales), _ptr(dL_drotations), _ptr(scratch), ctypes.c_int(bool(debug)),
                                _stream(dev))
            if rc < 0:
                raise _err(L, rc)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, rasterize_points.cu:212-231."""
    L = lib()
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    means3D = _f32c(means3D, "means3D")
    viewmatrix = _f32c(viewmatrix, "viewmatrix"); projmatrix = _f32c(projmatrix, "projmatrix")
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        with torch.cuda.device(dev):
            rc = L.gsr_mark_visible(ctypes.c_int(P), _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix),
                                    _ptr(present), _stream(dev))
        if rc < 0:
            raise _err(L, rc)
    return present


# ---- introspection of the opaque buffers (tests / debugging; include/gsrast.h gsr_inspect_*) ----

def inspect_geometry(geomBuffer, radii):
    L = lib()
    dev = radii.device
    P = radii.numel()
    out = dict(means2D=torch.empty((P, 2), dtype=torch.float32, device=dev),
               depths=torch.empty((P,), dtype=torch.float32, device=dev),
               conic_opacity=torch.empty((P, 4), dtype=torch.float32, device=dev),
               rgb=torch.empty((P, 3), dtype=torch.float32, device=dev),
               clamped=torch.empty((P, 3), dtype=torch.uint8, device=dev),
               tiles_touched=torch.empty((P,), dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        rc = L.gsr_inspect_geometry(_ptr(geomBuffer), ctypes.c_int(P), _ptr(radii), _ptr(out["means2D"]),
                                    _ptr(out["depths"]), _ptr(out["conic_opacity"]), _ptr(out["rgb"]),
                                    _ptr(out["clamped"]), _ptr(out["tiles_touched"]), _stream(dev))
    if rc < 0:
        raise _err(L, rc)
    return out


def inspect_binning(binningBuffer, imgBuffer, R, width, height):
    L = lib()
    dev = imgBuffer.device
    T = ((width + 15) // 16) * ((height + 15) // 16)
    point_list = torch.empty((max(int(R), 0),), dtype=torch.int32, device=dev)
    ranges = torch.empty((T, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.gsr_inspect_binning(_ptr(binningBuffer), _ptr(imgBuffer), ctypes.c_int(int(R)), ctypes.c_int(width),
                                   ctypes.c_int(height), _ptr(point_list), _ptr(ranges), _stream(dev))
    if rc < 0:
        raise _err(L, rc)
    return point_list, ranges


def inspect_image(imgBuffer, width, height):
    L = lib()
    dev = imgBuffer.device
    final_T = torch.empty((height, width), dtype=torch.float32, device=dev)
    n_contrib = torch.empty((height, width), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.gsr_inspect_image(_ptr(imgBuffer), ctypes.c_int(width), ctypes.c_int(height), _ptr(final_T),
                                 _ptr(n_contrib), _stream(dev))
    if rc < 0:
        raise _err(L, rc)
    return final_T, n_contrib


def set_profiling(enable):
    lib().gsr_set_profiling(ctypes.c_int(bool(enable)))


def last_forward_ms():
    """Mean per-stage GPU milliseconds over the forward calls since set_profiling(True) (HIP events
    recorded on the launch stream); None if nothing was recorded."""
    a = (ctypes.c_float * 5)()
    n = lib().gsr_last_forward_ms(a)
    if not n:
        return None
    d = dict(zip(("preprocess", "scan", "scatter", "sort", "composite"), list(a)))
    d["calls"] = n
    return d


def last_backward_ms():
    a = (ctypes.c_float * 2)()
    n = lib().gsr_last_backward_ms(a)
    if not n:
        return None
    d = dict(zip(("composite_bwd", "preprocess_bwd"), list(a)))
    d["calls"] = n
    return d
