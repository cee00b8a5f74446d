"""`_C`: the operator layer the reference exposes through pybind
($RAST/ext.cpp:15-19 -> $RAST/rasterize_points.cu:35-231), re-created over the C ABI of libgsrast.so
(include/gsrast.h).  Same three entry points, same argument order, same return tuples:

    rasterize_gaussians(...)           -> (num_rendered, color, depth, median, opacity, radii, geomBuffer, binningBuffer, imgBuffer)
    rasterize_gaussians_backward(...)  -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]

The three entry points are implemented natively in csrc/torch_binding.cpp (pybind11 module `_Cnative`, host C++
only: shape checks, output allocation through torch's caching allocator, torch's current HIP stream) over the C
ABI; this Python module forwards to it and adds ctypes access to the library's introspection / profiling entry
points for tests.  All compute happens in hand-written HIP kernels inside libgsrast.so.  There is no CPU or
PyTorch fallback: if the libraries are missing, or the tensors are not on a ROCm device, the calls raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsrast.so")
_lib = None


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
        if L.gsr_abi_version() != 6:
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


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_native = None


def native():
    """The compiled torch adapter gaustudio_amd/_Cnative*.so (csrc/torch_binding.cpp, pybind11): the same role as
    the reference's `_C` extension module.  Linked against libgsrast.so; no fallback if it is missing."""
    global _native
    if _native is None:
        lib()                                            # loud, explicit error if the HIP library is not built
        try:
            from . import _Cnative as n
        except ImportError as e:
            raise ImportError(
                "gaustudio_amd/_Cnative*.so not found or not loadable: build it with `make -C gaustudio_amd/csrc` "
                f"(__graft_entry__.build()). gaustudio_amd has no CPU / pure-Python fallback. [{e}]") from e
        _native = n
    return _native


def _opts(options):
    """Per-call options (gaustudio_amd/options.py): an explicit tuple, or what the calling thread's `with options(...)`
    blocks say."""
    if options is None:
        from .options import current
        options = current()
    return [int(v) for v in options]


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                        campos, prefiltered, debug, options=None):
    """RasterizeGaussiansCUDA, rasterize_points.cu:35-121 -> csrc/torch_binding.cpp RasterizeGaussians.
    `options` (trailing, optional, not in the reference): the per-call options tuple of gaustudio_amd/options.py."""
    return native().rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, float(scale_modifier),
                                        cov3D_precomp, viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy),
                                        int(image_height), int(image_width), sh, int(degree), campos,
                                        bool(prefiltered), bool(debug), _opts(options))


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_median_depth, dL_dout_final_opacity, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug, options=None, image_height=-1, image_width=-1):
    """RasterizeGaussiansBackwardCUDA, rasterize_points.cu:123-210 -> torch_binding.cpp RasterizeGaussiansBackward.
    An upstream gradient may be ABSENT (None, or the reference's empty tensor): the loss does not use that output, its
    gradient is zero and nothing is read for it (include/gsrast.h gsr_backward); `image_height` / `image_width` are only
    needed when all four are absent (the size is otherwise taken from one that is present)."""
    import torch
    e = lambda t: torch.Tensor([]) if t is None else t
    dL_dout_color, dL_dout_depth, dL_dout_median_depth, dL_dout_final_opacity = (
        e(dL_dout_color), e(dL_dout_depth), e(dL_dout_median_depth), e(dL_dout_final_opacity))
    return native().rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations,
                                                 float(scale_modifier), cov3D_precomp, viewmatrix, projmatrix,
                                                 float(tan_fovx), float(tan_fovy), dL_dout_color, dL_dout_depth,
                                                 dL_dout_median_depth, dL_dout_final_opacity, sh, int(degree), campos,
                                                 geomBuffer, int(R), binningBuffer, imageBuffer, bool(debug),
                                                 _opts(options), int(image_height), int(image_width))


def sh_grad_from_colors(means3D, campos, colors, degree, out):
    """out[P,M,3] = sum over the N views (ascending) of basis(dir_r) (x) colors[r]: the SH gradient of a multi-view step from the
    per-view colour gradients (include/gsrast.h gsr_sh_grad_from_colors; gaustudio_amd/parallel.py FactoredGradExchange)."""
    native().sh_grad_from_colors(means3D, campos, colors, int(degree), out)
    return out


# ---- compacted rows for the multi-GPU exchange (include/gsrast.h; csrc/gsr_comm.hip) ------------------------------------
def msg_header_words(P):
    L = lib()
    L.gsr_msg_header_words.restype = ctypes.c_size_t
    return int(L.gsr_msg_header_words(ctypes.c_int(int(P))))


def _rc(L, rc):
    if rc < 0:
        raise _err(L, rc)


def visible_index(radii, msg, counts):
    """Header (K, block bases, mask) of the Gaussians with radii > 0 into the int32 tensor `msg`; counts: int32 scratch [ceil(P/256)]."""
    L = lib()
    with torch.cuda.device(radii.device):
        _rc(L, L.gsr_visible_index(ctypes.c_int(radii.numel()), _ptr(radii), _ptr(msg), _ptr(counts), _stream(radii.device)))


def union_index(P, msgs, offsets, out_hdr, counts):
    """Header of the OR of the N = offsets.numel() messages' masks; message r starts at word offsets[r] (int64, device) of the
    int32 tensor msgs."""
    L = lib()
    with torch.cuda.device(msgs.device):
        _rc(L, L.gsr_union_index(ctypes.c_int(int(P)), ctypes.c_int(int(offsets.numel())), _ptr(msgs), _ptr(offsets),
                                 _ptr(out_hdr), _ptr(counts), _stream(msgs.device)))


def pack_rows(hdr, src, dst, dst_stride, col0=0):
    """dst[row * dst_stride + col0 + c] = src[g, c] for the header's Gaussians g (src float32 [P, C], dst float32 flat)."""
    L = lib()
    P, C = src.shape[0], src.numel() // max(src.shape[0], 1)
    with torch.cuda.device(src.device):
        _rc(L, L.gsr_pack_rows(ctypes.c_int(P), ctypes.c_int(C), _ptr(hdr), _ptr(src), _ptr(dst), ctypes.c_int(int(dst_stride)),
                               ctypes.c_int(int(col0)), _stream(src.device)))


def unpack_rows(hdr, src, src_stride, col0, dst):
    """dst[g, c] = src[row * src_stride + col0 + c] for the header's Gaussians g; the other rows of dst stay as they are."""
    L = lib()
    P, C = dst.shape[0], dst.numel() // max(dst.shape[0], 1)
    with torch.cuda.device(dst.device):
        _rc(L, L.gsr_unpack_rows(ctypes.c_int(P), ctypes.c_int(C), _ptr(hdr), _ptr(src), ctypes.c_int(int(src_stride)),
                                 ctypes.c_int(int(col0)), _ptr(dst), _stream(dst.device)))


def pack_geometry(hdr, g_means3D, g_opacity, g_scales, g_rotations, rows, flag):
    """rows[r, 0:11] = [means3D 3 | opacity 1 | scales 3 | rotations 4] gradients of the r-th Gaussian of the header; flag (int32[1],
    cleared by the caller) |= 1 when a Gaussian outside the header has a non-zero value."""
    L = lib()
    P = g_means3D.shape[0]
    with torch.cuda.device(rows.device):
        _rc(L, L.gsr_pack_geometry(ctypes.c_int(P), _ptr(hdr), _ptr(g_means3D), _ptr(g_opacity), _ptr(g_scales), _ptr(g_rotations),
                                   _ptr(rows), _ptr(flag), _stream(rows.device)))


def unpack_geometry(hdr, rows, g_means3D, g_opacity, g_scales, g_rotations):
    L = lib()
    P = g_means3D.shape[0]
    with torch.cuda.device(rows.device):
        _rc(L, L.gsr_unpack_geometry(ctypes.c_int(P), _ptr(hdr), _ptr(rows), _ptr(g_means3D), _ptr(g_opacity), _ptr(g_scales),
                                     _ptr(g_rotations), _stream(rows.device)))


def sh_grad_from_packed(means3D, campos, msgs, offsets, degree, out):
    """sh_grad_from_colors reading N = offsets.numel() packed messages (header + rows of 3 floats, message r at word offsets[r] of the
    int32 tensor msgs) instead of dense colours."""
    L = lib()
    P, M = out.shape[0], out.shape[1]
    with torch.cuda.device(out.device):
        _rc(L, L.gsr_sh_grad_from_packed(ctypes.c_int(P), ctypes.c_int(int(degree)), ctypes.c_int(M), ctypes.c_int(int(offsets.numel())),
                                         _ptr(means3D), _ptr(campos), _ptr(msgs), _ptr(offsets), _ptr(out), _stream(out.device)))
    return out


def set_grad_arena(outs, keys=(), sh_chunks=1, hook=None, colors_out=None, band_split=0, band_hook=None, class_hook=None):
    """One-shot destination tensors [dL_dmeans3D, dL_dsh, dL_dopacity, dL_dscales, dL_drotations] for the next
    rasterize_gaussians_backward whose inputs [means3D, sh, scales, rotations] have the data pointers `keys`
    (0 / empty = any); see gaustudio_amd/parallel.py.  With sh_chunks > 1 and a hook, the SH stage of that backward
    runs in Gaussian ranges and hook(c, g0, g1) is called after range c has been enqueued.  colors_out [P,3]: that
    backward runs its SH stage in the factored form -- the clamp-masked colour gradient goes there and the returned
    dL_dsh is None (outs may then be [] or five tensors whose second is ignored); a `hook` given together with colors_out is
    called (no arguments) once the geometry stage -- which then already leaves the masked colour gradient in colors_out -- has
    been enqueued and BEFORE the SH-direction stage: the caller starts the all-gather of the slot there.
    band_split > 0 with band_hook (and colors_out + hook): that backward runs BANDED (include/gsrast.h GSR_BWD_PART_BAND_*): the image
    is cut at tile row band_split; class_hook(first[P], second[P]) receives the two Gaussian classes of the cut from the FORWARD of these
    parameters (on the caller's thread, right behind the forward's kernels),
    band_hook() is called once the first band -- compositing above the cut + the per-Gaussian stage of the Gaussians that end there,
    whose rows of colors_out are final then -- has been enqueued, hook() after the second band.  Same bits as the unbanded backward.
    [] without colors_out disarms."""
    native().set_grad_arena(list(outs), [int(k) for k in keys], int(sh_chunks), hook, colors_out, int(band_split), band_hook, class_hook)


def band_classes(radii, geom_buffer, split_tile_row):
    """(first[P], second[P]) int32: the two Gaussian classes of a banded backward cut at tile row `split_tile_row`
    (include/gsrast.h gsr_band_classes)."""
    import torch
    L = lib()
    P = int(radii.shape[0])
    first = torch.empty(P, dtype=torch.int32, device=radii.device)
    second = torch.empty(P, dtype=torch.int32, device=radii.device)
    with torch.cuda.device(radii.device):
        rc = L.gsr_band_classes(ctypes.c_int(P), _ptr(radii), _ptr(geom_buffer), ctypes.c_int(int(split_tile_row)), _ptr(first), _ptr(second),
                                _stream(radii.device))
    if rc < 0:
        raise _err(L, rc)
    return first, second


def set_option(name, value):
    """Process-wide tunables of libgsrast.so (include/gsrast.h gsr_set_option).  Only `fast_exp` (the process default of the
    compositing kernels' exp: 1 = v_exp_f32, the default; 0 = the reproducible polynomial, bit-identical to the CPU oracle)
    changes a result bit; the autograd Functions resolve it at forward time and keep it with the graph (options.resolved)."""
    L = lib()
    rc = L.gsr_set_option(name.encode(), ctypes.c_int(int(value)))
    if rc < 0:
        raise _err(L, rc)


def get_option(name):
    return int(lib().gsr_get_option(name.encode()))


def selftest(device=None):
    """Bit mask of the device arithmetic identities composite_bwd relies on (3 = all hold)."""
    L = lib()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        rc = L.gsr_selftest(_stream(dev))
    if rc < 0:
        raise _err(L, rc)
    return rc


def inspect_counts(imgBuffer, width, height):
    """{'num_binned': instances actually binned (length of point_list), 'max_tile': longest tile list,
    'num_rendered': the reference-defined count rasterize_gaussians returned}."""
    L = lib()
    dev = imgBuffer.device
    out = (ctypes.c_uint32 * 4)()
    with torch.cuda.device(dev):
        rc = L.gsr_inspect_counts(_ptr(imgBuffer), ctypes.c_int(width), ctypes.c_int(height), out, _stream(dev))
    if rc < 0:
        raise _err(L, rc)
    return dict(num_binned=int(out[0]), max_tile=int(out[1]), num_rendered=int(out[2]), overflow=int(out[3]))


def inspect_staged(imgBuffer, width, height):
    """Instances composite_fwd actually staged for the frame (it stops fetching a tile's list once every pixel has saturated)."""
    import ctypes
    L = lib()
    out = ctypes.c_ulonglong(0)
    _rc(L, L.gsr_inspect_staged(_ptr(imgBuffer), ctypes.c_int(int(width)), ctypes.c_int(int(height)), ctypes.byref(out), _stream(imgBuffer.device)))
    return int(out.value)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, rasterize_points.cu:212-231 -> torch_binding.cpp markVisible."""
    return native().mark_visible(means3D, viewmatrix, projmatrix)


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
    """False / 0: off.  True / 1: one HIP event at every stage boundary of every forward / backward (nine per step: ~3 % of a C3
    step).  2: only the two boundaries of composite_fwd, the north-star kernel (what bench.py's timed steps record)."""
    lib().gsr_set_profiling(ctypes.c_int(int(enable)))


def last_forward_ms():
    """Mean per-stage GPU milliseconds over the forward calls since set_profiling(...) (HIP events
    recorded on the launch stream; a stage that was not recorded -- level 2 records `composite` only -- reads 0); None if
    nothing was recorded."""
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
