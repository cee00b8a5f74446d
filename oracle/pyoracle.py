"""ctypes front-end of the CPU ORACLE (oracle/gsr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (gaustudio_amd / gaustudio_diff_gaussian_rasterization) never does.

`forward()` / `backward()` mirror the argument meaning of the reference's
CudaRasterizer::Rasterizer::forward / backward
($RAST/cuda_rasterizer/rasterizer_impl.cu:198-343, :347-452) with numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gsr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_exp.restype = ctypes.c_float
        _LIB.orc_exp.argtypes = [ctypes.c_float]
        _LIB.orc_count_rendered.restype = ctypes.c_int64
        _LIB.orc_rects.restype = ctypes.c_int64
        _LIB.orc_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a if a.size else None


def exp(p):
    L = lib()
    p = np.asarray(p, dtype=np.float32)
    return np.array([L.orc_exp(ctypes.c_float(float(v))) for v in p.ravel()], dtype=np.float32).reshape(p.shape)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(ctypes.c_int(int(n)))


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    if P:
        lib().orc_mark_visible(P, _p(means3D), _p(_f32(viewmatrix).ravel()), _p(_f32(projmatrix).ravel()), _p(out))
    return out.astype(bool)


def forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree=0,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            scale_modifier=1.0, prefiltered=False, bg=None, tile_step=1, tight=False):
    """Returns a dict with the five outputs + radii + every intermediate buffer (the state the
    reference keeps in geomBuffer / binningBuffer / imgBuffer).

    tight=False follows the reference exactly (getRect squares).  tight=True bins into the product's tight rects minus their
    dead corner tiles (gs_tight_rect / gs_dead_corners, restated in gsr_oracle.c:orc_rects): `num_rendered` stays the reference's count, `num_binned`,
    `tiles_touched`, `point_list`, `ranges`, `n_contrib` describe the shorter lists; images must not change."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    shs = _f32(shs); colors_precomp = _f32(colors_precomp); scales = _f32(scales)
    rotations = _f32(rotations); cov3D_precomp = _f32(cov3D_precomp); opacities = _f32(opacities)
    view = _f32(viewmatrix).ravel(); proj = _f32(projmatrix).ravel(); cam = _f32(campos).ravel()
    M = 0 if shs is None else shs.shape[1]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st = dict(P=P, W=W, H=H, M=M, D=sh_degree)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), np.float32)
    st["depths"] = np.zeros(P, np.float32)
    st["cov3D"] = np.zeros((P, 6), np.float32)
    st["rgb"] = np.zeros((P, 3), np.float32)
    st["conic_opacity"] = np.zeros((P, 4), np.float32)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    rc = L.orc_preprocess(P, int(sh_degree), M, _p(means3D), _p(scales), ctypes.c_float(scale_modifier),
                          _p(rotations), _p(opacities), _p(shs), _p(cov3D_precomp), _p(colors_precomp),
                          _p(view), _p(proj), _p(cam), int(W), int(H), ctypes.c_float(tanfovx),
                          ctypes.c_float(tanfovy), int(bool(prefiltered)), _p(st["radii"]), _p(st["means2D"]),
                          _p(st["depths"]), _p(st["cov3D"]), _p(st["rgb"]), _p(st["conic_opacity"]),
                          _p(st["tiles_touched"]), _p(st["clamped"]))
    if rc != 0:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    R = int(L.orc_count_rendered(P, _p(st["tiles_touched"])))
    st["num_rendered"] = R
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    if tight:
        st["rects"] = np.zeros((P, 4), np.int32)
        st["tiles_touched"] = np.zeros(P, np.uint32)
        st["dead_corners"] = np.zeros(P, np.uint8)       # bits 0..3: TL, TR, BL, BR tile of the rect not binned
        R = int(L.orc_rects(P, int(W), int(H), _p(st["means2D"]), _p(st["conic_opacity"]), _p(st["radii"]), 1,
                            _p(st["rects"]), _p(st["tiles_touched"]), _p(st["dead_corners"])))
        pl = np.zeros(max(R, 1), np.uint32)
        L.orc_bin_sort_rects(P, int(W), int(H), _p(st["depths"]), _p(st["rects"]), _p(st["dead_corners"]),
                             ctypes.c_int64(R), _p(pl), _p(st["ranges"]))
    else:
        pl = np.zeros(max(R, 1), np.uint32)
        L.orc_bin_sort(P, int(W), int(H), _p(st["means2D"]), _p(st["depths"]), _p(st["radii"]),
                       ctypes.c_int64(R), _p(pl), _p(st["ranges"]))
    st["num_binned"] = R
    st["point_list"] = pl[:R]
    st["_pl_full"] = pl
    feat = colors_precomp if colors_precomp is not None else st["rgb"]
    st["features"] = feat
    st["color"] = np.zeros((3, H, W), np.float32)
    st["depth"] = np.zeros((1, H, W), np.float32)
    st["median"] = np.zeros((3, H, W), np.float32)
    st["opacity"] = np.zeros((1, H, W), np.float32)
    st["final_T"] = np.zeros((H, W), np.float32)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    L.orc_composite_fwd(int(W), int(H), _p(st["ranges"]), _p(pl), _p(st["means2D"]), _p(feat),
                        _p(st["depths"]), _p(st["conic_opacity"]), _p(st["color"]), _p(st["depth"]),
                        _p(st["median"]), _p(st["opacity"]), _p(st["final_T"]), _p(st["n_contrib"]),
                        int(tile_step))
    st["_inputs"] = dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales,
                         rotations=rotations, cov3D_precomp=cov3D_precomp, opacities=opacities, view=view,
                         proj=proj, cam=cam, tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier,
                         bg=np.zeros(3, np.float32) if bg is None else _f32(bg).ravel())
    return st


def backward(st, grad_color, grad_depth, grad_median, grad_opacity, tile_step=1, want_abs=True):
    """Backward for a state returned by forward().  Returns the reference's 8 gradient tensors
    (rasterize_points.cu:209) plus the composite-stage accumulators in double ("acc") and the
    sum of the contributions' magnitudes ("accabs": for cancelling terms the magnitude of the operands), component
    order documented in gsr_oracle.c, and "flip9": per Gaussian the
    |dL_dmedian| of ill-conditioned median-threshold events (slack for component 9)."""
    L = lib()
    inp = st["_inputs"]
    P, W, H, M, D = st["P"], st["W"], st["H"], st["M"], st["D"]
    g_color = _f32(grad_color).reshape(3, H, W)
    g_depth = _f32(grad_depth).reshape(H, W)
    g_median = _f32(grad_median).reshape(3, H, W)
    g_op = _f32(grad_opacity).reshape(H, W)
    acc = np.zeros((P, 10), np.float64)
    accabs = np.zeros((P, 10), np.float64) if want_abs else None
    flip9 = np.zeros(P, np.float64) if want_abs else None
    L.orc_composite_bwd(int(W), int(H), _p(inp["bg"]), _p(st["ranges"]), _p(st["_pl_full"]), _p(st["means2D"]),
                        _p(st["conic_opacity"]), _p(st["features"]), _p(st["depths"]), _p(st["final_T"]),
                        _p(st["n_contrib"]), _p(g_color), _p(g_depth), _p(g_median), _p(g_op), _p(acc),
                        _p(accabs), _p(flip9), int(tile_step))
    out = finish_backward(st, acc.astype(np.float32))
    out["acc"] = acc
    out["accabs"] = accabs
    out["flip9"] = flip9        # |dL_dmedian| of median-threshold events within rounding distance of 0.5, per Gaussian
    return out


def finish_backward(st, a32):
    """Per-Gaussian half of the backward (computeCov2DCUDA + preprocessCUDA, backward.cu:144-274,346-412)
    from composite-stage sums a32[P,10] (component order of gsr_oracle.c) given in float32."""
    L = lib()
    inp = st["_inputs"]
    P, W, H, M, D = st["P"], st["W"], st["H"], st["M"], st["D"]
    a32 = np.ascontiguousarray(a32, dtype=np.float32)
    dL_dmean2D = np.zeros((P, 3), np.float32); dL_dmean2D[:, :2] = a32[:, 0:2]
    dL_dconic = np.zeros((P, 4), np.float32); dL_dconic[:, 0] = a32[:, 2]; dL_dconic[:, 1] = a32[:, 3]; dL_dconic[:, 3] = a32[:, 4]
    dL_dopacity = np.ascontiguousarray(a32[:, 5:6])
    dL_dcolor = np.ascontiguousarray(a32[:, 6:9])
    dL_ddepth = np.ascontiguousarray(a32[:, 9])
    out = dict(dL_dmeans2D=dL_dmean2D, dL_dconic=dL_dconic, dL_dopacity=dL_dopacity,
               dL_dcolors=dL_dcolor, dL_ddepths=dL_ddepth)
    out["dL_dmeans3D"] = np.zeros((P, 3), np.float32)
    out["dL_dcov3D"] = np.zeros((P, 6), np.float32)
    out["dL_dsh"] = np.zeros((P, M, 3), np.float32)
    out["dL_dscales"] = np.zeros((P, 3), np.float32)
    out["dL_drotations"] = np.zeros((P, 4), np.float32)
    cov3D = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else st["cov3D"]
    L.orc_preprocess_bwd(P, int(D), int(M), _p(inp["means3D"]), _p(st["radii"]), _p(inp["shs"]),
                         _p(st["clamped"]), _p(inp["scales"]), _p(inp["rotations"]),
                         ctypes.c_float(inp["scale_modifier"]), _p(cov3D), _p(inp["view"]), _p(inp["proj"]),
                         int(W), int(H), ctypes.c_float(inp["tanfovx"]), ctypes.c_float(inp["tanfovy"]),
                         _p(inp["cam"]), _p(dL_dmean2D), _p(dL_dconic), _p(out["dL_dmeans3D"]), _p(dL_dcolor),
                         _p(dL_ddepth), _p(out["dL_dcov3D"]), _p(out["dL_dsh"]), _p(out["dL_dscales"]),
                         _p(out["dL_drotations"]))
    return out
