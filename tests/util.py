"""Shared helpers of the parity tests: run the same seeded scene through the HIP path (through
gaustudio_amd._C -> the C ABI of libgsrast.so) and through the CPU oracle."""
import numpy as np
import torch

from gaustudio_amd import scenes


def scene_kwargs(sc, use_sh=True, use_cov=False):
    """The optional-input combinations of GaussianRasterizer.forward."""
    kw = {}
    if use_sh:
        kw["shs"] = sc.shs
    else:
        kw["colors_precomp"] = torch.sigmoid(sc.shs[:, 0, :]).contiguous()
    if use_cov:
        from oracle import pyoracle as po  # noqa: F401  (cov from scale/rot via torch below)
        r, x, y, z = sc.rotations.unbind(1)
        R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                         torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                         torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
        S = R @ torch.diag_embed(sc.scales ** 2) @ R.transpose(1, 2)
        kw["cov3D_precomp"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()
    else:
        kw["scales"] = sc.scales
        kw["rotations"] = sc.rotations
    return kw


def oracle_forward(po, sc, cam, D, kw, scale_modifier=1.0, bg=None, prefiltered=False, tight=True):
    """tight=True: the oracle bins into the product's tight rects (oracle/gsr_oracle.c:orc_rects), so that lists,
    ranges and n_contrib are comparable bit for bit; tests/test_oracle_golden.py proves on the CPU that this changes
    no image bit with respect to tight=False, the reference's own binning."""
    npk = {k: v.numpy() for k, v in kw.items()}
    return po.forward(sc.means3D.numpy(), sc.opacities.numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                      cam.campos.numpy(), cam.width, cam.height, cam.tanfovx, cam.tanfovy, sh_degree=D,
                      scale_modifier=scale_modifier, bg=None if bg is None else bg.numpy(),
                      prefiltered=prefiltered, tight=tight, **npk)


def hip_forward(sc, cam, D, kw, scale_modifier=1.0, bg=None, prefiltered=False, debug=False, device="cuda"):
    """Calls _C.rasterize_gaussians exactly as _RasterizeGaussians.forward does and decodes the opaque
    buffers through the gsr_inspect_* entry points."""
    from gaustudio_amd import _C
    e = torch.Tensor([])
    g = lambda k: kw[k].to(device) if k in kw else e
    bg = torch.zeros(3) if bg is None else bg
    out = _C.rasterize_gaussians(bg, sc.means3D.to(device), g("colors_precomp"), sc.opacities.to(device), g("scales"),
                                 g("rotations"), scale_modifier, g("cov3D_precomp"), cam.viewmatrix.to(device),
                                 cam.projmatrix.to(device), cam.tanfovx, cam.tanfovy, cam.height, cam.width,
                                 g("shs"), D, cam.campos.to(device), prefiltered, debug)
    R, color, depth, median, opacity, radii, geom, binning, img = out
    st = dict(num_rendered=R, color=color, depth=depth, median=median, opacity=opacity, radii=radii,
              geom=geom, binning=binning, img=img)
    st.update(_C.inspect_geometry(geom, radii) if radii.numel() else {})
    if radii.numel():
        cnt = _C.inspect_counts(img, cam.width, cam.height)
        assert cnt["num_rendered"] == R
        st["num_binned"] = cnt["num_binned"]
        st["point_list"], st["ranges"] = _C.inspect_binning(binning, img, cnt["num_binned"], cam.width, cam.height)
        st["final_T"], st["n_contrib"] = _C.inspect_image(img, cam.width, cam.height)
    return st


def ab_variants():
    """Were the superseded per-wave compositing kernels (fwd_variant 1, bwd_variant bit 1) compiled into libgsrast.so?  The shipped
    library is built without them (csrc/Makefile AB=1 adds them); the tests run their A/B legs only where they exist."""
    from gaustudio_amd import _C
    return _C.get_option("ab_variants") == 1


def to_np(t):
    return t.detach().cpu().numpy()


def assert_bits_equal(a, b, name):
    """Bit-exact float comparison (+0 == -0 is allowed, NaN never appears)."""
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    bad = a != b
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError(f"{name}: {int(bad.sum())} of {a.size} elements differ; first at {tuple(i)}: "
                             f"{a[tuple(i)]!r} vs {b[tuple(i)]!r}; max abs diff {np.abs(a.astype(np.float64) - b.astype(np.float64)).max()}")


def compare_forward_exact(hs, os_, vis_only=True):
    """HIP state vs oracle state: every output and every intermediate must match to the bit."""
    radii = to_np(hs["radii"])
    assert_bits_equal(radii, os_["radii"], "radii")
    assert hs["num_rendered"] == os_["num_rendered"], (hs["num_rendered"], os_["num_rendered"])   # reference-defined
    assert hs["num_binned"] == os_["num_binned"], (hs["num_binned"], os_["num_binned"])
    vis = radii > 0
    for k in ("means2D", "depths", "conic_opacity", "rgb", "clamped", "tiles_touched"):
        a = to_np(hs[k]); b = os_[k]
        if k == "tiles_touched":
            a = a.astype(np.uint32)
        if k == "clamped" and os_["_inputs"]["colors_precomp"] is not None:
            continue
        if k == "rgb" and os_["_inputs"]["colors_precomp"] is not None:
            b = os_["_inputs"]["colors_precomp"]
        assert_bits_equal(a[vis], b[vis], k)
    assert_bits_equal(to_np(hs["ranges"]).astype(np.uint32)[_nonempty(os_)], os_["ranges"][_nonempty(os_)], "ranges")
    assert_bits_equal(to_np(hs["point_list"]).astype(np.uint32), os_["point_list"], "point_list")
    assert_bits_equal(to_np(hs["n_contrib"]).astype(np.uint32), os_["n_contrib"], "n_contrib")
    assert_bits_equal(to_np(hs["final_T"]), os_["final_T"], "final_T")
    for k in ("color", "depth", "median", "opacity"):
        assert_bits_equal(to_np(hs[k]), os_[k], k)


def _nonempty(os_):
    r = os_["ranges"]
    return r[:, 1] > r[:, 0]


class GsrOptions(__import__("ctypes").Structure):
    """include/gsrast.h gsr_options (ABI v6)."""
    _fields_ = [(n, __import__("ctypes").c_int32) for n in ("struct_bytes", "tight_binning", "cull", "fwd_variant", "bwd_variant",
                                                           "speculative", "tile_row_lo", "tile_row_hi", "fast_exp", "forward_only")]


def make_options(L, struct_bytes=None, **fields):
    import ctypes
    o = GsrOptions()
    L.gsr_options_init(ctypes.byref(o))
    assert o.struct_bytes == ctypes.sizeof(GsrOptions) and all(getattr(o, n) == -1 for n, _ in GsrOptions._fields_[1:])
    for k, v in fields.items():
        setattr(o, k, int(v))
    if struct_bytes is not None:
        o.struct_bytes = int(struct_bytes)      # an older caller's (shorter) struct: the fields beyond it count as -1
    return o


class _BackwardOut(dict):
    """hip_backward_raw's result: the gradient tensors by name (+ "acc"); `.scratch` = the backward's scratch buffer, for `reuse`."""


def hip_backward_raw(hs, sc, cam, D, kw, grads, scale_modifier=1.0, bg=None, debug=False, options=None, no_dcov=False, parts=3,
                     sh_g0=0, sh_g1=None, reuse=None):
    """gsr_backward (or, with `options` = a GsrOptions / None-able dict, gsr_backward_ex) called straight through ctypes
    with a test-owned scratch buffer, so that the composite-stage accumulator rows (scratch[P,12]) can be inspected next
    to the 8 outputs.  `reuse`: the dict a previous call returned -- its output tensors and its scratch are used again (the
    staged / banded forms of one backward: gsr_backward_ex `parts`, `sh_g0`, `sh_g1`)."""
    import ctypes
    from gaustudio_amd import _C
    L = _C.lib()
    dev = hs["radii"].device
    P = sc.means3D.shape[0]
    M = kw["shs"].shape[1] if "shs" in kw else 0
    e = torch.Tensor([])
    g = lambda k: kw[k].to(dev).contiguous() if k in kw else e
    bg = torch.zeros(3) if bg is None else bg
    gc, gd, gm, go = [None if t is None else t.to(dev).contiguous() for t in grads]      # None: NULL = absent = zero (include/gsrast.h)
    fo = dict(dtype=torch.float32, device=dev)
    if reuse is not None:
        out = {k: reuse[k] for k in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")}
        scratch = reuse.scratch
    else:
        out = dict(dL_dmeans2D=torch.full((P, 3), float("nan"), **fo), dL_dopacity=torch.full((P, 1), float("nan"), **fo),
                   dL_dcolors=torch.full((P, 3), float("nan"), **fo), dL_dmeans3D=torch.full((P, 3), float("nan"), **fo),
                   dL_dcov3D=torch.full((P, 6), float("nan"), **fo), dL_dsh=torch.full((P, M, 3), float("nan"), **fo),
                   dL_dscales=torch.full((P, 3), float("nan"), **fo), dL_drotations=torch.full((P, 4), float("nan"), **fo))
        nscratch = L.gsr_backward_scratch_bytes(ctypes.c_int(P), ctypes.c_int(hs["num_rendered"]))
        scratch = torch.full((nscratch,), 0xAB, dtype=torch.uint8, device=dev)   # poison: the library must zero it
    means = sc.means3D.to(dev); shs = g("shs"); col = g("colors_precomp"); scl = g("scales"); rot = g("rotations")
    cov = g("cov3D_precomp")
    dcov = None if no_dcov else out["dL_dcov3D"]     # NULL is allowed when cov3D_precomp is NULL (include/gsrast.h)
    view = cam.viewmatrix.to(dev); proj = cam.projmatrix.to(dev); cpos = cam.campos.to(dev)
    p = _C._ptr
    if options is not None:
        opt = options if isinstance(options, GsrOptions) else make_options(L, **options)
        rc = L.gsr_backward_ex(ctypes.byref(opt), ctypes.c_int(parts), ctypes.c_int(int(sh_g0)), ctypes.c_int(P if sh_g1 is None else int(sh_g1)), ctypes.c_int(P), ctypes.c_int(D),
                               ctypes.c_int(M), ctypes.c_int(hs["num_rendered"]), p(bg), ctypes.c_int(cam.width), ctypes.c_int(cam.height),
                               p(means), p(shs), p(None), p(col), p(scl), ctypes.c_float(scale_modifier), p(rot), p(cov), ctypes.c_int(0),
                               ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy), p(hs["radii"]), p(hs["geom"]), p(hs["binning"]),
                               p(hs["img"]), p(gc), p(gd), p(gm), p(go), p(out["dL_dmeans2D"]), p(out["dL_dopacity"]), p(out["dL_dcolors"]),
                               p(out["dL_dmeans3D"]), p(dcov), p(out["dL_dsh"]), p(None), p(out["dL_dscales"]),
                               p(out["dL_drotations"]), p(scratch), ctypes.c_int(bool(debug)), _C._stream(dev))
        if rc < 0:
            raise _C._err(L, rc)
        rc = 0
    else:
        rc = L.gsr_backward(ctypes.c_int(P), ctypes.c_int(D), ctypes.c_int(M), ctypes.c_int(hs["num_rendered"]), p(bg),
                        ctypes.c_int(cam.width), ctypes.c_int(cam.height), p(means), p(shs), p(col), p(scl),
                        ctypes.c_float(scale_modifier), p(rot), p(cov), p(view), p(proj), p(cpos),
                        ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy), p(hs["radii"]), p(hs["geom"]),
                        p(hs["binning"]), p(hs["img"]), p(gc), p(gd), p(gm), p(go), p(out["dL_dmeans2D"]),
                        p(out["dL_dopacity"]), p(out["dL_dcolors"]), p(out["dL_dmeans3D"]), p(dcov),
                        p(out["dL_dsh"]), p(out["dL_dscales"]), p(out["dL_drotations"]), p(scratch),
                        ctypes.c_int(bool(debug)), _C._stream(dev))
    if rc < 0:
        raise _C._err(L, rc)
    acc = torch.empty((P, 10), **fo)
    rc = L.gsr_inspect_backward_sums(p(hs["geom"]), p(scratch), ctypes.c_int(P), ctypes.c_int(hs["num_rendered"]),
                                     p(hs["radii"]), p(acc), _C._stream(dev))
    if rc < 0:
        raise _C._err(L, rc)
    torch.cuda.synchronize()
    out["acc"] = acc
    out = _BackwardOut(out)
    out.scratch = scratch          # (an attribute, not a key: callers iterate over the gradient tensors)
    return out
