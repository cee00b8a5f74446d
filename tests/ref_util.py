"""ctypes driver of oracle/_ref/libgsref.so: the REFERENCE's own kernels (hipified test-only by
oracle/build_ref.sh) running on the GPU.  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import torch

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
# two builds of the same reference sources (oracle/build_ref.sh): hipcc's default contraction, and -ffp-contract=off
_FILES = {"default": "libgsref.so", "nocontract": "libgsref_nocontract.so"}
_PATH = os.path.join(_DIR, _FILES["default"])
_L = {}


def available(variant="default"):
    return os.path.exists(os.path.join(_DIR, _FILES[variant])) and torch.cuda.is_available()


def lib(variant="default"):
    if variant not in _L:
        L = ctypes.CDLL(os.path.join(_DIR, _FILES[variant]))
        L.ref_create.restype = ctypes.c_void_p
        L.ref_destroy.argtypes = [ctypes.c_void_p]
        _L[variant] = L
    return _L[variant]


def _p(t):
    if t is None or t.numel() == 0:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def run(sc, cam, D, kw, grads=None, scale_modifier=1.0, bg=None, dev="cuda", variant="default"):
    """Forward (+ backward when grads is given) of the reference rasterizer.  Returns a dict of CPU tensors."""
    L = lib(variant)
    h = ctypes.c_void_p(L.ref_create())
    try:
        P = sc.means3D.shape[0]
        W, H = cam.width, cam.height
        g = lambda k: kw[k].to(dev).contiguous() if k in kw else None
        means = sc.means3D.to(dev).contiguous(); opac = sc.opacities.to(dev).contiguous()
        shs, col, scl, rot, cov = g("shs"), g("colors_precomp"), g("scales"), g("rotations"), g("cov3D_precomp")
        M = shs.shape[1] if shs is not None else 0
        view = cam.viewmatrix.to(dev).contiguous(); proj = cam.projmatrix.to(dev).contiguous()
        cpos = cam.campos.to(dev).contiguous()
        bgd = (torch.zeros(3) if bg is None else bg).to(dev).contiguous()
        fo = dict(dtype=torch.float32, device=dev)
        color = torch.empty(3, H, W, **fo); depth = torch.empty(1, H, W, **fo)
        median = torch.empty(3, H, W, **fo); opacity = torch.empty(1, H, W, **fo)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        R = L.ref_forward(h, P, D, M, _p(bgd), W, H, _p(means), _p(shs), _p(col), _p(opac), _p(scl),
                          ctypes.c_float(scale_modifier), _p(rot), _p(cov), _p(view), _p(proj), _p(cpos),
                          ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy), 0, _p(color), _p(depth),
                          _p(median), _p(opacity), _p(radii))
        if R < 0:
            raise RuntimeError("reference forward failed")
        out = dict(num_rendered=R, color=color.cpu(), depth=depth.cpu(), median=median.cpu(), opacity=opacity.cpu(),
                   radii=radii.cpu())
        if grads is not None:
            gc, gd, gm, go = [t.to(dev).contiguous() for t in grads]
            z = lambda *s: torch.zeros(*s, **fo)
            G = dict(dL_dmeans2D=z(P, 3), dL_dconic=z(P, 4), dL_dopacity=z(P, 1), dL_dcolors=z(P, 3), dL_ddepths=z(P),
                     dL_dmeans3D=z(P, 3), dL_dcov3D=z(P, 6), dL_dsh=z(P, max(M, 1), 3), dL_dscales=z(P, 3),
                     dL_drotations=z(P, 4))
            torch.cuda.synchronize()
            rc = L.ref_backward(h, P, D, M, _p(bgd), W, H, _p(means), _p(shs), _p(col), _p(scl),
                                ctypes.c_float(scale_modifier), _p(rot), _p(cov), _p(view), _p(proj), _p(cpos),
                                ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy), _p(radii), _p(gc), _p(gd),
                                _p(gm), _p(go), _p(G["dL_dmeans2D"]), _p(G["dL_dconic"]), _p(G["dL_dopacity"]),
                                _p(G["dL_dcolors"]), _p(G["dL_ddepths"]), _p(G["dL_dmeans3D"]), _p(G["dL_dcov3D"]),
                                _p(G["dL_dsh"]), _p(G["dL_dscales"]), _p(G["dL_drotations"]))
            if rc != 0:
                raise RuntimeError("reference backward failed")
            if M == 0:
                G["dL_dsh"] = torch.zeros(P, 0, 3)
            out.update({k: v.cpu() for k, v in G.items()})
        return out
    finally:
        L.ref_destroy(h)
