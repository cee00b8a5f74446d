#!/usr/bin/env python
"""GPU A/B (optional): the REFERENCE's own kernels (hipified test-only, oracle/_ref/libgsref.so)
timed on the same C3 inputs as bench.py, next to this repository's operator.  Test infrastructure: not part of the
product, not used by bench.py."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ref_util  # noqa: E402
from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer, runtime, scenes  # noqa: E402

P, W, H, D = 1_000_000, 1920, 1080, 3
dev = "cuda"
runtime.warm_start(torch.device("cuda:0"))
cam = scenes.make_camera(W, H)
sc = scenes.make_scene(P, cam, seed=0)
grads = [g.to(dev) for g in scenes.make_output_grads(cam, seed=1)]
L = ref_util.lib()
h = ctypes.c_void_p(L.ref_create())
p = ref_util._p
means, opac, shs = sc.means3D.to(dev), sc.opacities.to(dev), sc.shs.to(dev).contiguous()
scl, rot = sc.scales.to(dev), sc.rotations.to(dev)
view, proj, cpos = cam.viewmatrix.to(dev).contiguous(), cam.projmatrix.to(dev).contiguous(), cam.campos.to(dev)
bgd = torch.zeros(3, device=dev)
fo = dict(dtype=torch.float32, device=dev)
cf = ctypes.c_float


def ref_step():
    # what rasterize_points.cu:68-81,160-169 does around the kernels: filled outputs (the shim memsets them),
    # zeroed gradient tensors
    color = torch.empty((3, H, W), **fo); depth = torch.empty((1, H, W), **fo)
    median = torch.empty((3, H, W), **fo); opacity = torch.empty((1, H, W), **fo)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    R = L.ref_forward(h, P, D, 16, p(bgd), W, H, p(means), p(shs), p(None), p(opac), p(scl), cf(1.0), p(rot), p(None),
                      p(view), p(proj), p(cpos), cf(cam.tanfovx), cf(cam.tanfovy), 0, p(color), p(depth), p(median),
                      p(opacity), p(radii))
    assert R > 0
    z = lambda *s: torch.zeros(*s, **fo)
    G = [z(P, 3), z(P, 4), z(P, 1), z(P, 3), z(P), z(P, 3), z(P, 6), z(P, 16, 3), z(P, 3), z(P, 4)]
    rc = L.ref_backward(h, P, D, 16, p(bgd), W, H, p(means), p(shs), p(None), p(scl), cf(1.0), p(rot), p(None), p(view),
                        p(proj), p(cpos), cf(cam.tanfovx), cf(cam.tanfovy), p(radii), *[p(g) for g in grads],
                        *[p(g) for g in G])
    assert rc == 0
    return R


params = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
m2 = torch.zeros_like(params["means3D"], requires_grad=True)
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bgd, 1.0, view, proj, D, cpos, False, False)
rast = GaussianRasterizer(rs)


def our_step():
    for q in params.values():
        q.grad = None
    c, _, d, m, o = rast(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                         scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([c, d, m, o], grads)


for name, fn, n in (("reference kernels (hipified, test-only)", ref_step, 10), ("gaustudio_amd", our_step, 30)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: {ms:.3f} ms/step fwd+bwd at C3 = {W * H / ms / 1e3:.1f} Mpixels/s")
L.ref_destroy(h)
