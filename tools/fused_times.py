#!/usr/bin/env python
"""C3 step time from RAW attributes: torch activations + cat + standard operator vs FusedGaussianRasterizer (row f1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer, runtime, scenes  # noqa: E402
from gaustudio_amd.fused import FusedGaussianRasterizer  # noqa: E402

P, W, H, D = 1_000_000, 1920, 1080, 3
dev = torch.device("cuda:0")
runtime.warm_start(dev)
cam = scenes.make_camera(W, H)
sc = scenes.make_scene(P, cam, seed=0)
raw = dict(xyz=sc.means3D, f_dc=sc.shs[:, :1].contiguous(), f_rest=sc.shs[:, 1:].contiguous(),
           opacity=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)), scale=torch.log(sc.scales), rot=sc.rotations * 1.3)
raw = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}
grads = [g.to(dev) for g in scenes.make_output_grads(cam, seed=1)]
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, cam.viewmatrix.to(dev),
                                   cam.projmatrix.to(dev), D, cam.campos.to(dev), False, False)
m2 = torch.zeros_like(raw["xyz"], requires_grad=True)


def unfused():
    out = GaussianRasterizer(rs)(means3D=raw["xyz"], means2D=m2, opacities=torch.sigmoid(raw["opacity"]),
                                 shs=torch.cat((raw["f_dc"], raw["f_rest"]), dim=1), scales=torch.exp(raw["scale"]),
                                 rotations=torch.nn.functional.normalize(raw["rot"]))
    torch.autograd.backward([out[0], out[2], out[3], out[4]], grads)


def fused():
    out = FusedGaussianRasterizer(rs)(means3D=raw["xyz"], means2D=m2, raw_opacities=raw["opacity"], f_dc=raw["f_dc"],
                                      f_rest=raw["f_rest"], raw_scales=raw["scale"], raw_rotations=raw["rot"])
    torch.autograd.backward([out[0], out[2], out[3], out[4]], grads)


for name, fn in (("unfused", unfused), ("fused", fused)):
    for _ in range(5):
        for p in raw.values():
            p.grad = None
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        for p in raw.values():
            p.grad = None
        fn()
    torch.cuda.synchronize()
    print(f"{name} {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms/step")
