#!/usr/bin/env python
"""Row f2 measurement: the two post-render HIP kernels against the HBM roofline, next to the same maths written as
the torch op sequence Camera.depth2point / depth2normal issue on the GPU (datasets/__init__.py:106-112,307-380)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.getcwd())
from gaustudio_amd import postprocess as pp, scenes  # noqa: E402

dev = torch.device("cuda:0")


def torch_depth2point(depth, K, c2w=None):
    H, W = depth.shape
    vx = torch.arange(W, dtype=torch.float32, device=depth.device) / (W - 1)
    vy = torch.arange(H, dtype=torch.float32, device=depth.device) / (H - 1)
    vy, vx = torch.meshgrid(vy, vx, indexing="ij")
    ndc = torch.stack([vx, vy, depth], dim=-1)
    cz = ndc[..., 2:3]
    cxy = ndc[..., :2] * torch.tensor([[W - 1, H - 1]], device=depth.device) * cz
    cam = torch.cat([cxy, cz], dim=-1) @ torch.inverse(K.t())
    if c2w is None:
        return cam
    cam = cam.reshape(-1, 3)
    w = torch.cat([cam, torch.ones_like(cam[..., 0:1])], dim=-1) @ c2w.transpose(0, 1)
    return w[..., :3].reshape(H, W, 3)


def torch_depth2normal(depth, K, k=3, d_min=1e-3, d_max=1e5):
    pts = torch_depth2point(depth, K)[None].permute(0, 3, 1, 2)
    k = (k - 1) // 2
    _, _, H, W = pts.shape
    pad = F.pad(pts, (k, k, k, k), value=0)
    val = ((pad[:, 2:] > d_min) & (pad[:, 2:] < d_max)).float()
    vert = pad[:, :, :H, k:k + W] - pad[:, :, 2 * k:2 * k + H, k:k + W]
    hori = pad[:, :, k:k + H, :W] - pad[:, :, k:k + H, 2 * k:2 * k + W]
    ok = (val[:, :, k:k + H, k:k + W] * val[:, :, :H, k:k + W] * val[:, :, 2 * k:2 * k + H, k:k + W]
          * val[:, :, k:k + H, :W] * val[:, :, k:k + H, 2 * k:2 * k + W]) > 0.5
    n = F.normalize(-torch.linalg.cross(vert, hori, dim=1), p=2.0, dim=1, eps=1e-12)
    n[~ok.repeat(1, 3, 1, 1)] = -1
    return n.squeeze(0).permute(1, 2, 0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for W, H in ((1920, 1080), (3840, 2160)):
    cam = scenes.make_camera(W, H)
    f = W / (2 * cam.tanfovx)
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float32, device=dev)
    E = cam.viewmatrix.t().contiguous().to(dev)
    c2w = torch.inverse(E)
    g = torch.Generator().manual_seed(0)
    depth = (3 + torch.rand(H, W, generator=g)).to(dev)
    nbytes = H * W * 16
    Kh, Eh = K.cpu(), E.cpu()      # Camera.intrinsics / .extrinsics are CPU tensors in the reference (datasets/__init__.py:226-237)
    for name, ours, ref in (("depth_to_points(world)", lambda: pp.depth_to_points(depth, Kh, Eh, "world"),
                             lambda: torch_depth2point(depth, K, c2w)),
                            ("depth_to_normals(camera)", lambda: pp.depth_to_normals(depth, Kh),
                             lambda: torch_depth2normal(depth, K))):
        a, b = timeit(ours), timeit(ref)
        print(f"{W}x{H} {name}: HIP {a * 1e3:.1f} us = {nbytes / a / 1e6:.0f} GB/s algorithmic ({nbytes / a / 1e6 / 8000 * 100:.1f} % of 8 TB/s)"
              f" | torch op sequence {b * 1e3:.1f} us | {b / a:.1f}x")
