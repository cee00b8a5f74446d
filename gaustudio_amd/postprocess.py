"""Post-render epilogue on the GPU: what gs-extract-mesh / gs-extract-pcd do right after
the operator (gaustudio/scripts/extract_mesh.py:101-110, extract_pcd.py:325-329), as two HIP streaming kernels
instead of ~15 torch ops with [H,W,3] intermediates.

Mirrors gaustudio.datasets.Camera.depth2point / depth2normal (datasets/__init__.py:307-380): same arguments, with
the camera given by its `intrinsics` [3,3] and `extrinsics` [4,4] (world-to-camera) tensors.
"""
import ctypes

import torch

from . import _C


def _host16(t, n):
    if t is None:
        return None
    a = (ctypes.c_float * n)(*[float(v) for v in t.detach().cpu().reshape(-1).tolist()])
    return a


def _check(depth):
    if depth.dim() != 2:
        raise ValueError("depth must have shape [H, W]")
    if not depth.is_cuda:
        raise RuntimeError(f"depth is on '{depth.device}': gaustudio_amd runs on ROCm devices only (no CPU fallback)")
    if depth.dtype != torch.float32:
        raise RuntimeError("depth must be float32")
    return depth.contiguous()


def depth_to_points(depth, intrinsics, extrinsics=None, coordinate="camera"):
    """Camera.depth2point: [H,W] depth -> [H,W,3] points in 'camera' or 'world' coordinates."""
    if coordinate not in ("camera", "world"):
        raise ValueError("Invalid coordinate system.")
    depth = _check(depth)
    H, W = depth.shape
    L = _C.lib()
    out = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device)
    K = _host16(intrinsics, 9)
    E = _host16(extrinsics, 16) if coordinate == "world" else None
    if coordinate == "world" and E is None:
        raise ValueError("extrinsics are required for world coordinates")
    with torch.cuda.device(depth.device):
        rc = L.gsr_depth_to_points(_C._ptr(depth), ctypes.c_int(W), ctypes.c_int(H), K, E, _C._ptr(out), _C._stream(depth.device))
    if rc < 0:
        raise RuntimeError(f"gsr_depth_to_points failed (rc={rc}): singular intrinsics / extrinsics?")
    return out


def depth_to_normals(depth, intrinsics, extrinsics=None, k=3, d_min=1e-3, d_max=100000.0, coordinate="camera"):
    """Camera.depth2normal: [H,W] depth -> [H,W,3] normals, (-1,-1,-1) where invalid."""
    if coordinate not in ("camera", "world"):
        raise ValueError("Invalid coordinate system.")
    depth = _check(depth)
    H, W = depth.shape
    L = _C.lib()
    out = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device)
    K = _host16(intrinsics, 9)
    E = _host16(extrinsics, 16) if coordinate == "world" else None
    if coordinate == "world" and E is None:
        raise ValueError("extrinsics are required for world coordinates")
    with torch.cuda.device(depth.device):
        rc = L.gsr_depth_to_normals(_C._ptr(depth), ctypes.c_int(W), ctypes.c_int(H), K, ctypes.c_int(int(k)),
                                    ctypes.c_float(d_min), ctypes.c_float(d_max), E, _C._ptr(out), _C._stream(depth.device))
    if rc < 0:
        raise RuntimeError(f"gsr_depth_to_normals failed (rc={rc})")
    return out


def depth_epilogue(depth, intrinsics, extrinsics=None, opacity=None, min_opacity=0.5, k=3, d_min=1e-3, d_max=100000.0,
                   coordinate="world", want_points=True, want_normals=True):
    """The step between the render and the fusion of gs-extract-mesh in one kernel (extract_mesh.py:101-110 +
    depth2normal): `depth[opacity < min_opacity] = 0`, `depth2point(depth, coordinate)` and `depth2normal(depth, k, d_min,
    d_max, coordinate)` -- returns (points [H,W,3] or None, normals [H,W,3] or None), value for value what the three
    separate steps give."""
    if coordinate not in ("camera", "world"):
        raise ValueError("Invalid coordinate system.")
    depth = _check(depth)
    H, W = depth.shape
    if opacity is not None:
        opacity = _check(opacity.reshape(H, W))
    K = _host16(intrinsics, 9)
    E = _host16(extrinsics, 16) if coordinate == "world" else None
    if coordinate == "world" and E is None:
        raise ValueError("extrinsics are required for world coordinates")
    pts = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device) if want_points else None
    nrm = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device) if want_normals else None
    L = _C.lib()
    with torch.cuda.device(depth.device):
        rc = L.gsr_depth_epilogue(_C._ptr(depth), _C._ptr(opacity), ctypes.c_float(float(min_opacity)), ctypes.c_int(W),
                                  ctypes.c_int(H), K, ctypes.c_int(int(k)), ctypes.c_float(d_min), ctypes.c_float(d_max), E,
                                  _C._ptr(pts), _C._ptr(nrm), _C._stream(depth.device))
    if rc < 0:
        raise RuntimeError(f"gsr_depth_epilogue failed (rc={rc}): k > 5, or singular intrinsics / extrinsics?")
    return pts, nrm


def masked_bilateral_filter(depth_map, mask, d=3, sigma_color=75, sigma_space=75):
    """gaustudio/scripts/extract_pcd.py:185-238 masked_bilateral_filter, on the GPU (the reference goes through numpy and
    cv2 on the CPU): returns (filtered_depth, new_mask), new_mask in `mask`'s dtype.  cv2 is restated from its published
    float32 algorithm (parity unpinned: the library is not in this image)."""
    depth = _check(depth_map)
    H, W = depth.shape
    if mask.shape != depth.shape:
        raise ValueError("mask must have the depth map's shape")
    m8 = (mask != 0).to(device=depth.device, dtype=torch.uint8).contiguous()
    out = torch.empty_like(depth)
    new_mask = torch.empty_like(m8)
    scratch = torch.empty(2, dtype=torch.int32, device=depth.device)
    L = _C.lib()
    with torch.cuda.device(depth.device):
        rc = L.gsr_masked_bilateral(_C._ptr(depth), _C._ptr(m8), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(int(d)),
                                    ctypes.c_float(float(sigma_color)), ctypes.c_float(float(sigma_space)), _C._ptr(out),
                                    _C._ptr(new_mask), _C._ptr(scratch), _C._stream(depth.device))
    if rc < 0:
        raise RuntimeError(f"gsr_masked_bilateral failed (rc={rc}): d must be odd")
    return out, new_mask.to(mask.dtype)
