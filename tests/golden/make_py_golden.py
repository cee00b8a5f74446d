#!/usr/bin/env python
"""Generates tests/golden/py_sh_cov.npz by IMPORTING the reference's own Python helpers in this
container (they cannot travel to the GPU box, /root/reference does not exist there):

  /root/reference/gaustudio/utils/sh_utils.py:57-112   eval_sh   (SH -> RGB, same constants as auxiliary.h:22-38)
  /root/reference/gaustudio/models/utils.py:44-97      build_covariance_from_scaling_rotation

These are the only pieces of the reference that restate hot-path math outside the CUDA kernels
(SURVEY.md s4); the fixture pins the oracle's SH evaluation and cov3D construction against them.
models/utils.py hard-codes device='cuda' in torch.zeros; it is executed here with a torch proxy whose
zeros() drops the device argument (the source file itself is read from the reference, not copied)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/gaustudio"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gaustudio_amd import scenes  # noqa: E402


def load_sh_utils():
    spec = importlib.util.spec_from_file_location("ref_sh_utils", os.path.join(REF, "utils", "sh_utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_model_utils():
    class TorchProxy(types.ModuleType):
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def zeros(*a, device=None, **kw):
            return torch.zeros(*a, **kw)

    src = open(os.path.join(REF, "models", "utils.py")).read()
    ns = {"__name__": "ref_model_utils"}
    proxy = TorchProxy("torch")
    real_import = __import__

    def fake_import(name, *a, **kw):
        if name == "torch":
            return proxy
        return real_import(name, *a, **kw)

    ns["__builtins__"] = dict(vars(__import__("builtins")), __import__=fake_import)
    exec(compile(src, "models/utils.py", "exec"), ns)
    return ns


def main():
    shu = load_sh_utils()
    mu = load_model_utils()
    cam = scenes.make_camera(640, 480)
    sc = scenes.make_scene(4096, cam, seed=11)
    campos = torch.tensor([0.3, -0.2, 0.1])
    dirs = sc.means3D - campos[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = dict(means3D=sc.means3D.numpy(), shs=sc.shs.numpy(), campos=campos.numpy(), scales=sc.scales.numpy(),
               rotations=sc.rotations.numpy())
    shs_view = sc.shs.transpose(1, 2)                         # [P,3,16] as vanilla_renderer.py:45 builds it
    for deg in range(4):
        rgb = shu.eval_sh(deg, shs_view.double(), dirs.double())
        out[f"rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()      # vanilla_renderer.py:49
    for mod in (1.0, 1.7):
        cov = mu["build_covariance_from_scaling_rotation"](sc.scales.double(), mod, sc.rotations.double())
        out[f"cov3D_mod{mod}"] = cov.numpy()
    np.savez_compressed(os.path.join(HERE, "py_sh_cov.npz"), **out)
    print("wrote", os.path.join(HERE, "py_sh_cov.npz"), os.path.getsize(os.path.join(HERE, "py_sh_cov.npz")) // 1024, "KiB")


def make_post_golden():
    """tests/golden/py_post.npz: Camera.depth2point / depth2normal of the reference
    (/root/reference/gaustudio/datasets/__init__.py:307-380) on a synthetic depth map with holes."""
    import sys
    spec = importlib.util.spec_from_file_location("ref_datasets", os.path.join(REF, "datasets", "__init__.py"),
                                                  submodule_search_locations=[])
    m = importlib.util.module_from_spec(spec)
    sys.modules["ref_datasets"] = m
    try:
        spec.loader.exec_module(m)          # Camera is defined before the dataset loaders are imported (line 418)
    except ModuleNotFoundError:
        pass
    Camera = m.Camera
    g = torch.Generator().manual_seed(5)
    W, H = 96, 64
    a = 0.3
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    cam = Camera(R=R, T=np.array([0.2, -0.1, 1.5]), FoVx=math.radians(60), FoVy=math.radians(42), image_width=W,
                 image_height=H)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    depth = 3.0 + 0.5 * torch.sin(xx / 9.0) + 0.3 * torch.cos(yy / 7.0) + 0.02 * torch.randn(H, W, generator=g)
    depth[torch.rand(H, W, generator=g) < 0.05] = 0.0          # holes (masked pixels are zeroed in extract_mesh.py:107)
    out = dict(depth=depth.numpy(), intrinsics=cam.intrinsics.numpy(), extrinsics=cam.extrinsics.numpy())
    out["points_camera"] = cam.depth2point(depth, coordinate="camera").numpy()
    out["points_world"] = cam.depth2point(depth, coordinate="world").numpy()
    out["normals_camera"] = cam.depth2normal(depth, coordinate="camera").numpy()
    out["normals_world"] = cam.depth2normal(depth, coordinate="world").numpy()
    out["normals_camera_k5"] = cam.depth2normal(depth, k=5, coordinate="camera").numpy()
    path = os.path.join(HERE, "py_post.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_camera_golden():
    """tests/golden/py_cameras.npz + py_cameras.json: cameras.json entries pushed through the reference's
    JSON_to_camera (utils/cameras_utils.py:8-38) -> Camera (datasets/__init__.py:113-183) -> camera_to_JSON
    (datasets/utils.py:58-80).  plyfile (imported at the top of datasets/utils.py, absent here) is stubbed;
    the camera functions do not touch it."""
    import json
    import sys
    spec = importlib.util.spec_from_file_location("ref_datasets", os.path.join(REF, "datasets", "__init__.py"),
                                                  submodule_search_locations=[])
    ds = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(ds)
    except ModuleNotFoundError:
        pass
    pkg = types.ModuleType("gaustudio")
    pkg.datasets = ds
    ply = types.ModuleType("plyfile")
    ply.PlyData = ply.PlyElement = None
    saved = {k: sys.modules.get(k) for k in ("gaustudio", "gaustudio.datasets", "plyfile")}
    sys.modules.update({"gaustudio": pkg, "gaustudio.datasets": ds, "plyfile": ply})
    try:
        mods = []
        for rel in (("utils", "cameras_utils.py"), ("datasets", "utils.py")):
            sp = importlib.util.spec_from_file_location("ref_" + rel[1][:-3], os.path.join(REF, *rel))
            m = importlib.util.module_from_spec(sp)
            sp.loader.exec_module(m)
            mods.append(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    cu, du = mods
    rng = np.random.default_rng(17)
    entries = []
    for i in range(4):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        r, x, y, z = q
        rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                        [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                        [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        w, h = [(1920, 1080), (800, 800), (1296, 968), (640, 360)][i]
        entries.append({"id": i, "img_name": f"frame_{(7 * i) % 4:05d}", "width": w, "height": h,
                        "position": (rng.normal(size=3) * 3).tolist(), "rotation": [row.tolist() for row in rot],
                        "fy": float(h * rng.uniform(0.8, 1.6)), "fx": float(w * rng.uniform(0.5, 1.1))})
    out = {}
    back = []
    for i, e in enumerate(entries):
        cam = cu.JSON_to_camera(e)
        out[f"view_{i}"] = cam.world_view_transform.numpy()
        out[f"full_{i}"] = cam.full_proj_transform.numpy()
        out[f"campos_{i}"] = cam.camera_center.numpy()
        out[f"fov_{i}"] = np.array([cam.FoVx, cam.FoVy], dtype=np.float64)
        out[f"RT_{i}"] = np.concatenate([np.asarray(cam.R, dtype=np.float64).reshape(-1), np.asarray(cam.T, dtype=np.float64)])
        back.append(du.camera_to_JSON(i, cam))
    np.savez_compressed(os.path.join(HERE, "py_cameras.npz"), **out)
    with open(os.path.join(HERE, "py_cameras.json"), "w") as f:
        json.dump({"entries": entries, "roundtrip": back}, f, indent=1)
    print("wrote py_cameras.npz / py_cameras.json")


if __name__ == "__main__":
    import math
    main()
    make_post_golden()
    make_camera_golden()
