#!/usr/bin/env python
"""The gs-extract-mesh loop (gaustudio/scripts/extract_mesh.py:86-146) with every stage on the MI355X:

    Gaussian PLY + cameras.json  ->  GaussianRasterizer (median depth, opacity)  ->  depth_to_points
                                 ->  TSDFVolume.integrate                         ->  extract_triangle_mesh -> PLY

Runs on a synthetic shell of Gaussians written to / read back from the reference's on-disk formats, so it needs no
dataset:   python examples/extract_mesh_synthetic.py [out_dir]
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer, formats, postprocess as pp, scenes  # noqa: E402
from gaustudio_amd.tsdf import TSDFVolume  # noqa: E402


def write_inputs(out):
    g = torch.Generator().manual_seed(0)
    P = 200_000
    d = torch.randn(P, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    bumps = 1.0 + 0.08 * torch.sin(5 * d[:, 0:1]) * torch.cos(4 * d[:, 1:2])
    cloud = formats.GaussianCloud(xyz=d * bumps, f_dc=(torch.rand(P, 1, 3, generator=g) - 0.5) / 0.28209479177387814,
                                  f_rest=torch.zeros(P, 15, 3), opacity=torch.full((P, 1), 3.0),      # sigmoid -> 0.95
                                  scale=torch.full((P, 3), math.log(0.008)), rot=torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1))
    formats.export_gaussian_ply(os.path.join(out, "point_cloud.ply"), cloud)
    cams = []
    for i, c in enumerate(scenes.ring_cameras(24, 640, 480, radius=3.2, elevation=0.35)
                          + scenes.ring_cameras(12, 640, 480, radius=3.2, elevation=-0.8)):
        w2c = c.viewmatrix.t().numpy().astype(np.float64)
        c2w = np.linalg.inv(w2c)
        f = c.width / (2 * c.tanfovx)
        cams.append({"id": i, "img_name": f"view_{i:03d}", "width": c.width, "height": c.height,
                     "position": c2w[:3, 3].tolist(), "rotation": c2w[:3, :3].tolist(), "fx": f, "fy": f})
    with open(os.path.join(out, "cameras.json"), "w") as fjson:
        json.dump(cams, fjson)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "extract_mesh_out"
    os.makedirs(out, exist_ok=True)
    write_inputs(out)
    dev = torch.device("cuda:0")
    pcd = formats.load_gaussian_ply(os.path.join(out, "point_cloud.ply"), device=dev)
    cameras = formats.load_cameras_json(os.path.join(out, "cameras.json"))
    act = pcd.activated()
    volume = TSDFVolume(voxel_size=0.01, sdf_trunc=0.04, space_carving=False, capacity_blocks=1 << 18)   # extract_mesh.py:86
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rec in cameras:
        cam = rec.cam
        rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                           cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 0, cam.campos.to(dev), False, False)
        with torch.no_grad():
            _, _, _, median, opacity = GaussianRasterizer(rs)(means3D=act["means3D"], means2D=torch.zeros_like(act["means3D"]),
                                                               opacities=act["opacities"], shs=act["shs"], scales=act["scales"],
                                                               rotations=act["rotations"])
        depth = median[0].clone()
        invalid = opacity[0] < 0.5                                                  # extract_mesh.py:104-107
        depth[invalid] = 0
        f = cam.width / (2 * cam.tanfovx)
        K = torch.tensor([[f, 0, cam.width / 2], [0, f, cam.height / 2], [0, 0, 1]])
        # :110 compacts `pts[~invalid]` for the CPU library.  Here a masked pixel (depth 0) unprojects to the sensor origin,
        # which the integrate kernel skips, so the whole [H,W,3] map goes in -- as a map: the kernel then works in 32 x 32
        # pixel patches whose rays share voxels in both image directions (same volume, 2.3x faster than a flat list)
        pts = pp.depth_to_points(depth, K, cam.viewmatrix.t().contiguous(), "world")
        volume.integrate(pts, cam.campos)                                           # :115, points never leave the GPU
    vertices, faces = volume.extract_triangle_mesh(min_weight=5)                    # :145
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mesh = np.zeros(len(vertices), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4")])
    mesh["x"], mesh["y"], mesh["z"] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    formats.write_ply_vertices(os.path.join(out, "fused_mesh_vertices.ply"), mesh)
    np.save(os.path.join(out, "fused_mesh_faces.npy"), faces)
    r = np.linalg.norm(vertices, axis=1)
    print(f"{len(cameras)} views rendered, fused and meshed in {dt * 1e3:.0f} ms: {len(vertices)} vertices, {len(faces)} triangles, "
          f"radius {r.min():.3f} .. {r.max():.3f} (shell at 0.92 .. 1.08)")


if __name__ == "__main__":
    main()
