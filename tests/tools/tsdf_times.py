#!/usr/bin/env python
"""Row f3 measurement: TSDF integration of one 1080p depth map worth of points and mesh extraction on the GPU, next to
the CPU oracle (oracle/tsdf_oracle.c, a single-threaded restatement of vdbfusion's Integrate -- the reference runs the
real vdbfusion on the host after a full-frame D2H, extract_mesh.py:115) on a sample of the same points."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from gaustudio_amd.tsdf import TSDFVolume  # noqa: E402
from oracle import tsdf_pyoracle as to  # noqa: E402

dev = torch.device("cuda:0")
W, H, f = 1920, 1080, 1400.0
vs, tr = 0.01, 0.04                                   # the reference's scene settings (extract_mesh.py:86)
yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
depth = 2.0 + 0.3 * torch.sin(xx / 90.0) * torch.cos(yy / 70.0)          # a wavy wall ~2 m away
pts = torch.stack([(xx - W / 2) / f * depth, (yy - H / 2) / f * depth, depth], -1).reshape(-1, 3).to(dev)
origin = np.zeros(3, np.float32)
N = pts.shape[0]

vol = TSDFVolume(vs, tr, capacity_blocks=1 << 19)
vol.integrate(pts, origin)
torch.cuda.synchronize()
reps = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    vol.integrate(pts, origin)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
c, t, w, s = vol.export_voxels()
updates = int(w.sum().item()) // (reps + 1)
print(f"integrate: {N} points, {updates} voxel updates ({updates / N:.1f} per ray), {ms:.3f} ms = {N / ms / 1e3:.0f} Mpoints/s, "
      f"{updates / ms / 1e6:.2f} G voxel updates/s (pre-aggregated per workgroup in LDS); {vol.occupied_blocks()[0].shape[0]} blocks, {c.shape[0]} voxels")
torch.cuda.synchronize()
t0 = time.perf_counter()
V, T = vol.extract_triangle_mesh_device(min_weight=5)
torch.cuda.synchronize()
print(f"extract_triangle_mesh: {V.shape[0]} vertices, {T.shape[0]} triangles in {(time.perf_counter() - t0) * 1e3:.1f} ms (incl. torch sort/scan)")

sample = pts[:: 16].cpu().numpy()
ov = to.Volume(vs, tr)
t0 = time.perf_counter()
ov.integrate(sample, origin)
dt = time.perf_counter() - t0
print(f"CPU oracle (1 thread, restatement of vdbfusion Integrate): {len(sample)} points in {dt:.2f} s = {len(sample) / dt / 1e6:.3f} Mpoints/s"
      f" -> full frame ~{N / (len(sample) / dt):.1f} s; GPU/CPU = {N / ms * 1e3 / (len(sample) / dt):.0f}x")
