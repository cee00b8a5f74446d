#!/usr/bin/env python
"""Census for composite_bwd (VERDICT r5 #2a): what would a LENGTH-BALANCED assignment of a tile's sixteen 4x4 quarters to its
four waves buy over today's spatial grouping (wave = one 8x8 block = its four quarters)?

A wave walks, per 128-instance round, max over its four quarters of the quarter's list length (the quarters advance in
lock-step, an exhausted quarter idles on the sentinel), in groups of GSR_BWQ_U = 4 steps.  Counted per sampled tile:

  spatial          today's grouping                                   (gsr_kernels_bwd.hip composite_bwd_quarter_kernel)
  static/qmax      quarters sorted by their deepest n_contrib, grouped by fours; ONE permutation per tile, known before the
                   walk (the only kind a kernel can afford: the pixel state of a lane is loaded once)
  static/total     sorted by the total list length of the whole walk (needs a pre-pass over the masks: upper bound of "static")
  per-round        re-sorted in every round (not implementable -- a lane's T / S recurrences cannot change pixels: bound)
  ideal            no quarter ever waits: sum of the quarters' lengths / 4

for the SUM over waves (SIMD issue slots spent) and the MAX over waves (the workgroup's critical path between barriers).

Runs on the CPU (the oracle's forward supplies lists and n_contrib: zero GPU minutes) or with --gpu through the library.
TOOL, not product: imports the test-only oracle."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaustudio_amd import scenes  # noqa: E402

BATCH, U = 128, 4


def box_test(xy, co, bx0, by0, bx1, by1):
    """The kernels' conservative block test (gs_quarter_mask / gs_box_may_touch), as in tools/scene_stats.py."""
    ha, nb, hc = -0.5 * co[:, 0], -co[:, 1], -0.5 * co[:, 2]
    pcut = torch.clamp_min(-torch.log(255.0 * co[:, 3]) - 0.001, -80.0)
    X0, X1 = xy[:, 0] - bx1, xy[:, 0] - bx0
    Y0, Y1 = xy[:, 1] - by1, xy[:, 1] - by0
    xn = torch.where((X0 <= 0) & (X1 >= 0), torch.zeros_like(X0), torch.where(X0 > 0, X0, X1))
    yn = torch.where((Y0 <= 0) & (Y1 >= 0), torch.zeros_like(Y0), torch.where(Y0 > 0, Y0, Y1))
    inside = (xn == 0) & (yn == 0)
    dy = torch.minimum(torch.maximum(-0.5 * nb * xn / hc, Y0), Y1)
    px_ = ha * xn * xn + (hc * dy + nb * xn) * dy
    dx = torch.minimum(torch.maximum(-0.5 * nb * yn / ha, X0), X1)
    py_ = hc * yn * yn + (ha * dx + nb * yn) * dx
    best = torch.where(xn != 0, px_, torch.full_like(px_, -3e38))
    best = torch.where(yn != 0, torch.maximum(best, py_), best)
    return (pcut <= 0) & (inside | (best >= pcut - 0.05))


def workload(name):
    if name == "C3":
        cam = scenes.make_camera(1920, 1080); return scenes.make_scene(1_000_000, cam, seed=0), cam
    if name == "C2":
        cam = scenes.make_camera(800, 800); return scenes.make_scene(300_000, cam, seed=0), cam
    if name == "C5":
        cam = scenes.make_camera(3840, 2160); return scenes.make_scene(2_500_000, cam, seed=0), cam
    if name == "C2-clustered":
        sc = scenes.make_clustered_scene(300_000, 800, cam_distance=11.0, seed=0)      # bench.py's C2-clustered, view 1 of the ring
        return sc, scenes.ring_cameras(5, 800, 800, radius=11.0)[1]
    raise SystemExit(name)


def forward_state(sc, cam, gpu):
    if gpu:
        from util import hip_forward, scene_kwargs
        hs = hip_forward(sc, cam, 3, scene_kwargs(sc, True, False))
        return {k: hs[k].cpu() for k in ("ranges", "point_list", "means2D", "conic_opacity", "n_contrib")}
    from oracle import pyoracle as po
    from util import oracle_forward, scene_kwargs
    st = oracle_forward(po, sc, cam, 3, scene_kwargs(sc, True, False), tight=True)
    return {k: torch.from_numpy(np.ascontiguousarray(st[k]).astype(np.int64 if st[k].dtype.kind in "ui" else np.float32))
            for k in ("ranges", "point_list", "means2D", "conic_opacity", "n_contrib")}


def steps(c, groups):
    """c [16][nb] list lengths per quarter and round; groups = 4 index lists -> (sum over waves, max over waves), in steps rounded
    up to whole groups of U."""
    per_wave = torch.stack([c[g].max(0).values for g in groups])          # [4][nb]
    per_wave = (per_wave + U - 1) // U * U
    return int(per_wave.sum()), int(per_wave.max(0).values.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--tiles", type=int, default=200)
    ap.add_argument("--gpu", action="store_true")
    a = ap.parse_args()
    sc, cam = workload(a.workload)
    W, H = cam.width, cam.height
    st = forward_state(sc, cam, a.gpu)
    r = st["ranges"].long(); pl = st["point_list"].long(); xy = st["means2D"]; co = st["conic_opacity"]; nc = st["n_contrib"].long()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ncpad = torch.zeros(gy * 16, gx * 16, dtype=torch.long); ncpad[:H, :W] = nc
    ncp = ncpad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 16, 16)
    L = r[:, 1] - r[:, 0]
    cand = torch.nonzero(L > 0).flatten()
    g = torch.Generator().manual_seed(0)
    tiles = cand[torch.randperm(cand.numel(), generator=g)[:a.tiles]]
    spatial = [[(2 * by + qy) * 4 + 2 * bx + qx for qy in range(2) for qx in range(2)] for by in range(2) for bx in range(2)]
    tot = {k: [0, 0] for k in ("spatial", "static/qmax", "static/total", "per-round")}
    ideal = 0.0
    for t in tiles.tolist():
        ids = pl[int(r[t, 0]):int(r[t, 1])]
        tx, ty = t % gx, t // gx
        bmax = min(int(ncp[t].max()), len(ids))
        if bmax == 0:
            continue
        sel = ids[:bmax]
        pos = torch.arange(bmax)
        H16 = torch.zeros(16, bmax, dtype=torch.long)
        qmax = torch.zeros(16, dtype=torch.long)
        for q in range(16):
            qx0, qy0 = tx * 16 + (q & 3) * 4, ty * 16 + (q >> 2) * 4
            if qx0 > W - 1 or qy0 > H - 1:
                continue
            qmax[q] = int(ncp[t, (q >> 2) * 4:(q >> 2) * 4 + 4, (q & 3) * 4:(q & 3) * 4 + 4].max())
            H16[q] = (box_test(xy[sel], co[sel], qx0, qy0, min(qx0 + 3, W - 1), min(qy0 + 3, H - 1)) & (pos < qmax[q])).long()
        Hr = torch.flip(H16, dims=[1])
        pad = (-bmax) % BATCH
        c = torch.nn.functional.pad(Hr, (0, pad)).view(16, -1, BATCH).sum(2)          # [16][nb]
        ideal += float(c.sum()) / 4
        for k, v in zip(("spatial",), (steps(c, spatial),)):
            tot[k][0] += v[0]; tot[k][1] += v[1]
        for k, key in (("static/qmax", qmax), ("static/total", c.sum(1))):
            order = torch.argsort(key, descending=True, stable=True).tolist()
            v = steps(c, [order[4 * w:4 * w + 4] for w in range(4)])
            tot[k][0] += v[0]; tot[k][1] += v[1]
        # per-round re-sorting (bound)
        cs = torch.sort(c, dim=0, descending=True).values.view(4, 4, -1).max(1).values
        cs = (cs + U - 1) // U * U
        tot["per-round"][0] += int(cs.sum()); tot["per-round"][1] += int(cs.max(0).values.sum())
    base = tot["spatial"]
    print(f"{a.workload}: {len(tiles)} tiles, rounds of {BATCH}, groups of {U} steps; ideal (no waiting) {ideal:.0f} wave-steps")
    for k, v in tot.items():
        print(f"  {k:13s} sum over waves {v[0]:8d} ({v[0] / base[0]:.3f} of spatial; {v[0] / ideal:.3f} of ideal)   "
              f"max over waves (workgroup path) {v[1]:8d} ({v[1] / base[1]:.3f})")


if __name__ == "__main__":
    main()
