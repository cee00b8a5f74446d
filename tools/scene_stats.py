#!/usr/bin/env python
"""Work statistics of a bench workload on the GPU: per-tile list lengths, traversal depth (n_contrib),
fraction of (8x8 block, instance) pairs surviving an exact box cull, fraction of (pixel, instance) pairs live."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustudio_amd import scenes, _C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import hip_forward, scene_kwargs

P, W, H, D = 1_000_000, 1920, 1080, 3
cam = scenes.make_camera(W, H); sc = scenes.make_scene(P, cam, seed=0)
hs = hip_forward(sc, cam, D, scene_kwargs(sc, True, False))
r = hs["ranges"].long(); L = (r[:, 1] - r[:, 0])
print("R", hs["num_rendered"], "tiles", L.numel(), "len mean", float(L.float().mean()), "max", int(L.max()),
      "p50/p90/p99", [int(torch.quantile(L.float(), q)) for q in (0.5, 0.9, 0.99)])
nc = hs["n_contrib"].long(); fT = hs["final_T"]
print("n_contrib mean", float(nc.float().mean()), "max", int(nc.max()), "saturated pixels", float((fT < 1e-3).float().mean()))
gx = (W + 15) // 16
# per tile: traversal needed = max n_contrib over tile (block terminates when all done or list ends)
ncpad = torch.zeros(((H + 15) // 16) * 16, gx * 16, dtype=torch.long, device=nc.device); ncpad[:H, :W] = nc
tmax = ncpad.view(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values
print("sum over tiles of max n_contrib / R:", float(tmax.sum()) / hs["num_rendered"])
g = torch.Generator().manual_seed(0)
tiles = torch.randperm(L.numel(), generator=g)[:200]
xy = hs["means2D"]; co = hs["conic_opacity"]; pl = hs["point_list"].long()
tot_inst = tot_blockhit = tot_pairs = tot_live = 0
for t in tiles.tolist():
    a, b = int(r[t, 0]), int(r[t, 1])
    if b <= a: continue
    ids = pl[a:b]
    tx, ty = t % gx, t // gx
    ys, xs = torch.meshgrid(torch.arange(16, device=xy.device), torch.arange(16, device=xy.device), indexing="ij")
    pxs = (tx * 16 + xs).reshape(-1).float(); pys = (ty * 16 + ys).reshape(-1).float()
    dx = xy[ids, 0][None] - pxs[:, None]; dy = xy[ids, 1][None] - pys[:, None]
    c = co[ids]
    power = -0.5 * (c[:, 0][None] * dx * dx + c[:, 2][None] * dy * dy) - c[:, 1][None] * dx * dy
    alpha = torch.clamp_max(c[:, 3][None] * torch.exp(power), 0.99)
    live = (power <= 0) & (alpha >= 1 / 255)
    blk = ((ys // 8) * 2 + xs // 8).reshape(-1)
    bh = torch.stack([live[blk == k].any(0) for k in range(4)])
    tot_inst += ids.numel(); tot_blockhit += int(bh.sum()); tot_pairs += live.numel(); tot_live += int(live.sum())
print(f"sampled {len(tiles)} tiles: instances {tot_inst}; (block,instance) with any live pixel: {tot_blockhit / (4 * tot_inst):.3f}; live (pixel,instance) pairs: {tot_live / tot_pairs:.4f}")
