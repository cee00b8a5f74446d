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
print("R", hs["num_rendered"], "binned", hs["num_binned"], "tiles", L.numel(), "len mean", float(L.float().mean()), "max", int(L.max()),
      "p50/p90/p99", [int(torch.quantile(L.float(), q)) for q in (0.5, 0.9, 0.99)])
nc = hs["n_contrib"].long(); fT = hs["final_T"]
print("n_contrib mean", float(nc.float().mean()), "max", int(nc.max()), "saturated pixels", float((fT < 1e-3).float().mean()))
gx = (W + 15) // 16
# per tile: traversal needed = max n_contrib over tile (block terminates when all done or list ends)
ncpad = torch.zeros(((H + 15) // 16) * 16, gx * 16, dtype=torch.long, device=nc.device); ncpad[:H, :W] = nc
tmax = ncpad.view(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values
print("sum over tiles of max n_contrib / R:", float(tmax.sum()) / hs["num_binned"])
g = torch.Generator().manual_seed(0)
tiles = torch.randperm(L.numel(), generator=g)[:200]
xy = hs["means2D"]; co = hs["conic_opacity"]; pl = hs["point_list"].long()
tot_inst = tot_blockhit = tot_pairs = tot_live = tot_tilehit = 0
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
    tot_tilehit += int(live.any(0).sum())
    tot_inst += ids.numel(); tot_blockhit += int(bh.sum()); tot_pairs += live.numel(); tot_live += int(live.sum())
print(f"(tile,instance) with any live pixel in the 16x16 tile: {tot_tilehit / tot_inst:.3f}")
print(f"sampled {len(tiles)} tiles: instances {tot_inst}; (block,instance) with any live pixel: {tot_blockhit / (4 * tot_inst):.3f}; live (pixel,instance) pairs: {tot_live / tot_pairs:.4f}")

# ---- how tight is the kernel's conservative box test (gs_box_may_touch) compared with the exact "any live pixel"? ----
def box_test(xy, co, bx0, by0, bx1, by1):
    ha, nb, hc = -0.5 * co[:, 0], -co[:, 1], -0.5 * co[:, 2]
    pcut = torch.clamp_min(-torch.log(255.0 * co[:, 3]) - 0.001, -80.0)
    X0, X1 = xy[:, 0] - bx1, xy[:, 0] - bx0
    Y0, Y1 = xy[:, 1] - by1, xy[:, 1] - by0
    xn = torch.minimum(torch.clamp_min(X0, 0.0), X1); xn = torch.where((X0 <= 0) & (X1 >= 0), torch.zeros_like(xn), torch.where(X0 > 0, X0, X1))
    yn = torch.where((Y0 <= 0) & (Y1 >= 0), torch.zeros_like(Y0), torch.where(Y0 > 0, Y0, Y1))
    inside = (xn == 0) & (yn == 0)
    dy = torch.minimum(torch.maximum(-0.5 * nb * xn / hc, Y0), Y1)
    px_ = ha * xn * xn + (hc * dy + nb * xn) * dy
    dx = torch.minimum(torch.maximum(-0.5 * nb * yn / ha, X0), X1)
    py_ = hc * yn * yn + (ha * dx + nb * yn) * dx
    best = torch.where(xn != 0, px_, torch.full_like(px_, -3e38))
    best = torch.where(yn != 0, torch.maximum(best, py_), best)
    return (pcut <= 0) & (inside | (best >= pcut - 0.05))

tot = hit88 = hit816 = ex816 = 0
for t in tiles.tolist():
    a, b = int(r[t, 0]), int(r[t, 1])
    if b <= a: continue
    ids = pl[a:b]; tx, ty = t % gx, t // gx
    for q in range(4):
        bx0 = tx * 16 + (q & 1) * 8; by0 = ty * 16 + (q >> 1) * 8
        hit88 += int(box_test(xy[ids], co[ids], bx0, by0, min(bx0 + 7, W - 1), min(by0 + 7, H - 1)).sum())
    for w in range(2):
        bx0 = tx * 16 + w * 8; by0 = ty * 16
        hit816 += int(box_test(xy[ids], co[ids], bx0, by0, min(bx0 + 7, W - 1), min(by0 + 15, H - 1)).sum())
    tot += ids.numel()
print(f"kernel box test pass rate: 8x8 {hit88 / (4 * tot):.3f} (exact any-live {tot_blockhit / (4 * tot_inst):.3f}); 8x16 {hit816 / (2 * tot):.3f}")

# ---- what would per-quarter (4x4) instance lists inside a wave buy composite_fwd?  walk length of a wave =
# instances passing its 8x8 box test (now) vs max over its four 4x4 quarters of their own pass counts, the quarters
# re-synchronising every `batch` list entries (the staged batch).  Depth: the block's deepest n_contrib. ----
ncp = ncpad.view(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 16, 16)   # [tile][y][x]
it8 = 0; it4 = {64: 0, 256: 0, 1 << 30: 0}; it4_mean = 0; it2 = {256: 0}; it8e = {256: 0}; it8e_mean = 0
for t in tiles.tolist():
    a, b = int(r[t, 0]), int(r[t, 1])
    if b <= a: continue
    ids = pl[a:b]; tx, ty = t % gx, t // gx
    for blk in range(4):
        bx0 = tx * 16 + (blk & 1) * 8; by0 = ty * 16 + (blk >> 1) * 8
        depth = int(ncp[t, (blk >> 1) * 8:(blk >> 1) * 8 + 8, (blk & 1) * 8:(blk & 1) * 8 + 8].max())
        depth = min(len(ids), (depth + 63) // 64 * 64)    # saturation is checked per 64-entry sub-batch
        if depth == 0: continue
        sel = ids[:depth]
        h8 = box_test(xy[sel], co[sel], bx0, by0, min(bx0 + 7, W - 1), min(by0 + 7, H - 1))
        it8 += int(h8.sum())
        hq = []
        for q in range(4):
            qx0 = bx0 + (q & 1) * 4; qy0 = by0 + (q >> 1) * 4
            if qx0 > W - 1 or qy0 > H - 1: hq.append(torch.zeros_like(h8)); continue
            hq.append(box_test(xy[sel], co[sel], qx0, qy0, min(qx0 + 3, W - 1), min(qy0 + 3, H - 1)) & h8)
        hq = torch.stack(hq).long()                          # [4][depth]
        it4_mean += float(hq.sum()) / 4
        for bs in it4:
            if bs >= depth: it4[bs] += int(hq.sum(1).max())
            else:
                pad = (-depth) % bs
                hp = torch.nn.functional.pad(hq, (0, pad)).view(4, -1, bs).sum(2)
                it4[bs] += int(hp.max(0).values.sum())
        hh = torch.stack([hq[0] | hq[1], hq[2] | hq[3]])     # two 8x4 halves
        pad = (-depth) % 256
        it2[256] += int(torch.nn.functional.pad(hh, (0, pad)).view(2, -1, 256).sum(2).max(0).values.sum())
        # eight 4x2 sub-blocks (8 lanes each): would finer lists pay?  
        he = []
        for e in range(8):
            ex0 = bx0 + (e & 1) * 4; ey0 = by0 + (e >> 1) * 2
            if ex0 > W - 1 or ey0 > H - 1: he.append(torch.zeros_like(h8)); continue
            he.append(box_test(xy[sel], co[sel], ex0, ey0, min(ex0 + 3, W - 1), min(ey0 + 1, H - 1)) & h8)
        he = torch.stack(he).long()
        it8e_mean += float(he.sum()) / 8
        it8e[256] += int(torch.nn.functional.pad(he, (0, pad)).view(8, -1, 256).sum(2).max(0).values.sum())
print(f"wave walk length, 8x8 lists: {it8}; 4x4 quarter lists: mean-of-quarters {it4_mean:.0f} ({it4_mean / it8:.3f}), "
      + ", ".join(f"resync/{k if k < 1 << 30 else 'never'} {v} ({v / it8:.3f})" for k, v in it4.items())
      + f"; 8x4 halves resync/256 {it2[256]} ({it2[256] / it8:.3f})"
      + f"; 4x2 eighths: mean-of-eighths {it8e_mean:.0f} ({it8e_mean / it8:.3f}), resync/256 {it8e[256]} ({it8e[256] / it8:.3f})")

# ---- composite_bwd with per-quarter lists: steps of a workgroup per batch of B staged instances = max over its four
# waves of (max over the wave's quarters of its hits in the batch); walked back to front from the tile's deepest
# n_contrib, every quarter cut at its own deepest n_contrib. ----
res = {B: [0, 0, 0.0] for B in (32, 64, 128, 256)}   # [sum over batches of max_w n_w, sum of mean_w n_w, sum of mean over 16 quarters]
for t in tiles.tolist():
    a, b = int(r[t, 0]), int(r[t, 1])
    if b <= a: continue
    ids = pl[a:b]; tx, ty = t % gx, t // gx
    bmax = min(int(ncp[t].max()), len(ids))
    if bmax == 0: continue
    sel = ids[:bmax]
    H16 = torch.zeros(16, bmax, dtype=torch.long, device=xy.device)
    pos = torch.arange(bmax, device=xy.device)
    for blk in range(4):
        bx0 = tx * 16 + (blk & 1) * 8; by0 = ty * 16 + (blk >> 1) * 8
        for q in range(4):
            qx0 = bx0 + (q & 1) * 4; qy0 = by0 + (q >> 1) * 4
            if qx0 > W - 1 or qy0 > H - 1: continue
            qmax = int(ncp[t, qy0 - ty * 16:qy0 - ty * 16 + 4, qx0 - tx * 16:qx0 - tx * 16 + 4].max())
            H16[blk * 4 + q] = (box_test(xy[sel], co[sel], qx0, qy0, min(qx0 + 3, W - 1), min(qy0 + 3, H - 1)) & (pos < qmax)).long()
    Hr = torch.flip(H16, dims=[1])           # back to front
    for B in res:
        pad = (-bmax) % B
        hb = torch.nn.functional.pad(Hr, (0, pad)).view(4, 4, -1, B).sum(3)      # [wave][quarter][batch]
        nw = hb.max(1).values                                                     # [wave][batch]
        res[B][0] += int(nw.max(0).values.sum()); res[B][1] += float(nw.float().mean(0).sum()); res[B][2] += float(hb.float().mean((0, 1)).sum())
print("composite_bwd quarter-list census (steps per workgroup): " + "; ".join(
    f"B={B}: max-wave {v[0]}, mean-wave {v[1]:.0f} (max/mean {v[0] / v[1]:.3f}), mean-quarter {v[2]:.0f} (wave/quarter {v[1] / v[2]:.3f})" for B, v in res.items()))

# ---- composite_bwd: how often do two quarters of a wave meet in an instance within one group of GSR_BWQ_U = 4 list
# steps (the case the plane update's "turns" exist for)?  Per (wave, round of 128, group): the four quarters' list
# entries of the group, pairwise disjoint or not. ----
groups = coll = 0
for t in tiles.tolist():
    a, b = int(r[t, 0]), int(r[t, 1])
    if b <= a: continue
    ids = pl[a:b]; tx, ty = t % gx, t // gx
    bmax = min(int(ncp[t].max()), len(ids))
    if bmax == 0: continue
    sel = ids[:bmax]
    pos = torch.arange(bmax, device=xy.device)
    for blk in range(4):
        bx0 = tx * 16 + (blk & 1) * 8; by0 = ty * 16 + (blk >> 1) * 8
        hq = []
        for q in range(4):
            qx0 = bx0 + (q & 1) * 4; qy0 = by0 + (q >> 1) * 4
            if qx0 > W - 1 or qy0 > H - 1:
                hq.append(torch.zeros(bmax, dtype=torch.bool)); continue
            qmax = int(ncp[t, qy0 - ty * 16:qy0 - ty * 16 + 4, qx0 - tx * 16:qx0 - tx * 16 + 4].max())
            hq.append((box_test(xy[sel], co[sel], qx0, qy0, min(qx0 + 3, W - 1), min(qy0 + 3, H - 1)) & (pos < qmax)).cpu())
        for top in range(bmax, 0, -128):
            lo = max(0, top - 128)
            lists = [torch.nonzero(torch.flip(h[lo:top], dims=[0])).flatten().tolist() for h in hq]
            n = max(len(l) for l in lists)
            for g0 in range(0, n, 4):
                seen = set(); c = False
                for l in lists:
                    s_ = set(l[g0:g0 + 4])
                    if seen & s_: c = True
                    seen |= s_
                groups += 1; coll += int(c)
print(f"composite_bwd groups of 4 steps: {groups}; with two quarters of the wave in the same instance: {coll} ({coll / max(1, groups):.3f})")

# ---- dead instances (no live pixel in the tile) by their place in the Gaussian's tile rect: how many would a cull of the
# rect's four CORNER tiles remove?  (rect of a Gaussian = min / max tile of its instances: the binning emits whole rects) ----
tile_of = torch.repeat_interleave(torch.arange(L.numel(), device=pl.device), L)            # tile of every list position
txs, tys = tile_of % gx, tile_of // gx
big = 1 << 20
mnx = torch.full((P,), big, device=pl.device).scatter_reduce(0, pl, txs, "amin"); mxx = torch.full((P,), -1, device=pl.device).scatter_reduce(0, pl, txs, "amax")
mny = torch.full((P,), big, device=pl.device).scatter_reduce(0, pl, tys, "amin"); mxy = torch.full((P,), -1, device=pl.device).scatter_reduce(0, pl, tys, "amax")
n_inst = n_dead = n_corner = n_dead_corner = n_corner_elig = 0
for t in tiles.tolist():
    a, b = int(r[t, 0]), int(r[t, 1])
    if b <= a: continue
    ids = pl[a:b]; tx, ty = t % gx, t // gx
    ys, xs = torch.meshgrid(torch.arange(16, device=xy.device), torch.arange(16, device=xy.device), indexing="ij")
    pxs = (tx * 16 + xs).reshape(-1).float(); pys = (ty * 16 + ys).reshape(-1).float()
    inb = ((pxs < W) & (pys < H))[:, None]
    dx = xy[ids, 0][None] - pxs[:, None]; dy = xy[ids, 1][None] - pys[:, None]
    c = co[ids]
    power = -0.5 * (c[:, 0][None] * dx * dx + c[:, 2][None] * dy * dy) - c[:, 1][None] * dx * dy
    alpha = torch.clamp_max(c[:, 3][None] * torch.exp(power), 0.99)
    dead = ~(((power <= 0) & (alpha >= 1 / 255) & inb).any(0))
    w_ = mxx[ids] - mnx[ids] + 1; h_ = mxy[ids] - mny[ids] + 1
    corner = ((mnx[ids] == tx) | (mxx[ids] == tx)) & ((mny[ids] == ty) | (mxy[ids] == ty)) & (w_ >= 2) & (h_ >= 2)
    n_inst += ids.numel(); n_dead += int(dead.sum()); n_corner += int(corner.sum()); n_dead_corner += int((dead & corner).sum())
print(f"dead instances (no live pixel in the tile): {n_dead / n_inst:.4f} of the binned instances; at a corner of a >= 2x2 rect: "
      f"{n_dead_corner / n_inst:.4f} ({n_dead_corner / max(1, n_dead):.3f} of the dead; {n_dead_corner / max(1, n_corner):.3f} of the corner instances are dead)")
