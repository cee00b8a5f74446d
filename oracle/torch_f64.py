"""float64 torch-autograd restatement of the reference forward pass.  TEST INFRASTRUCTURE ONLY.

Purpose: an INDEPENDENT check of the C oracle's backward (oracle/gsr_oracle.c restates the
hand-derived gradients of $RAST/cuda_rasterizer/backward.cu; here the gradients come from autograd
through a float64 restatement of forward.cu, so a transcription error in either shows up).

The integer structure (tile ranges + depth-sorted lists, i.e. what rasterizer_impl.cu:70-138
produces) is taken from the oracle state; every floating-point quantity is recomputed here in
float64 from the raw inputs:
  projection / in_frustum       auxiliary.h:139-164, forward.cu:196-200
  cov3D from scale + raw quat   forward.cu:118-152 (q not normalised)
  EWA cov2D, +0.3, conic        forward.cu:74-113, 215-219
  SH -> RGB (+0.5, clamp)       forward.cu:20-71
  compositing                   forward.cu:315-396 (power>0 / alpha<1/255 / T<1e-4 rules, Q8; median, Q9)
Known, documented deviations of the reference backward from the true gradient are emulated where
they matter (Q14 extra opacity term) or avoided by the test scene (Q1 bg=0, Q6 no clamped
Gaussians, Q18 scale_modifier=1).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh_rgb(deg, sh, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def per_gaussian(means, scales, rots, shs, colors_precomp, view, proj, campos, W, H, tanfovx, tanfovy, deg,
                 scale_modifier=1.0):
    P = means.shape[0]
    ones = torch.ones(P, 1, dtype=means.dtype)
    ph = torch.cat([means, ones], 1) @ proj
    p_w = 1.0 / (ph[:, 3] + 0.0000001)
    ndc = ph[:, :2] * p_w[:, None]
    pv = torch.cat([means, ones], 1) @ view
    t = pv[:, :3]
    depth = t[:, 2]
    r, x, y, z = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]
    Rstd = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    s = scale_modifier * scales
    Sigma = Rstd @ torch.diag_embed(s * s) @ Rstd.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = t[:, 2]
    tx = torch.clamp(t[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -(fy * ty) / (tz * tz)], -1)], -2)      # [P,2,3]
    Wc = view[:3, :3].transpose(0, 1)                                                     # W2C rotation
    A = J @ Wc
    cov = A @ Sigma @ A.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], -1)
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    if colors_precomp is None:
        d = means - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = eval_sh_rgb(deg, shs, d)
    else:
        rgb = colors_precomp
    return dict(xy=torch.stack([px, py], -1), depth=depth, conic=conic, rgb=rgb, det=det, cov=torch.stack([a, b, c], -1))


def render(means, scales, rots, opac, shs, colors_precomp, cam, deg, ranges, point_list, scale_modifier=1.0):
    """Differentiable float64 render.  Returns (color[3,H,W], depth[1,H,W], median[3,H,W],
    opacity[1,H,W], n_contrib[H,W], extra) where extra['w_sum_op'](g_op) gives the Q14 term."""
    W, H = cam.width, cam.height
    view = cam.viewmatrix.double()
    proj = cam.projmatrix.double()
    campos = cam.campos.double()
    pg = per_gaussian(means, scales, rots, shs, colors_precomp, view, proj, campos, W, H, cam.tanfovx,
                      cam.tanfovy, deg, scale_modifier)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    color = torch.zeros(3, H, W, dtype=torch.float64)
    depth = torch.zeros(1, H, W, dtype=torch.float64)
    median = torch.zeros(3, H, W, dtype=torch.float64)
    median[0] = 15.0
    opacity = torch.zeros(1, H, W, dtype=torch.float64)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    weights = []   # (ids, ys, xs, w[Npix, L]) for the Q14 extra term
    pl = torch.as_tensor(point_list.astype("int64"))
    for tile in range(gx * gy):
        r0, r1 = int(ranges[tile][0]), int(ranges[tile][1])
        tx, ty = tile % gx, tile // gx
        x0, y0 = tx * 16, ty * 16
        x1, y1 = min(x0 + 16, W), min(y0 + 16, H)
        if r1 <= r0:
            continue
        ids = pl[r0:r1]
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        pxf = xs.reshape(-1).double()
        pyf = ys.reshape(-1).double()
        xy = pg["xy"][ids]
        con = pg["conic"][ids]
        dx = xy[None, :, 0] - pxf[:, None]
        dy = xy[None, :, 1] - pyf[:, None]
        power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
        alpha = torch.clamp_max(opac[ids, 0][None] * torch.exp(power), 0.99)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        om = torch.where(valid, 1 - alpha, torch.ones_like(alpha))
        Tincl = torch.cumprod(om, dim=1)
        Tbefore = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], 1)
        stop = valid & (Tincl < 0.0001)
        stopped = torch.cumsum(stop.to(torch.int64), 1) > 0
        applied = valid & ~stopped
        w = torch.where(applied, alpha * Tbefore, torch.zeros_like(alpha))
        rgb = pg["rgb"][ids]
        dep = pg["depth"][ids]
        col = w @ rgb
        dacc = w @ dep
        Tfin = torch.prod(torch.where(applied, 1 - alpha, torch.ones_like(alpha)), dim=1)
        ismed = applied & (Tbefore > 0.5) & (Tbefore * (1 - alpha) < 0.5)
        first = (torch.cumsum(ismed.to(torch.int64), 1) == 1) & ismed
        has = first.any(dim=1)
        med_d = torch.where(has, (first.double() * dep[None]).sum(1), torch.full_like(dacc, 15.0))
        med_w = (first.double() * w).sum(1)
        med_id = (first.double() * ids[None].double()).sum(1)
        L = ids.shape[0]
        idx1 = torch.arange(1, L + 1)[None].expand_as(applied)
        nc = torch.where(applied, idx1, torch.zeros_like(idx1)).max(dim=1).values
        hh, ww = y1 - y0, x1 - x0
        color[:, y0:y1, x0:x1] = col.t().reshape(3, hh, ww)
        depth[0, y0:y1, x0:x1] = dacc.reshape(hh, ww)
        median[0, y0:y1, x0:x1] = med_d.reshape(hh, ww)
        median[1, y0:y1, x0:x1] = med_w.reshape(hh, ww)
        median[2, y0:y1, x0:x1] = med_id.reshape(hh, ww)
        opacity[0, y0:y1, x0:x1] = (1 - Tfin).reshape(hh, ww)
        n_contrib[y0:y1, x0:x1] = nc.reshape(hh, ww)
        weights.append((ids, ys.reshape(-1), xs.reshape(-1), w.detach()))

    def q14_extra(g_op):
        """sum over pixels of alpha*T_before*dL_dfinal_opacity per Gaussian (backward.cu:575), [P]."""
        out = torch.zeros(means.shape[0], dtype=torch.float64)
        for ids, ys, xs, w in weights:
            g = g_op[0, ys, xs].double()
            out.index_add_(0, ids, (w * g[:, None]).sum(0))
        return out

    return color, depth, median, opacity, n_contrib, dict(q14_extra=q14_extra, pg=pg)
