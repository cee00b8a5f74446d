"""numpy restatement of the reference's post-render helpers.  TEST INFRASTRUCTURE ONLY.
Follows gaustudio/datasets/__init__.py: ndc_2_cam (:106-112), Camera.depth2point (:307-339),
Camera.depth2normal (:342-380).  Pinned against tests/golden/py_post.npz (outputs of the reference's own Camera
class, tests/golden/make_py_golden.py)."""
import numpy as np


def depth2point(depth, K, w2c=None):
    depth = depth.astype(np.float32)
    H, W = depth.shape
    x = (np.arange(W, dtype=np.float32) / np.float32(W - 1))[None, :] * np.float32(W - 1) * depth
    y = (np.arange(H, dtype=np.float32) / np.float32(H - 1))[:, None] * np.float32(H - 1) * depth
    cam = np.stack([x, y, depth], -1).astype(np.float64) @ np.linalg.inv(K.astype(np.float64).T)
    if w2c is None:
        return cam.astype(np.float32)
    c2w = np.linalg.inv(w2c.astype(np.float64))
    world = np.concatenate([cam, np.ones_like(cam[..., :1])], -1) @ c2w.T
    return world[..., :3].astype(np.float32)


def depth2normal(depth, K, w2c=None, k=3, d_min=1e-3, d_max=100000.0):
    P = depth2point(depth, K).astype(np.float64)
    H, W = depth.shape
    k = (k - 1) // 2
    Pp = np.zeros((H + 2 * k, W + 2 * k, 3))
    Pp[k:k + H, k:k + W] = P
    valid_p = (Pp[..., 2] > d_min) & (Pp[..., 2] < d_max)
    vert = Pp[:H, k:k + W] - Pp[2 * k:2 * k + H, k:k + W]
    hori = Pp[k:k + H, :W] - Pp[k:k + H, 2 * k:2 * k + W]
    valid = (valid_p[k:k + H, k:k + W] & valid_p[:H, k:k + W] & valid_p[2 * k:2 * k + H, k:k + W]
             & valid_p[k:k + H, :W] & valid_p[k:k + H, 2 * k:2 * k + W])
    n = -np.cross(vert, hori)
    n = n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
    if w2c is not None:
        n = n @ np.linalg.inv(w2c[:3, :3].astype(np.float64)).T
    n[~valid] = -1
    return n.astype(np.float32)


def masked_bilateral_filter(depth, mask, d=3, sigma_color=75.0, sigma_space=75.0):
    """gaustudio/scripts/extract_pcd.py:185-238 with cv2.dilate / cv2.bilateralFilter restated from OpenCV's published
    float32 algorithm (cv2 is not in this image: PARITY UNPINNED against the library itself): d x d dilation of the
    invalid mask ignoring out-of-image pixels; bilateral filter with radius max(d // 2, 1), circular window,
    BORDER_REFLECT_101, weights exp(-r^2 / (2 ss^2)) * exp(-dv^2 / (2 sc^2)) (cv2 tabulates the second factor)."""
    depth = depth.astype(np.float32); valid = mask != 0
    H, W = depth.shape
    r = d // 2
    new_mask = np.ones((H, W), bool)
    for j in range(-r, r + 1):
        for i in range(-r, r + 1):
            sh = np.ones((H, W), bool)
            ys, xs = slice(max(0, -j), min(H, H - j)), slice(max(0, -i), min(W, W - i))
            yd, xd = slice(max(0, j), min(H, H + j)), slice(max(0, i), min(W, W + i))
            sh[ys, xs] = valid[yd, xd]
            new_mask &= sh
    out = depth.copy()
    if not new_mask.any():
        return out, new_mask
    vmin, vmax = depth[new_mask].min(), depth[new_mask].max()
    rng = np.float32(vmax - vmin)
    if not rng > 0:
        return out, new_mask
    norm = np.where(new_mask, (depth - vmin) / rng, np.float32(0)).astype(np.float32)
    radius = max(d // 2, 1)
    pad = np.pad(norm, radius, mode="reflect")
    sc, ss = np.float32(-0.5 / (sigma_color * sigma_color)), np.float32(-0.5 / (sigma_space * sigma_space))
    num = np.zeros((H, W), np.float32); den = np.zeros((H, W), np.float32)
    for j in range(-radius, radius + 1):
        for i in range(-radius, radius + 1):
            if i * i + j * j > radius * radius:
                continue
            v = pad[radius + j:radius + j + H, radius + i:radius + i + W]
            w = (np.exp(np.float32(i * i + j * j) * ss) * np.exp((v - norm) ** 2 * sc)).astype(np.float32)
            num += v * w; den += w
    filt = (num / den) * rng + vmin
    out[new_mask] = filt[new_mask]
    return out, new_mask
