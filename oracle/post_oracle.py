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
