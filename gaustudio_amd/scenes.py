"""Synthetic scenes and cameras for tests and bench.py.

Camera matrices follow the conventions the op consumes from gaustudio's Camera dataclass
(/root/reference/gaustudio/datasets/__init__.py:52-104 getWorld2View2/getProjectionMatrix,
:154-159 world_view_transform = W2C^T, full_proj_transform = view @ proj, :182-183 camera_center).
Everything is generated on the CPU from a seeded torch.Generator so that the same bytes can be fed
to the HIP path and to the oracle.
"""
import math
from typing import NamedTuple, Optional

import numpy as np
import torch


class Cam(NamedTuple):
    width: int
    height: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor   # [4,4] = W2C^T (flat float[16] is column-major W2C)
    projmatrix: torch.Tensor   # [4,4] = viewmatrix @ P^T
    campos: torch.Tensor       # [3]


def projection_matrix(znear, zfar, fovx, fovy):
    """OpenGL-style projection with z_sign=+1, centred principal point (datasets/__init__.py:66-104)."""
    t = math.tan(fovy / 2) * znear
    r = math.tan(fovx / 2) * znear
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2.0 * znear / (2 * r)
    P[1, 1] = 2.0 * znear / (2 * t)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(width, height, fovx_deg=60.0, R=None, T=None, znear=0.1, zfar=100.0) -> Cam:
    """R is the camera-to-world rotation, T the world-to-camera translation (3DGS convention,
    datasets/__init__.py:52-64: W2C[:3,:3] = R^T, W2C[:3,3] = T)."""
    fovx = math.radians(fovx_deg)
    tanfovx = math.tan(fovx / 2)
    tanfovy = tanfovx * height / width
    fovy = 2 * math.atan(tanfovy)
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    T = np.zeros(3) if T is None else np.asarray(T, dtype=np.float64)
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.T
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    view = torch.tensor(np.float32(Rt)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    campos = torch.inverse(view)[3, :3].contiguous()
    return Cam(width, height, tanfovx, tanfovy, view, full, campos)


def look_at_camera(width, height, eye, target, fovx_deg=60.0, up=(0.0, -1.0, 0.0)) -> Cam:
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)     # camera-to-world, columns = camera axes (x right, y down, z fwd)
    T = -R.T @ eye
    return make_camera(width, height, fovx_deg, R=R, T=T)


class Scene(NamedTuple):
    means3D: torch.Tensor     # [P,3]
    scales: torch.Tensor      # [P,3]  post-activation
    rotations: torch.Tensor   # [P,4]  normalised (r,x,y,z)
    opacities: torch.Tensor   # [P,1]  post-activation
    shs: torch.Tensor         # [P,16,3]


def make_scene(P, cam: Cam, seed=0, sigma_px_median=1.5, sigma_px_logstd=0.6, zmin=2.0, zmax=20.0) -> Scene:
    """SURVEY.md s8d generator for a camera at the origin looking down +z."""
    g = torch.Generator().manual_seed(seed)
    z = torch.rand(P, generator=g) * (zmax - zmin) + zmin
    ux = (torch.rand(P, generator=g) * 2 - 1) * 1.1
    uy = (torch.rand(P, generator=g) * 2 - 1) * 1.1
    means = torch.stack([z * cam.tanfovx * ux, z * cam.tanfovy * uy, z], dim=1)
    sigma_px = torch.exp(torch.randn(P, generator=g) * sigma_px_logstd + math.log(sigma_px_median))
    sigma = sigma_px * z * 2 * cam.tanfovx / cam.width
    scales = sigma[:, None] * (torch.rand(P, 3, generator=g) * 1.5 + 0.5)
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 2)
    shs = torch.randn(P, 16, 3, generator=g) * 0.1
    shs[:, 0, :] = (torch.rand(P, 3, generator=g) * 2 - 1) / 0.28209479177387814
    return Scene(means.float().contiguous(), scales.float().contiguous(), rot.float().contiguous(),
                 opac.float().contiguous(), shs.float().contiguous())


def make_ball_scene(P, radius=4.0, seed=0, sigma=0.02) -> Scene:
    """Gaussians scattered in a ball around the world origin, for inward-looking ring cameras
    (BASELINE config C4/C5 stand-in)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(P, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    r = radius * torch.rand(P, 1, generator=g) ** (1.0 / 3.0)
    means = d * r
    scales = sigma * torch.exp(torch.randn(P, 1, generator=g) * 0.6) * (torch.rand(P, 3, generator=g) * 1.5 + 0.5)
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 2)
    shs = torch.randn(P, 16, 3, generator=g) * 0.1
    shs[:, 0, :] = (torch.rand(P, 3, generator=g) * 2 - 1) / 0.28209479177387814
    return Scene(means.float().contiguous(), scales.float().contiguous(), rot.float().contiguous(),
                 opac.float().contiguous(), shs.float().contiguous())


def make_clustered_scene(P, width, cam_distance=11.0, fovx_deg=60.0, radius=3.0, clusters=48, seed=0, sigma_px_median=1.5,
                         sigma_px_logstd=0.7, big_fraction=0.01, big_px=(50.0, 60.0)) -> Scene:
    """A NON-UNIFORM scene for inward-looking ring cameras at `cam_distance` (the structure of a trained capture rather than
    of SURVEY.md s8d's uniform cloud): Gaussians in `clusters` blobs inside a ball of `radius` -- blob sizes log-uniform over
    1.5 decades, populations Zipf-like, so a few screen regions hold most of the instances --, nothing outside the ball (a
    ring camera sees well over 20 % empty tiles), footprints log-normal with a `big_fraction` tail of splats whose sigma
    is `big_px` pixels at the ball's distance (the floaters / background blobs of real scenes: every one of them sits in
    hundreds of tile lists)."""
    g = torch.Generator().manual_seed(seed)
    cdir = torch.randn(clusters, 3, generator=g)
    cdir = cdir / cdir.norm(dim=1, keepdim=True)
    ccen = cdir * (radius * 0.85 * torch.rand(clusters, 1, generator=g) ** (1.0 / 3.0))
    crad = radius * 10 ** (-1.7 + 1.5 * torch.rand(clusters, generator=g))             # 0.02 .. 0.63 of the ball
    w = 1.0 / torch.arange(1, clusters + 1, dtype=torch.float32) ** 0.9                   # Zipf-like populations
    which = torch.multinomial(w / w.sum(), P, replacement=True, generator=g)
    means = ccen[which] + torch.randn(P, 3, generator=g) * (crad[which][:, None] * 0.5)
    far = means.norm(dim=1) > radius
    means[far] = means[far] / means[far].norm(dim=1, keepdim=True) * radius
    px_to_world = cam_distance * 2 * math.tan(math.radians(fovx_deg) / 2) / width          # one pixel at the ball's distance
    sigma_px = torch.exp(torch.randn(P, generator=g) * sigma_px_logstd + math.log(sigma_px_median))
    big = torch.rand(P, generator=g) < big_fraction
    sigma_px[big] = big_px[0] + (big_px[1] - big_px[0]) * torch.rand(int(big.sum()), generator=g)
    jitter = torch.rand(P, 3, generator=g) * 1.5 + 0.5
    jitter[big] = jitter[big] / 2.0                  # big splats: `big_px` is the sigma of their LARGEST axis (0.25 .. 1 of it on the others)
    jitter[big, 0] = 1.0
    scales = (sigma_px * px_to_world)[:, None] * jitter
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 2)
    opac[big] = opac[big] * 0.3                                                            # large splats are faint in trained scenes
    shs = torch.randn(P, 16, 3, generator=g) * 0.1
    shs[:, 0, :] = (torch.rand(P, 3, generator=g) * 2 - 1) / 0.28209479177387814
    return Scene(means.float().contiguous(), scales.float().contiguous(), rot.float().contiguous(),
                 opac.float().contiguous(), shs.float().contiguous())


def ring_cameras(n, width, height, radius=10.0, fovx_deg=60.0, elevation=0.2):
    cams = []
    for k in range(n):
        a = 2 * math.pi * k / n
        eye = (radius * math.cos(a), -radius * elevation, radius * math.sin(a))
        cams.append(look_at_camera(width, height, eye, (0.0, 0.0, 0.0), fovx_deg))
    return cams


def make_output_grads(cam: Cam, seed=1):
    """Upstream gradients for the backward pass."""
    g = torch.Generator().manual_seed(seed)
    H, W = cam.height, cam.width
    return (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g) * 0.1,
            torch.randn(3, H, W, generator=g) * 0.1, torch.randn(1, H, W, generator=g) * 0.1)
