"""TSDF fusion + marching cubes.  PARITY UNPINNED against vdbfusion (not vendored, not installed):
the HIP path is checked against oracle/tsdf_oracle.c (a restatement of the published VDBFusion algorithm) -- integer
state bit-exact, float mean to 1e-6 -- and the derived marching-cubes tables are checked for orientation and
watertightness.  See DESIGN.md s8."""
import collections
import itertools
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaustudio_amd", "csrc"))
import gen_mc_tables as g  # noqa: E402

from oracle import tsdf_pyoracle as to  # noqa: E402


def _sphere_scan(n=6000, centre=(0.1, 0.2, 1.5), radius=0.5, seed=0, origins=((0, 0, 0), (1.5, 0, 1.5), (-1.2, 0.3, 1.6)),
                 min_cos=0.0):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c = np.array(centre, np.float32)
    pts = (c + radius * d).astype(np.float32)
    out = []
    for o in origins:
        o = np.array(o, np.float32)
        # front-facing part of the sphere (min_cos > 0: drop grazing hits, where a projective SDF is noisy)
        vis = ((o - pts) * (pts - c)).sum(1) > min_cos * np.linalg.norm(o - pts, axis=1) * radius
        out.append((pts[vis], o))
    return out


def _directed_edges(tris):
    E = collections.Counter()
    for a, b, c in tris.tolist():
        for x, y in ((a, b), (b, c), (c, a)):
            E[(x, y)] += 1
    return E


# ---------------------------------------------------------------------------------------------- CPU: tables, oracle

def test_mc_tables_header_is_current():
    path = os.path.join(ROOT, "gaustudio_amd", "csrc", "gsr_mc_tables.h")
    assert open(path).read() == g.header_text(), "run gaustudio_amd/csrc/gen_mc_tables.py"


def test_mc_tables_use_exactly_the_crossing_edges_and_point_outwards():
    table, mask = g.build()
    C = np.array(g.CORNERS, float)
    assert sum(len(t) for t in table) == 820 and max(len(t) for t in table) == 5

    def grad(f, p):
        x, y, z = p
        out = np.zeros(3)
        for i, (cx, cy, cz) in enumerate(g.CORNERS):
            wx, wy, wz = (x if cx else 1 - x), (y if cy else 1 - y), (z if cz else 1 - z)
            out += f[i] * np.array([(1 if cx else -1) * wy * wz, wx * (1 if cy else -1) * wz, wx * wy * (1 if cz else -1)])
        return out

    for case in range(256):
        f = np.array([-1.0 if (case >> i) & 1 else 1.0 for i in range(8)])
        cross = {e for e, (a, b) in enumerate(g.EDGES) if f[a] * f[b] < 0}
        used = {e for t in table[case] for e in t}
        assert used == cross and mask[case] == sum(1 << e for e in cross)
        V = {e: (C[a] + C[b]) / 2 for e, (a, b) in enumerate(g.EDGES)}
        for t in table[case]:
            a, b, c = (V[e] for e in t)
            assert np.cross(b - a, c - a) @ grad(f, (a + b + c) / 3) > 1e-9, (case, t)   # normal towards tsdf > 0


def test_mc_tables_give_closed_consistently_oriented_surfaces():
    table, _ = g.build()
    N = 14
    xs = np.linspace(-1, 1, N)
    X, Y, Z = np.meshgrid(xs, xs, xs, indexing="ij")
    for trial in range(2):
        F = np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 0.62 + 0.3 * np.sin(3 * X + trial) * np.cos(4 * Y) * np.sin(2 * Z + 1)
        E = collections.Counter()
        for i, j, k in itertools.product(range(N - 1), repeat=3):
            f = [F[i + cx, j + cy, k + cz] for cx, cy, cz in g.CORNERS]
            case = sum(1 << c for c in range(8) if f[c] < 0)
            for t in table[case]:
                vid = []
                for e in t:
                    a, b = g.EDGES[e]
                    pa = tuple(np.add((i, j, k), g.CORNERS[a])); pb = tuple(np.add((i, j, k), g.CORNERS[b]))
                    vid.append((min(pa, pb), max(pa, pb)))
                for q in range(3):
                    E[(vid[q], vid[(q + 1) % 3])] += 1
        assert E and all(c == 1 and E.get((b, a), 0) == 1 for (a, b), c in E.items())


def test_oracle_wall_scan_zero_crossing_and_running_average():
    """A wall at z = 1 scanned from the origin: tsdf > 0 in front, < 0 behind, band limited to +-sdf_trunc; the
    sequential float running average (vdbfusion's update) equals the mean of the fixed-point sums to 1e-6."""
    ys, xs = np.meshgrid(np.linspace(-0.3, 0.3, 61), np.linspace(-0.3, 0.3, 61), indexing="ij")
    pts = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)], 1).astype(np.float32)
    v = to.Volume(0.02, 0.08)
    v.integrate(pts, np.zeros(3, np.float32))
    c, t, w, s = v.export()
    zc = (c[:, 2].astype(np.float64) + 0.5) * 0.02
    assert (t[zc < 0.95] > 0).all() and (t[zc > 1.05] < 0).all()
    assert zc.min() > 1 - 0.08 - 0.04 and zc.max() < 1 + 0.08 + 0.04 and np.abs(t).max() <= 0.08 + 1e-7
    assert np.abs(s / w * (0.08 / 2 ** 15) - t).max() < 2e-6          # 2^-15 fixed point: 1.2e-6 per observation
    V, T = to.extract_mesh(c, w, s, 0.02, 0.08, min_weight=1)
    assert len(T) > 500 and np.abs(V[:, 2] - 1.0).max() < 0.012          # the wall, to half a voxel
    n = np.cross(V[T[:, 1]] - V[T[:, 0]], V[T[:, 2]] - V[T[:, 0]])
    assert (n[:, 2] < 0).all()                                           # facing the sensor (towards tsdf > 0)


# ---------------------------------------------------------------------------------------------- GPU

def _gpu_volume(vs, tr, scans, capacity=1 << 12, **kw):
    from gaustudio_amd.tsdf import TSDFVolume
    vol = TSDFVolume(vs, tr, capacity_blocks=capacity, **kw)
    for pts, o in scans:
        vol.integrate(torch.from_numpy(pts).cuda(), o)
    return vol


@pytest.mark.gpu
@pytest.mark.parametrize("carve", [False, True])
def test_integrate_matches_oracle_exactly(carve):
    scans = _sphere_scan()
    ov = to.Volume(0.05, 0.2, carve)
    for pts, o in scans:
        ov.integrate(pts, o)
    oc, ot, ow, os_ = ov.export()
    vol = _gpu_volume(0.05, 0.2, scans, capacity=1 << 14, space_carving=carve)
    c, t, w, s = [x.cpu().numpy() for x in vol.export_voxels()]
    assert np.array_equal(c, oc) and np.array_equal(w, ow) and np.array_equal(s, os_)      # integer state: bit-exact
    # mean of the exact fixed-point sum vs vdbfusion's sequential float running average: the latter drifts by up to
    # ~one float ulp of sdf_trunc per observation (hundreds of observations per voxel with space carving)
    # (+ half a fixed-point step, sdf_trunc / 2^16, of quantisation)
    assert (np.abs(t - ot) <= 1e-7 + 0.2 / 2 ** 16 + 2.0 * ow * np.float32(2.0 ** -23) * 0.2).all()


@pytest.mark.gpu
def test_masked_pixels_at_the_sensor_origin_need_no_compaction():
    """gs-extract-mesh compacts `points[~invalid_mask]` before vdbfusion (extract_mesh.py:110).  depth2point maps a masked
    pixel (depth 0) onto the sensor origin, and the integrate kernel skips zero-length rays: a whole point map with the
    invalid pixels left in gives the very same volume as the compacted one -- also when a workgroup's FIRST point is such a
    pixel (the local aggregation window is centred on the first active point)."""
    scans = _sphere_scan(n=30000, seed=5)
    compacted = _gpu_volume(0.05, 0.2, scans, capacity=1 << 14)
    rng = np.random.default_rng(2)
    padded = []
    for pts, o in scans:
        keep = rng.random(len(pts) * 2) < 0.5                      # half of the "pixels" are masked
        keep[::256] = False                                          # in particular every workgroup's first one
        # masked pixels sit at the origin up to the rounding of an unprojection (1e-6 of a unit here), not exactly on it
        full = np.tile(np.asarray(o, np.float32), (len(keep), 1)) + rng.uniform(-1e-6, 1e-6, (len(keep), 3)).astype(np.float32)
        idx = np.nonzero(keep)[0][:len(pts)]
        full[idx] = pts[:len(idx)]
        padded.append((full, o))
        # compare against integrating exactly the kept points
    ref = _gpu_volume(0.05, 0.2, [(f[np.linalg.norm(f - np.asarray(o, np.float32), axis=1) > 1e-4], o) for f, o in padded], capacity=1 << 14)
    got = _gpu_volume(0.05, 0.2, padded, capacity=1 << 14)
    for x, y in zip(ref.export_voxels(), got.export_voxels()):
        assert torch.equal(x, y)
    assert compacted.export_voxels()[0].shape[0] > 0


@pytest.mark.gpu
def test_masked_pixels_far_from_the_world_origin_with_small_voxels():
    """a sensor at coordinates of a few hundred units and 2-cm voxels -- the unprojected position of a masked pixel
    differs from `origin` by the rounding of the coordinates (3e-5 here), far more than a thousandth of a voxel (2e-5); the
    skip threshold also scales with |origin|, so such pixels still carve nothing."""
    from gaustudio_amd.tsdf import TSDFVolume
    rng = np.random.default_rng(11)
    o = np.array([310.0, -120.0, 45.0], np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (o + d * 2.0).astype(np.float32)
    masked = (o + rng.uniform(-3e-5, 3e-5, (20000, 3))).astype(np.float32)          # ~1 ulp of 310 is 3e-5
    vols = []
    for cloud in (pts, np.concatenate([masked[:64], pts, masked[64:]])):
        v = TSDFVolume(voxel_size=0.02, sdf_trunc=0.08, space_carving=False, device="cuda", capacity_blocks=1 << 16)
        v.integrate(torch.from_numpy(cloud).cuda(), torch.from_numpy(o).cuda())
        vols.append(v.export_voxels())
    for x, y in zip(*vols):
        assert torch.equal(x, y)
    assert vols[0][0].shape[0] > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(64, 96), (37, 53), (32, 32), (1, 300), (200, 7), (270, 480)])
def test_point_map_in_patches_gives_the_same_volume_as_the_flat_list(H, W):
    """integrate([H,W,3]) -- workgroups take 32 x 32 patches of the map (gsr_tsdf_integrate_map) -- against
    integrate([H*W,3]): identical volumes (integer adds commute), ragged edges and masked pixels (points at the origin)
    included."""
    from gaustudio_amd.tsdf import TSDFVolume
    rng = np.random.default_rng(H * 1000 + W)
    o = np.array([0.1, -0.2, 0.3], np.float32)
    u, v = np.meshgrid(np.linspace(-0.6, 0.6, W, dtype=np.float32), np.linspace(-0.4, 0.4, H, dtype=np.float32))
    d = np.stack([u, v, np.ones_like(u)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    depth = (2.0 + 0.3 * np.sin(7 * u) * np.cos(5 * v) + rng.uniform(0, 0.01, u.shape)).astype(np.float32)
    pts = o + d * depth[..., None]
    pts[rng.random((H, W)) < 0.2] = o                        # masked pixels
    pmap = torch.from_numpy(pts.astype(np.float32)).cuda()
    a = TSDFVolume(0.03, 0.12, capacity_blocks=1 << 13)
    a.integrate(pmap.reshape(-1, 3), o)
    b = TSDFVolume(0.03, 0.12, capacity_blocks=1 << 13)
    b.integrate(pmap, o)
    xa, xb = a.export_voxels(), b.export_voxels()
    assert xa[0].shape[0] > 0
    for x, y in zip(xa, xb):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_integrate_is_order_independent_and_deterministic():
    scans = _sphere_scan(n=20000, seed=3)
    a = _gpu_volume(0.02, 0.08, scans, capacity=1 << 14)
    rng = np.random.default_rng(1)
    shuffled = [(pts[rng.permutation(len(pts))], o) for pts, o in reversed(scans)]
    split = []
    for pts, o in shuffled:
        split += [(pts[: len(pts) // 3], o), (pts[len(pts) // 3:], o)]
    b = _gpu_volume(0.02, 0.08, split, capacity=1 << 14)
    for x, y in zip(a.export_voxels(), b.export_voxels()):
        assert torch.equal(x, y)
    va, ta = a.extract_triangle_mesh_device(min_weight=1)
    vb, tb = b.extract_triangle_mesh_device(min_weight=1)
    assert torch.equal(va, vb) and torch.equal(ta, tb)        # blocks are emitted in key order, not hash order


@pytest.mark.gpu
@pytest.mark.parametrize("min_weight,fill_holes", [(1, True), (3, True), (0.5, False)])
def test_mesh_matches_oracle_marching_cubes(min_weight, fill_holes):
    scans = _sphere_scan(n=5000, seed=5)
    vol = _gpu_volume(0.05, 0.2, scans)
    c, t, w, s = [x.cpu().numpy() for x in vol.export_voxels()]
    V, T = vol.extract_triangle_mesh(fill_holes=fill_holes, min_weight=min_weight)
    _, bc = vol.occupied_blocks()
    oV, oT = to.extract_mesh(c, w, s, 0.05, 0.2, min_weight=min_weight, fill_holes=fill_holes,
                             blocks={tuple(b) for b in bc.cpu().numpy().tolist()})
    assert len(T) > 1000 and V.dtype == np.float64 and T.dtype == np.int32

    def canon(Vx, Tx):
        tri = Vx.astype(np.float32)[Tx]                           # [nt,3,3]
        out = set()
        for a in tri:
            rows = [tuple(r) for r in a.tolist()]
            k = rows.index(min(rows))
            out.add(tuple(rows[k:] + rows[:k]))                   # rotation-invariant, orientation-preserving
        return out

    assert canon(V, T) == canon(oV, oT)
    # shared vertices: one per crossing edge (positions can coincide only when a corner value is exactly 0)
    assert len(V) == len(oV) and len(V) - len(np.unique(V.astype(np.float32), axis=0)) <= 8 and len(V) < 0.7 * len(T)


def _depth_camera_scans(n_views=48, res=160, radius=0.5, dist=2.0, min_cos=0.6):
    """Dense pinhole depth maps of a sphere at the origin from a Fibonacci sphere of viewpoints (ray spacing on the
    surface ~4 mm): the coverage a rendered depth map gives, unlike scattered samples.  Grazing hits are dropped."""
    scans = []
    golden = np.pi * (3.0 - np.sqrt(5.0))
    for k in range(n_views):
        z = 1 - 2 * (k + 0.5) / n_views
        rho = np.sqrt(1 - z * z)
        o = dist * np.array([rho * np.cos(golden * k), rho * np.sin(golden * k), z])
        fwd = -o / dist
        up = np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        half = 1.15 * radius / np.sqrt(dist * dist - radius * radius)              # tan of the half field of view
        u, v = np.meshgrid(np.linspace(-half, half, res), np.linspace(-half, half, res))
        d = fwd[None, None] + u[..., None] * right + v[..., None] * up
        d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).reshape(-1, 3)
        b = d @ o
        disc = b * b - (dist * dist - radius * radius)
        hit = disc > 0
        t = -b[hit] - np.sqrt(disc[hit])
        pts = o + t[:, None] * d[hit]
        keep = -(d[hit] * pts).sum(1) / radius > min_cos
        scans.append((pts[keep].astype(np.float32), o.astype(np.float32)))
    return scans


@pytest.mark.gpu
def test_full_coverage_gives_a_watertight_sphere():
    scans = _depth_camera_scans()
    vol = _gpu_volume(0.02, 0.06, scans, capacity=1 << 15)
    V, T = vol.extract_triangle_mesh(min_weight=1)
    E = _directed_edges(T)
    assert all(c == 1 and E.get((b, a), 0) == 1 for (a, b), c in E.items())       # closed 2-manifold, consistent winding
    assert len(V) - len(E) // 2 + len(T) == 2                                       # Euler characteristic of a sphere
    r = np.linalg.norm(V, axis=1)
    assert abs(r.mean() - 0.5) < 0.004 and r.min() > 0.47 and r.max() < 0.53
    n = np.cross(V[T[:, 1]] - V[T[:, 0]], V[T[:, 2]] - V[T[:, 0]])
    assert ((n * V[T].mean(1)).sum(1) > 0).all()                                    # outward normals


@pytest.mark.gpu
def test_hash_overflow_is_reported_and_cpu_is_refused():
    from gaustudio_amd.tsdf import TSDFVolume
    scans = _sphere_scan()
    vol = _gpu_volume(0.01, 0.04, scans, capacity=16)
    with pytest.raises(RuntimeError, match="overflowed"):
        vol.extract_triangle_mesh(min_weight=1)
    with pytest.raises(RuntimeError, match="ROCm"):
        TSDFVolume(0.01, 0.04, device="cpu")
    empty = TSDFVolume(0.01, 0.04, capacity_blocks=16)
    V, T = empty.extract_triangle_mesh()
    assert V.shape == (0, 3) and T.shape == (0, 3)


@pytest.mark.gpu
def test_render_to_mesh_pipeline_stays_on_the_gpu():
    """The gs-extract-mesh loop (extract_mesh.py:95-145) on a synthetic shell of Gaussians: render -> median depth ->
    mask -> world points -> integrate (device tensors throughout) -> mesh."""
    from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer, postprocess as pp, scenes
    from gaustudio_amd.tsdf import TSDFVolume
    g_ = torch.Generator().manual_seed(0)
    P = 60000
    d = torch.randn(P, 3, generator=g_)
    d = d / d.norm(dim=1, keepdim=True)
    dev = torch.device("cuda:0")
    means = (d * 1.0).to(dev)                                                 # unit sphere shell
    scales = torch.full((P, 3), 0.012, device=dev)
    rots = torch.tensor([[1.0, 0, 0, 0]], device=dev).repeat(P, 1)
    opac = torch.full((P, 1), 0.95, device=dev)
    cols = torch.rand(P, 3, generator=g_).to(dev)
    vol = TSDFVolume(0.02, 0.08, capacity_blocks=1 << 15)
    for cam in scenes.ring_cameras(12, 320, 240, radius=3.0, elevation=0.3) + scenes.ring_cameras(6, 320, 240, radius=3.0, elevation=-0.9):
        rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                           cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 0, cam.campos.to(dev), False, False)
        with torch.no_grad():
            _, _, _, median, opacity = GaussianRasterizer(rs)(means3D=means, means2D=torch.zeros_like(means), opacities=opac,
                                                               colors_precomp=cols, scales=scales, rotations=rots)
        depth = median[0].clone()
        invalid = opacity[0] < 0.5
        depth[invalid] = 0
        f = cam.width / (2 * cam.tanfovx)
        K = torch.tensor([[f, 0, cam.width / 2], [0, f, cam.height / 2], [0, 0, 1]])
        E = cam.viewmatrix.t().contiguous()
        pts = pp.depth_to_points(depth, K, E, "world")[~invalid]
        vol.integrate(pts, cam.campos)
    V, T = vol.extract_triangle_mesh(min_weight=2)
    assert len(T) > 20000
    r = np.linalg.norm(V, axis=1)
    assert abs(np.median(r) - 1.0) < 0.03 and (np.abs(r - 1.0) < 0.1).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("seed,vs,tr,carve", [(0, 0.013, 0.05, False), (1, 0.1, 0.3, False), (2, 0.05, 0.11, True), (3, 0.02, 0.02, False)])
def test_integrate_random_clouds_negative_coordinates_and_degenerate_points(seed, vs, tr, carve):
    """Unstructured points around the origin (all eight octants of voxel / block coordinates), several origins, plus
    degenerate inputs -- a point equal to the origin, axis-parallel rays, NaN / inf points -- bit-exact integer state."""
    rng = np.random.default_rng(seed)
    scans = []
    for _ in range(3):
        o = rng.normal(size=3).astype(np.float32) * 0.7
        pts = (rng.random((3000, 3)) * 2 - 1).astype(np.float32)
        pts[0] = o                                            # zero-length ray: skipped
        pts[1] = o + np.array([0.5, 0, 0], np.float32)        # axis-parallel: two DDA axes never step
        pts[2] = o + np.array([0, 0, -0.37], np.float32)
        pts[3] = np.array([np.nan, 0, 0], np.float32)
        pts[4] = np.array([np.inf, 1, 1], np.float32)
        scans.append((pts, o))
    ov = to.Volume(vs, tr, carve)
    for pts, o in scans:
        ov.integrate(pts, o)
    oc, ot, ow, os_ = ov.export()
    vol = _gpu_volume(vs, tr, scans, capacity=1 << 16, space_carving=carve)
    c, t, w, s = [x.cpu().numpy() for x in vol.export_voxels()]
    assert (c < 0).any() and (c > 0).any()
    assert np.array_equal(c, oc) and np.array_equal(w, ow) and np.array_equal(s, os_)
