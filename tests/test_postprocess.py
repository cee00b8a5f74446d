"""Post-render epilogue: depth -> points / normals.
CPU: the numpy restatement (oracle/post_oracle.py) against outputs of the reference's own Camera class
(tests/golden/py_post.npz).  GPU: the HIP kernels (through the C ABI) against the same fixture and, at 1080p,
against the restatement."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "py_post.npz")
TOL = 2e-5   # float32 torch (batched matmul, unknown contraction) vs our arithmetic, values are O(1..5)


def _mask_eq(a, b):
    inv_a = (a == -1).all(-1); inv_b = (b == -1).all(-1)
    return inv_a, inv_b


def test_numpy_restatement_matches_reference_camera():
    from oracle import post_oracle as po
    z = np.load(GOLD)
    K, E, d = z["intrinsics"], z["extrinsics"], z["depth"]
    assert np.abs(po.depth2point(d, K) - z["points_camera"]).max() < TOL
    assert np.abs(po.depth2point(d, K, E) - z["points_world"]).max() < TOL
    for key, kw in (("normals_camera", {}), ("normals_world", dict(w2c=E)), ("normals_camera_k5", dict(k=5))):
        n = po.depth2normal(d, K, **kw)
        ia, ib = _mask_eq(n, z[key])
        assert np.array_equal(ia, ib), key
        assert np.abs(n - z[key]).max() < 1e-4, key
        assert 0.05 < ia.mean() < 0.6


@pytest.mark.gpu
def test_hip_epilogue_matches_reference_camera_fixture():
    from gaustudio_amd import postprocess as pp
    z = np.load(GOLD)
    K, E = torch.from_numpy(z["intrinsics"]), torch.from_numpy(z["extrinsics"])
    d = torch.from_numpy(z["depth"]).cuda()
    assert np.abs(pp.depth_to_points(d, K).cpu().numpy() - z["points_camera"]).max() < TOL
    assert np.abs(pp.depth_to_points(d, K, E, "world").cpu().numpy() - z["points_world"]).max() < TOL
    for key, kw in (("normals_camera", {}), ("normals_world", dict(extrinsics=E, coordinate="world")),
                    ("normals_camera_k5", dict(k=5))):
        n = pp.depth_to_normals(d, K, **kw).cpu().numpy()
        ia, ib = _mask_eq(n, z[key])
        assert np.array_equal(ia, ib), key
        assert np.abs(n - z[key]).max() < 1e-4, key
    # tap distances beyond the LDS halo-tile path (k=7, 11 -> 3, 5) use the per-pixel kernel: against the numpy restatement
    from oracle import post_oracle as po
    for k in (7, 11):
        n = pp.depth_to_normals(d, K, k=k).cpu().numpy()
        ref = po.depth2normal(z["depth"], z["intrinsics"], k=k)
        ia, ib = _mask_eq(n, ref)
        assert np.array_equal(ia, ib), k
        assert np.abs(n - ref).max() < 1e-4, k


@pytest.mark.gpu
def test_hip_epilogue_on_a_rendered_1080p_depth_and_throughput():
    """The gs-extract-mesh sequence on the C3 frame: render -> median depth -> mask -> world points / normals."""
    from oracle import post_oracle as po
    from gaustudio_amd import postprocess as pp, scenes
    from util import hip_forward, scene_kwargs
    cam = scenes.make_camera(1920, 1080)
    sc = scenes.make_scene(200000, cam, seed=0, sigma_px_median=3.0)
    hs = hip_forward(sc, cam, 0, scene_kwargs(sc, True, False))
    depth = hs["median"][0].clone()
    depth[hs["opacity"][0] < 0.5] = 0.0                                   # extract_mesh.py:104-107
    fx, fy = 1920 / (2 * cam.tanfovx), 1080 / (2 * cam.tanfovy)
    K = torch.tensor([[fx, 0, 960.0], [0, fy, 540.0], [0, 0, 1]])
    E = cam.viewmatrix.t().contiguous()                                   # Camera.extrinsics = world_view_transform^T
    pts = pp.depth_to_points(depth, K, E, "world")
    nrm = pp.depth_to_normals(depth, K, E, coordinate="world")
    dn = depth.cpu().numpy()
    assert np.abs(pts.cpu().numpy() - po.depth2point(dn, K.numpy(), E.numpy())).max() < 1e-4
    ref_n = po.depth2normal(dn, K.numpy(), E.numpy())
    got_n = nrm.cpu().numpy()
    ia, ib = _mask_eq(got_n, ref_n)
    assert (ia != ib).mean() < 1e-5 and np.abs(got_n - ref_n)[~(ia | ib)].max() < 2e-3
    # throughput: 4 B read + 12 B written per pixel
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        pp.depth_to_normals(depth, K, E, coordinate="world")
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 50 * 1e3
    print(f"depth_to_normals 1080p: {us:.1f} us -> {1920 * 1080 * 16 / us / 1e3:.0f} GB/s")


def test_epilogue_rejects_cpu_tensors():
    from gaustudio_amd import postprocess as pp
    with pytest.raises(RuntimeError, match="ROCm devices only"):
        pp.depth_to_points(torch.zeros(4, 4), torch.eye(3))
    with pytest.raises(ValueError, match="Invalid coordinate"):
        pp.depth_to_points(torch.zeros(4, 4), torch.eye(3), coordinate="ndc")


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(2, 3), (63, 7), (64, 8), (65, 9), (130, 17), (257, 33)])
def test_epilogue_at_block_boundaries_against_numpy_restatement(W, H):
    """Image sizes around the 64x8 (normals, LDS halo tile) and 64x4 (points) block shapes, tap distances on both
    kernels' paths, camera and world coordinates -- against oracle/post_oracle.py."""
    from oracle import post_oracle as po
    from gaustudio_amd import postprocess as pp
    rng = np.random.default_rng(W * 1000 + H)
    depth = (2.0 + rng.random((H, W))).astype(np.float32)
    depth[rng.random((H, W)) < 0.1] = 0.0
    K = np.array([[W * 0.9, 0, W / 2], [0, W * 0.9, H / 2], [0, 0, 1]], np.float32)
    a = 0.4
    E = np.eye(4, dtype=np.float32)
    E[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    E[:3, 3] = [0.3, -0.2, 1.0]
    d = torch.from_numpy(depth).cuda()
    Kt, Et = torch.from_numpy(K), torch.from_numpy(E)
    assert np.abs(pp.depth_to_points(d, Kt).cpu().numpy() - po.depth2point(depth, K)).max() < 1e-5
    assert np.abs(pp.depth_to_points(d, Kt, Et, "world").cpu().numpy() - po.depth2point(depth, K, E)).max() < 1e-5
    for k in (1, 3, 5, 7):
        for w2c, kw in ((None, {}), (E, dict(extrinsics=Et, coordinate="world"))):
            n = pp.depth_to_normals(d, Kt, k=k, **kw).cpu().numpy()
            ref = po.depth2normal(depth, K, w2c, k=k)
            ia, ib = _mask_eq(n, ref)
            assert np.array_equal(ia, ib), (k, w2c is not None)
            assert np.abs(n - ref)[~ia].max(initial=0.0) < 2e-4, (k, w2c is not None)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(2, 3), (64, 8), (65, 9), (130, 17), (480, 270)])
def test_fused_epilogue_equals_the_three_separate_steps(W, H):
    """gsr_depth_epilogue (opacity mask + depth2point + depth2normal in one pass) against the separate steps
    -- torch mask, gsr_depth_to_points, gsr_depth_to_normals -- bit for bit, in camera and world coordinates, with and
    without an opacity map, one output or both, for the three tap distances of the tiled path."""
    from gaustudio_amd import postprocess as pp
    rng = np.random.default_rng(W * 7 + H)
    depth = torch.from_numpy((2.0 + rng.random((H, W))).astype(np.float32)).cuda()
    opac = torch.from_numpy(rng.random((H, W)).astype(np.float32)).cuda()
    K = torch.tensor([[W * 0.9, 0, W / 2], [0, W * 0.9, H / 2], [0, 0, 1]], dtype=torch.float32)
    a = 0.4
    E = torch.eye(4)
    E[:3, :3] = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
    E[:3, 3] = torch.tensor([0.3, -0.2, 1.0])
    for coord in ("camera", "world"):
        for use_op in (True, False):
            masked = depth.masked_fill(opac < 0.5, 0.0) if use_op else depth
            for k in (1, 3, 5):
                p_ref = pp.depth_to_points(masked, K, E, coord)
                n_ref = pp.depth_to_normals(masked, K, E, k=k, coordinate=coord)
                p, n = pp.depth_epilogue(depth, K, E, opacity=opac if use_op else None, min_opacity=0.5, k=k, coordinate=coord)
                assert torch.equal(p, p_ref) and torch.equal(n, n_ref), (coord, use_op, k)
    p, n = pp.depth_epilogue(depth, K, E, opacity=opac, want_normals=False)
    assert n is None and torch.equal(p, pp.depth_to_points(depth.masked_fill(opac < 0.5, 0.0), K, E, "world"))
    p, n = pp.depth_epilogue(depth, K, E, opacity=opac, want_points=False)
    assert p is None and torch.equal(n, pp.depth_to_normals(depth.masked_fill(opac < 0.5, 0.0), K, E, coordinate="world"))
    with pytest.raises(RuntimeError):
        pp.depth_epilogue(depth, K, E, k=7)                     # tap distance 3: the separate entry points


def test_masked_bilateral_restatement_on_a_hand_checked_case():
    """extract_pcd.py:185-238 restated (cv2 absent: its dilate / bilateralFilter follow OpenCV's published float32
    algorithm).  With sigma = 75 on a [0, 1]-normalised image the weights are 1 to within 1e-4, so the filter is the
    mean over the centre and its 4-neighbours (circular window of radius 1) of the normalised image in which everything
    outside the eroded mask is 0."""
    from oracle import post_oracle as po
    d = np.arange(25, dtype=np.float32).reshape(5, 5) + 10.0
    m = np.ones((5, 5), np.uint8); m[0, 0] = 0
    out, nm = po.masked_bilateral_filter(d, m)
    want = np.ones((5, 5), bool); want[:2, :2] = False          # the 3x3 windows that contain (0, 0)
    assert np.array_equal(nm, want)
    assert np.array_equal(out[~nm], d[~nm])                      # invalid pixels keep their depth
    vmin, vmax = d[want].min(), d[want].max()
    norm = np.where(want, (d - vmin) / (vmax - vmin), 0.0)
    # interior pixel (2, 2): neighbours (1,2) (3,2) (2,1) (2,3) all valid
    mean = (norm[2, 2] + norm[1, 2] + norm[3, 2] + norm[2, 1] + norm[2, 3]) / 5 * (vmax - vmin) + vmin
    assert abs(out[2, 2] - mean) < 2e-3
    # pixel (2, 1) has the invalidated (1, 1) as a neighbour: it enters the average as 0, as in the reference
    mean = (norm[2, 1] + 0.0 + norm[3, 1] + norm[2, 0] + norm[2, 2]) / 5 * (vmax - vmin) + vmin
    assert abs(out[2, 1] - mean) < 2e-3
    # border pixel (4, 4): reflect-101 brings (3, 4) and (4, 3) in twice
    mean = (norm[4, 4] + 2 * norm[3, 4] + 2 * norm[4, 3]) / 5 * (vmax - vmin) + vmin
    assert abs(out[4, 4] - mean) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,d,sc,ss", [(97, 61, 3, 75.0, 75.0), (640, 360, 5, 0.05, 1.5), (33, 17, 7, 0.2, 3.0), (5, 4, 3, 75.0, 75.0)])
def test_hip_masked_bilateral_matches_the_restatement(W, H, d, sc, ss):
    from gaustudio_amd import postprocess as pp
    from oracle import post_oracle as po
    rng = np.random.default_rng(W * 7 + d)
    depth = (2.0 + 3.0 * rng.random((H, W)) + 0.3 * np.sin(np.arange(W) / 5.0)[None, :]).astype(np.float32)
    mask = rng.random((H, W)) > 0.03
    ref, ref_mask = po.masked_bilateral_filter(depth, mask, d, sc, ss)
    got, got_mask = pp.masked_bilateral_filter(torch.from_numpy(depth).cuda(), torch.from_numpy(mask).cuda(), d, sc, ss)
    assert got_mask.dtype == torch.bool and np.array_equal(got_mask.cpu().numpy(), ref_mask)
    assert np.abs(got.cpu().numpy() - ref).max() < 2e-5
    # all-invalid mask and a constant image come back unchanged
    z, zm = pp.masked_bilateral_filter(torch.from_numpy(depth).cuda(), torch.zeros(H, W, dtype=torch.bool).cuda(), d, sc, ss)
    assert torch.equal(z.cpu(), torch.from_numpy(depth)) and not zm.any()
    c, _ = pp.masked_bilateral_filter(torch.full((H, W), 3.5).cuda(), torch.ones(H, W, dtype=torch.bool).cuda(), d, sc, ss)
    assert torch.equal(c.cpu(), torch.full((H, W), 3.5))
