"""-m gpu: behaviour of the operator interface on a real device (SURVEY.md s8b contracts that need a GPU)."""
import os

import numpy as np
import pytest
import torch

from gaustudio_amd import scenes

from util import hip_backward_raw, hip_forward, scene_kwargs, to_np

pytestmark = pytest.mark.gpu


def _settings(cam, D=3, dev="cuda", bg=None, debug=False, mod=1.0):
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy,
                                         torch.zeros(3) if bg is None else bg, mod, cam.viewmatrix.to(dev),
                                         cam.projmatrix.to(dev), D, cam.campos.to(dev), False, debug)


def _render(sc, cam, rs, dev="cuda", leaves=None):
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizer
    if leaves is None:
        leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    out = GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                 shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
    return out, leaves, means2D


def test_two_forwards_then_backward_of_the_first_and_retain_graph():
    """The opaque buffers belong to each call (ctx.save_for_backward); a later forward must not disturb an
    earlier graph, and backward may run twice with retain_graph."""
    camA, camB = scenes.make_camera(160, 96), scenes.make_camera(96, 160)
    sc = scenes.make_scene(4000, camA, seed=3, sigma_px_median=2.0)
    (cA, rA, dA, mA, oA), leaves, _ = _render(sc, camA, _settings(camA))
    (cB, *_), _, _ = _render(sc, camB, _settings(camB), leaves=leaves)
    g = torch.autograd.grad(cA.sum() + dA.sum(), list(leaves.values()), retain_graph=True)
    g2 = torch.autograd.grad(cA.sum() + dA.sum(), list(leaves.values()))
    for a, b in zip(g, g2):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * float(a.abs().max()))   # 2-wave LDS meeting is unordered
    cB.sum().backward()
    assert all(v.grad is not None and torch.isfinite(v.grad).all() for v in leaves.values())


def test_no_grad_and_non_contiguous_inputs():
    cam = scenes.make_camera(128, 80)
    sc = scenes.make_scene(3000, cam, seed=9)
    ref = hip_forward(sc, cam, 3, scene_kwargs(sc, True, False))
    dev = "cuda"
    big = torch.zeros(3000, 6, device=dev)
    big[:, ::2] = sc.means3D.to(dev)
    means_nc = big[:, ::2]                                      # stride-2 view, same values
    assert not means_nc.is_contiguous()
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizer
    with torch.no_grad():
        color, radii, depth, median, opac = GaussianRasterizer(_settings(cam))(
            means3D=means_nc, means2D=torch.zeros(3000, 3, device=dev), opacities=sc.opacities.to(dev),
            shs=sc.shs.to(dev), scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
    assert torch.equal(color, ref["color"]) and torch.equal(radii, ref["radii"]) and not color.requires_grad


def test_bg_on_device_or_host_and_viewmatrix_on_host():
    cam = scenes.make_camera(96, 64)
    sc = scenes.make_scene(1500, cam, seed=2)
    kw = scene_kwargs(sc, True, False)
    a = hip_forward(sc, cam, 2, kw, bg=torch.ones(3))           # CPU bg, as gaustudio's renderers pass it
    b = hip_forward(sc, cam, 2, kw, bg=torch.ones(3).cuda())
    assert torch.equal(a["color"], b["color"])
    from gaustudio_amd import _C
    e = torch.Tensor([])
    out = _C.rasterize_gaussians(torch.zeros(3), sc.means3D.cuda(), e, sc.opacities.cuda(), sc.scales.cuda(),
                                 sc.rotations.cuda(), 1.0, e, cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy,
                                 cam.height, cam.width, sc.shs.cuda(), 2, cam.campos, False, False)   # host camera tensors
    assert torch.equal(out[1], a["color"])


def test_debug_mode_syncs_per_stage_and_dumps_snapshot_on_failure(tmp_path, monkeypatch):
    cam = scenes.make_camera(64, 64)
    sc = scenes.make_scene(800, cam, seed=5)
    (color, *_), leaves, _ = _render(sc, cam, _settings(cam, debug=True))
    color.sum().backward()
    assert torch.isfinite(leaves["means3D"].grad).all()
    # failure path: SH degree 3 with only 4 stored coefficients is rejected by the library -> snapshot_fw.dump
    monkeypatch.chdir(tmp_path)
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizer
    with pytest.raises(RuntimeError, match="SH degree"):
        GaussianRasterizer(_settings(cam, D=3, debug=True))(
            means3D=sc.means3D.cuda(), means2D=torch.zeros(800, 3, device="cuda"), opacities=sc.opacities.cuda(),
            shs=sc.shs[:, :4].contiguous().cuda(), scales=sc.scales.cuda(), rotations=sc.rotations.cuda())
    assert os.path.exists(tmp_path / "snapshot_fw.dump")
    snap = torch.load(tmp_path / "snapshot_fw.dump")
    assert isinstance(snap, tuple) and snap[1].shape == (800, 3) and snap[1].device.type == "cpu"


def test_wrong_dtype_and_shape_errors():
    from gaustudio_amd import _C
    cam = scenes.make_camera(32, 32)
    e = torch.Tensor([])
    args = lambda means, col: (torch.zeros(3), means, col, torch.ones(4, 1).cuda(), torch.ones(4, 3).cuda(),
                               torch.ones(4, 4).cuda(), 1.0, e, cam.viewmatrix.cuda(), cam.projmatrix.cuda(), cam.tanfovx,
                               cam.tanfovy, 32, 32, e, 0, cam.campos.cuda(), False, False)
    with pytest.raises(RuntimeError, match="float32"):
        _C.rasterize_gaussians(*args(torch.zeros(4, 3, dtype=torch.float64).cuda(), torch.zeros(4, 3).cuda()))
    with pytest.raises(RuntimeError, match=r"colors_precomp must have dimensions \(num_points, 3\)"):
        _C.rasterize_gaussians(*args(torch.zeros(4, 3).cuda(), torch.zeros(4, 4).cuda()))
    with pytest.raises(RuntimeError, match="provide precomputed Gaussian colors"):        # rasterizer_impl.cu:245-248
        _C.rasterize_gaussians(*args(torch.zeros(4, 3).cuda(), e))


def test_two_dimensional_scales_padded_like_vanilla_renderer():
    """vanilla_renderer.py:38-39 pads 2-D scales with 1e-7; degenerate (flat) Gaussians must render finitely."""
    cam = scenes.make_camera(96, 96)
    sc = scenes.make_scene(1000, cam, seed=4, sigma_px_median=3.0)
    s2 = torch.cat([sc.scales[:, :2], torch.zeros_like(sc.scales[:, :1]) + 1e-7], dim=-1)
    sc = sc._replace(scales=s2.contiguous())
    hs = hip_forward(sc, cam, 3, scene_kwargs(sc, True, False))
    assert torch.isfinite(hs["color"]).all() and hs["num_rendered"] > 0


def test_streams_follow_torch_current_stream():
    cam = scenes.make_camera(128, 128)
    sc = scenes.make_scene(5000, cam, seed=6)
    kw = scene_kwargs(sc, True, False)
    ref = hip_forward(sc, cam, 3, kw)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        got = hip_forward(sc, cam, 3, kw)
    s.synchronize()
    assert torch.equal(ref["color"], got["color"])


def test_profiling_stage_times_reported():
    from gaustudio_amd import _C
    cam = scenes.make_camera(256, 256)
    sc = scenes.make_scene(20000, cam, seed=7)
    _C.set_profiling(True)
    for _ in range(3):
        hip_forward(sc, cam, 0, scene_kwargs(sc, True, False))
    ms = _C.last_forward_ms()
    _C.set_profiling(False)
    assert ms is not None and ms["calls"] == 3 and ms["composite"] > 0 and ms["preprocess"] > 0
    assert _C.last_forward_ms() is None


def test_instance_count_beyond_int32_is_an_error_not_a_wraparound():
    """150 k Gaussians that each cover all 32 400 tiles of a 3840x2160 frame are 4.86e9 instances: more than the
    reference's `int num_rendered` can hold and past a 32-bit wrap (it would come back as ~5.7e8)."""
    from gaustudio_amd import _C
    cam = scenes.make_camera(3840, 2160)
    P = 150_000
    g = torch.Generator().manual_seed(1)
    means = torch.cat([torch.randn(P, 2, generator=g) * 0.05, 5.0 + torch.rand(P, 1, generator=g)], dim=1).cuda()
    e = torch.Tensor([])
    with pytest.raises(RuntimeError, match="instances"):
        _C.rasterize_gaussians(torch.zeros(3), means, torch.rand(P, 3, generator=g).cuda(), torch.full((P, 1), 0.01).cuda(),
                               torch.full((P, 3), 4.0).cuda(), torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1).cuda(), 1.0, e,
                               cam.viewmatrix.cuda(), cam.projmatrix.cuda(), cam.tanfovx, cam.tanfovy, 2160, 3840, e, 0,
                               cam.campos.cuda(), False, False)
    # the library is usable afterwards
    out = _C.rasterize_gaussians(torch.zeros(3), means[:10], torch.rand(10, 3).cuda(), torch.full((10, 1), 0.5).cuda(),
                                 torch.full((10, 3), 0.01).cuda(), torch.tensor([[1.0, 0, 0, 0]]).repeat(10, 1).cuda(), 1.0, e,
                                 cam.viewmatrix.cuda(), cam.projmatrix.cuda(), cam.tanfovx, cam.tanfovy, 2160, 3840, e, 0,
                                 cam.campos.cuda(), False, False)
    assert out[0] > 0


def test_grad_arena_gradients_are_born_in_the_flat_bucket():
    """parallel.FlatGradBucket.arm(): the next backward writes straight into the all-reduce buffer, autograd adopts
    the slices as p.grad (no pack copy), values identical to the ordinary path; one-shot; refuses when a p.grad exists."""
    from gaustudio_amd import parallel
    cam = scenes.make_camera(160, 96)
    sc = scenes.make_scene(3000, cam, seed=4)
    rs = _settings(cam)
    g = [x.cuda() for x in scenes.make_output_grads(cam, seed=2)]
    (c, _, d, m, o), leaves, _ = _render(sc, cam, rs)
    torch.autograd.backward([c, d, m, o], g)
    want = {k: v.grad.clone() for k, v in leaves.items()}

    (c, _, d, m, o), leaves2, _ = _render(sc, cam, rs)
    bucket = parallel.FlatGradBucket(list(leaves2.values()), roles=leaves2)
    bucket.flat.fill_(float("nan"))
    assert bucket.arm()
    torch.autograd.backward([c, d, m, o], g, retain_graph=True)
    views = bucket.views()
    aliased = [p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, views)]
    assert all(aliased), aliased
    bucket.pack()                                   # nothing to copy
    for k, p in leaves2.items():
        assert torch.equal(p.grad, want[k]), k
    assert not torch.isnan(bucket.flat).any()
    # a p.grad exists now: arming again is refused, and a second backward accumulates normally (2x)
    assert not bucket.arm()
    torch.autograd.backward([c, d, m, o], g)
    for k, p in leaves2.items():
        assert torch.allclose(p.grad, 2 * want[k], rtol=1e-6, atol=0), k
    # one-shot: after being consumed the arena is gone
    (c, _, d, m, o), leaves3, _ = _render(sc, cam, rs)
    torch.autograd.backward([c, d, m, o], g)
    assert all(p.grad.data_ptr() != v.data_ptr() for p, v in zip(leaves3.values(), views))


def test_tile_band_sharding_of_one_view(oracle):
    """one view split by tile rows across processes (emulated here by rendering the bands one after the
    other).  Inside its band every output bit equals the full render's AND THE CPU ORACLE'S full render (the mode's own
    oracle comparison, not only the chain of trust through the library's full view); outside it the image is an empty
    scene's, radii / num_rendered describe the whole view, and the bands' per-Gaussian gradients SUM to the full view's and,
    within the fp32 summation bound of tests/test_gpu_backward.py, to the oracle's double-summed gradients."""
    from gaustudio_amd import parallel
    from util import oracle_forward
    cam = scenes.make_camera(640, 360)                      # 23 tile rows
    sc = scenes.make_scene(60000, cam, seed=14)
    kw = scene_kwargs(sc, True, False)
    grads = scenes.make_output_grads(cam)
    full = hip_forward(sc, cam, 3, kw)
    gfull = hip_backward_raw(full, sc, cam, 3, kw, grads)
    os_ = oracle_forward(oracle, sc, cam, 3, kw)
    ob = oracle.backward(os_, *[g.numpy() for g in grads])
    assert parallel.tile_row_band(360, 0, 2) == (0, 12) and parallel.tile_row_band(360, 1, 2) == (12, 23)
    acc = None
    covered = torch.zeros(360, dtype=torch.bool)
    for rank in range(2):
        lo, hi = parallel.tile_row_band(cam.height, rank, 2)
        with parallel.tile_band(lo, hi) as band:
            part = hip_forward(sc, cam, 3, kw)
            gpart = hip_backward_raw(part, sc, cam, 3, kw, grads)
        rows = band.band_rows(cam.height)
        covered[rows] = True
        assert part["num_rendered"] == full["num_rendered"] and torch.equal(part["radii"], full["radii"])
        assert part["num_binned"] < full["num_binned"]
        for k in ("color", "depth", "median", "opacity"):
            assert torch.equal(part[k][:, rows], full[k][:, rows]), k
            assert np.array_equal(to_np(part[k][:, rows]), os_[k].reshape(part[k].shape)[:, rows]), k + " vs the oracle"
        outside = torch.ones(360, dtype=torch.bool)
        outside[rows] = False
        assert float(part["color"][:, outside].abs().sum()) == 0.0 and float(part["opacity"][:, outside].abs().sum()) == 0.0
        assert bool((part["median"][0][outside] == 15.0).all())
        acc = {k: v.clone() for k, v in gpart.items()} if acc is None else {k: acc[k] + gpart[k] for k in acc}
    assert bool(covered.all())
    for k, v in gfull.items():
        scale = float(v.abs().max())
        assert float((acc[k] - v).abs().max()) <= 1e-5 * scale, k       # two partial sums instead of one: fp32 re-association only
        if k in ob and ob[k].size:
            keep = ob["flip9"] == 0                                       # as tests/test_gpu_backward.py::_check
            err = np.abs(to_np(acc[k]).reshape(ob[k].shape) - ob[k])[keep].max() / max(np.abs(ob[k]).max(), 1e-30)
            assert err < 2e-4, (k, err)


def test_longest_first_tile_order_changes_no_bit():
    """Skewed frames (a tile list > 1024 entries and > 4x the mean; here the clustered bench scene: lists p50 ~ 10, max ~ 58 k) run
    their tiles longest-first in composite_fwd (by list length, from the SECOND such frame on: the regime is remembered per
    device) and in composite_bwd (by the forward's walk lengths) instead of in XCD bands -- gsr_set_option("tile_order").  Which
    workgroup takes which tile changes no output, no per-pixel state and no gradient bit."""
    from gaustudio_amd import _C
    from util import hip_backward_raw
    W = H = 800
    sc = scenes.make_clustered_scene(300_000, W, cam_distance=11.0, seed=0)
    cam = scenes.ring_cameras(5, W, H, radius=11.0)[1]
    kw = scene_kwargs(sc, True, False)
    grads = scenes.make_output_grads(cam, seed=3)
    res = {}
    for order in (0, 1):
        _C.set_option("tile_order", order)
        try:
            hip_forward(sc, cam, 3, kw)                       # first frame of the regime: remembers "skewed"
            hs = hip_forward(sc, cam, 3, kw)                  # second: composite_fwd runs ordered (when enabled)
            hb = hip_backward_raw(hs, sc, cam, 3, kw, grads)
        finally:
            _C.set_option("tile_order", 1)
        res[order] = (hs, hb)
    for k in ("color", "depth", "median", "opacity", "radii", "final_T", "n_contrib", "point_list", "ranges"):
        assert torch.equal(res[0][0][k], res[1][0][k]), k
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "acc"):
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    assert _C.get_option("tile_order") == 1


def test_forward_only_skips_what_only_a_backward_reads_and_refuses_one():
    """gsr_options.forward_only (set by the autograd Functions for calls under torch.no_grad()): the forward leaves out the
    36 B per Gaussian it otherwise keeps for the SH backward -- same images, same state -- and a backward on its buffers
    is refused instead of reading garbage."""
    from gaustudio_amd import _C
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    cam = scenes.make_camera(200, 120)
    sc = scenes.make_scene(5000, cam, seed=21, sigma_px_median=2.5)
    kw = scene_kwargs(sc, True, False)
    normal = hip_forward(sc, cam, 3, kw)
    e = torch.Tensor([])
    out = _C.rasterize_gaussians(torch.zeros(3), sc.means3D.to(dev), e, sc.opacities.to(dev), sc.scales.to(dev), sc.rotations.to(dev), 1.0, e,
                                 cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.tanfovx, cam.tanfovy, cam.height, cam.width,
                                 sc.shs.to(dev), 3, cam.campos.to(dev), False, False, options=(-1,) * 8 + (1,))
    for i, k in ((1, "color"), (2, "depth"), (3, "median"), (4, "opacity"), (5, "radii")):
        assert torch.equal(out[i], normal[k]), k
    fo = dict(normal, geom=out[6], binning=out[7], img=out[8], num_rendered=out[0])
    with pytest.raises(RuntimeError, match="forward_only"):
        hip_backward_raw(fo, sc, cam, 3, kw, scenes.make_output_grads(cam))
    # the refusal also covers the SH stage alone (parts without MAIN reads the jacobians the forward did not keep) ...
    with pytest.raises(RuntimeError, match="forward_only"):
        hip_backward_raw(fo, sc, cam, 3, kw, scenes.make_output_grads(cam), options={}, parts=2)
    # ... and does not depend on the host-side memory of the forwards: with that memory dropped, a backward (debug or not) reads the
    # forward's own record in the image buffer (round 6; rounds 4-5 kept 64 entries and only a debug backward looked at the device)
    keep = [hip_forward(scenes.make_scene(64, cam, seed=s), cam, 0, scene_kwargs(scenes.make_scene(64, cam, seed=s), True, False))["img"] for s in range(70)]
    with pytest.raises(RuntimeError, match="forward_only"):
        hip_backward_raw(fo, sc, cam, 3, kw, scenes.make_output_grads(cam), debug=True)
    _C.set_option("forget_forwards", 1)
    with pytest.raises(RuntimeError, match="forward_only"):
        hip_backward_raw(fo, sc, cam, 3, kw, scenes.make_output_grads(cam))
    del keep
    # through autograd: a no_grad render followed by a differentiable one of the same rasterizer object
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    P = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    r = GaussianRasterizer(rs)
    with torch.no_grad():
        a = r(means3D=P["means3D"], means2D=torch.zeros_like(P["means3D"]), opacities=P["opacities"], shs=P["shs"], scales=P["scales"], rotations=P["rotations"])
    b = r(means3D=P["means3D"], means2D=torch.zeros_like(P["means3D"]), opacities=P["opacities"], shs=P["shs"], scales=P["scales"], rotations=P["rotations"])
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    b[0].sum().backward()
    assert P["means3D"].grad is not None and bool(torch.isfinite(P["means3D"].grad).all()) and float(P["means3D"].grad.abs().sum()) > 0


def test_staged_instance_count_of_composite_fwd():
    """gsr_inspect_staged: the instances composite_fwd fetched before every pixel of a tile had saturated (bench.py's `roofline` divides
    THESE bytes by the kernel time).  Bounds per frame: at least every tile's list up to its last contributor (rounded up to the
    256-entry batches the kernel stages), at most the binned instances; a sparse frame stages everything, a dense one (long
    lists, opaque splats) a fraction."""
    from gaustudio_amd import _C
    cam = scenes.make_camera(320, 200)
    for P, sigma, want_all in ((3000, 1.5, True), (150000, 6.0, False)):
        sc = scenes.make_scene(P, cam, seed=3, sigma_px_median=sigma)
        hs = hip_forward(sc, cam, 1, scene_kwargs(sc, True, False))
        staged = _C.inspect_staged(hs["img"], cam.width, cam.height)
        r = hs["ranges"].long()
        total = (r[:, 1] - r[:, 0])
        gx = (cam.width + 15) // 16
        nc = torch.zeros(((cam.height + 15) // 16) * 16, gx * 16, dtype=torch.long, device=total.device)
        nc[:cam.height, :cam.width] = hs["n_contrib"].long()
        last = nc.view(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values           # last contributor per tile
        lower = torch.minimum(total, (last + 255) // 256 * 256)
        assert int(lower.sum()) <= staged <= hs["num_binned"], (int(lower.sum()), staged, hs["num_binned"])
        if want_all:
            assert staged == hs["num_binned"]
        else:
            assert staged < 0.8 * hs["num_binned"] and int(total.max()) > 1024


def test_plain_c_abi_backward_follows_its_forwards_mode_with_100_forwards_outstanding():
    """VERDICT r5 #7 / ADVICE r5: a plain-C-ABI caller (gsr_forward / gsr_backward, no gsr_options) that mixes the two exp modes in
    one process through the process default and keeps 100 forwards alive before running their backwards: every backward runs in
    ITS forward's mode whatever the process default says by then -- from the host-side map (one entry per live image buffer; it
    was a 64-entry ring that silently fell back to the process default) and, with the map dropped, from the forward's own 4-byte
    control word in the image buffer.  Gradients bit-equal to the per-mode references."""
    from gaustudio_amd import _C
    cam = scenes.make_camera(160, 96)
    sc = scenes.make_scene(3000, cam, seed=2)
    kw = scene_kwargs(sc, True, False)
    grads = scenes.make_output_grads(cam)
    keys = ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations")
    old = _C.get_option("fast_exp")
    try:
        want = {}
        for mode in (0, 1):
            _C.set_option("fast_exp", mode)
            st = hip_forward(sc, cam, 3, kw)
            want[mode] = hip_backward_raw(st, sc, cam, 3, kw, grads, options=dict(fast_exp=mode))
        assert not torch.equal(want[0]["dL_dmeans3D"], want[1]["dL_dmeans3D"])
        rng = np.random.default_rng(0)
        modes = [int(m) for m in rng.integers(0, 2, 100)]
        states = []
        for m in modes:                                         # 100 forwards, interleaved modes, all kept alive
            _C.set_option("fast_exp", m)
            states.append(hip_forward(sc, cam, 3, kw))          # rasterize_gaussians without options -> plain gsr_forward
        assert len({int(s["img"].data_ptr()) for s in states}) == 100
        for forget in (False, True):
            for i in (list(range(100)) if not forget else [0, 1, 2, 50, 98, 99]):
                if forget:
                    _C.set_option("forget_forwards", 1)         # the map does not know this buffer: the control word is read
                _C.set_option("fast_exp", 1 - modes[i])         # the process default now says the OTHER mode
                g = hip_backward_raw(states[i], sc, cam, 3, kw, grads)     # plain gsr_backward, no options, no debug
                for k in keys:
                    assert torch.equal(g[k], want[modes[i]][k]), (forget, i, modes[i], k)
        # naming the other mode explicitly is still an error, map or no map
        _C.set_option("forget_forwards", 1)
        with pytest.raises(RuntimeError, match="fast_exp differs"):
            hip_backward_raw(states[0], sc, cam, 3, kw, grads, options=dict(fast_exp=1 - modes[0]))
    finally:
        _C.set_option("fast_exp", old)
