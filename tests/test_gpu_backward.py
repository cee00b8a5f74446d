"""-m gpu: backward parity of the HIP path (through the C ABI) against the CPU oracle.

Two checks, both rigorous:
  (1) composite stage.  The reference sums per-Gaussian contributions with float atomicAdd in an
      unspecified order (backward.cu:559-607); our kernel reduces each wave with a fixed DPP network and
      then adds the two waves of a tile and the tiles of a Gaussian in fixed order.  The oracle sums the SAME
      fp32 contributions in double and also returns S = sum of their magnitudes -- for the terms that come out
      of a cancellation (dL_dalpha = <colour - accumulated colour, dL_dpixel> ...), the magnitude of what was
      subtracted, because a differently rounded evaluation (one scalar recurrence instead of five, v_rcp instead
      of a division) is accurate relative to THAT, not to the cancelled result.  Any-order fp32 evaluation of n
      such terms differs from the exact sum by at most ~n*2^-24*S, so we require |hip - oracle| <= 4e-5*S + 1e-30
      (n <~ 600 per Gaussian here); typical observed error is ~1e-7*S.
  (2) per-Gaussian stage (cov2D / projection / SH / cov3D backward).  Fed with the HIP accumulator
      rows, the oracle's restatement of backward.cu:144-412 must reproduce the HIP outputs BIT-EXACTLY.
"""
import numpy as np
import pytest
import torch

from gaustudio_amd import scenes

from util import ab_variants, assert_bits_equal, hip_backward_raw, hip_forward, oracle_forward, scene_kwargs, to_np

pytestmark = pytest.mark.gpu

GRAD_KEYS = ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations")


def _check(oracle, sc, cam, D, kw, scale_modifier=1.0, bg=None, seed=1, e2e_tol=2e-4):
    grads = scenes.make_output_grads(cam, seed=seed)
    os_ = oracle_forward(oracle, sc, cam, D, kw, scale_modifier, bg)
    ob = oracle.backward(os_, *[g.numpy() for g in grads])
    hs = hip_forward(sc, cam, D, kw, scale_modifier, bg)
    hb = hip_backward_raw(hs, sc, cam, D, kw, grads, scale_modifier, bg)
    # (0) round 6: the same backward in two BANDS (cut at a seed-dependent tile row) + the SH stage: every output bit for bit
    if hs["num_rendered"] > 0:
        gy = (cam.height + 15) // 16
        S = (seed * 7 + 3) % (gy + 1)
        hbb = hip_backward_raw(hs, sc, cam, D, kw, grads, scale_modifier, bg, options={}, parts=1 | 16, sh_g0=S)
        hbb = hip_backward_raw(hs, sc, cam, D, kw, grads, scale_modifier, bg, options={}, parts=1 | 32, sh_g0=S, reuse=hbb)
        hbb = hip_backward_raw(hs, sc, cam, D, kw, grads, scale_modifier, bg, options={}, parts=2, reuse=hbb)
        for k in GRAD_KEYS + ("acc",):
            if k == "dL_dsh" and "shs" not in kw:
                continue
            assert torch.equal(hbb[k], hb[k]), f"banded backward (cut at tile row {S} of {gy}) differs in {k}"
    # (1) composite-stage sums
    acc = to_np(hb["acc"]).astype(np.float64)
    err = np.abs(acc - ob["acc"])
    bound = 4e-5 * ob["accabs"] + 1e-30
    # the median-depth gradient goes to the Gaussian at which a T reconstructed by division crosses 0.5
    # (backward.cu:566): events within rounding distance of the threshold may land on a neighbour (or nowhere)
    bound[:, 9] += ob["flip9"]
    worst = float((err / np.maximum(bound, 1e-300)).max() * 4e-5)
    assert (err <= bound).all(), f"composite_bwd sums outside the fp32 summation bound: worst err/S = {worst:.3e}"
    vis = os_["radii"] > 0
    assert not np.abs(acc[~vis]).any()
    # (2) per-Gaussian stage, bit-exact given the same sums
    fin = oracle.finish_backward(os_, to_np(hb["acc"]))
    for k in GRAD_KEYS:
        a = to_np(hb[k])
        assert np.isfinite(a).all(), f"{k}: non-finite / unwritten output"
        b = fin[k]
        if k in ("dL_dsh",) and "shs" not in kw:
            continue
        assert_bits_equal(a.reshape(b.shape), b, k)
    # and the end-to-end numbers against the double-summed oracle, relative to each tensor's scale
    rel = {}
    for k in GRAD_KEYS:
        b = ob[k]
        if b.size == 0:
            continue
        a = to_np(hb[k]).reshape(b.shape)
        keep = ob["flip9"] == 0                      # Gaussians without an ill-conditioned median event
        rel[k] = float(np.abs(a - b)[keep].max() / max(np.abs(b).max(), 1e-30)) if keep.any() else 0.0
        assert rel[k] < e2e_tol, (k, rel[k])
    return rel, worst


@pytest.mark.parametrize("D", [0, 3])
def test_backward_c1(oracle, D):
    cam = scenes.make_camera(400, 400)
    sc = scenes.make_scene(10000, cam, seed=0)
    _check(oracle, sc, cam, D, scene_kwargs(sc, True, False))


def test_backward_variants(oracle):
    cam = scenes.make_camera(333, 211)
    sc = scenes.make_scene(6000, cam, seed=4, sigma_px_median=3.0)
    _check(oracle, sc, cam, 2, scene_kwargs(sc, True, False), scale_modifier=1.3)
    _check(oracle, sc, cam, 0, scene_kwargs(sc, False, False))                 # precomputed colours
    _check(oracle, sc, cam, 3, scene_kwargs(sc, True, True))                   # precomputed covariance
    _check(oracle, sc, cam, 1, scene_kwargs(sc, True, False), bg=torch.tensor([1.0, 0.5, 0.25]))   # Q1: bg in backward only


def test_backward_ring_camera(oracle):
    sc = scenes.make_ball_scene(15000, radius=3.0, seed=9, sigma=0.04)
    cam = scenes.ring_cameras(5, 256, 192, radius=8.0)[2]
    _check(oracle, sc, cam, 3, scene_kwargs(sc, True, False))


@pytest.mark.parametrize("D", [3, 1])      # 1: [P,16,3] rows wider than the active degree (preprocess_bwd_sh_wide_kernel)
def test_backward_zero_rows_for_culled(oracle, D):
    cam = scenes.make_camera(128, 96)
    sc = scenes.make_scene(4000, cam, seed=6)
    m = sc.means3D.clone()
    m[::3, 2] *= -1.0                                                          # a third behind the camera
    sc = sc._replace(means3D=m.contiguous())
    kw = scene_kwargs(sc, True, False)
    grads = scenes.make_output_grads(cam)
    hs = hip_forward(sc, cam, D, kw)
    hb = hip_backward_raw(hs, sc, cam, D, kw, grads)
    culled = to_np(hs["radii"]) == 0
    assert culled.sum() >= 4000 // 3
    for k in GRAD_KEYS:
        assert not to_np(hb[k])[culled].any(), k
    if D < 3:                                   # coefficients above the active degree get exact zeros, visible or not
        assert not to_np(hb["dL_dsh"])[:, (D + 1) ** 2:, :].any()
        assert to_np(hb["dL_dsh"])[~culled][:, :(D + 1) ** 2, :].any()
    _check(oracle, sc, cam, D, kw)


@pytest.mark.parametrize("fast_exp", [False, True])
def test_null_upstream_gradients_equal_explicit_zeros(oracle, fast_exp):
    """include/gsrast.h gsr_backward: each of the four upstream image gradients may be NULL = "the loss does not use that
    output" = zero; nothing is loaded for it.  Every output (and the composite-stage sums) must be BIT-equal to the call with
    explicit zero planes -- for every subset, in both exp modes, in the short- and the long-list regime, with a non-black
    background, through the per-quarter kernels (colour alone: the CONLY instantiation) and the per-wave A/B kernel -- and the
    explicit-zero call is held against the oracle as every other backward is."""
    import gaustudio_amd
    cases = [(scenes.make_camera(333, 211), dict(seed=4, sigma_px_median=3.0), 6000, None),
             (scenes.make_camera(96, 64), dict(seed=5, sigma_px_median=9.0), 9000, torch.tensor([1.0, 0.5, 0.25]))]   # long lists: row flags
    for cam, skw, P, bg in cases:
        sc = scenes.make_scene(P, cam, **skw)
        kw = scene_kwargs(sc, True, False)
        full = scenes.make_output_grads(cam, seed=3)
        zeros = [torch.zeros_like(g) for g in full]
        with gaustudio_amd.options(fast_exp=fast_exp):
            hs = hip_forward(sc, cam, 3, kw, bg=bg)
            for keep in ((1, 0, 0, 0), (1, 1, 0, 0), (0, 0, 1, 0), (0, 1, 1, 1), (1, 0, 0, 1), (0, 0, 0, 0)):
                dense = [g if k else z for g, z, k in zip(full, zeros, keep)]
                sparse = [g if k else None for g, k in zip(full, keep)]
                for variant in ({}, dict(bwd_variant=2)) if (not fast_exp and ab_variants()) else ({},):
                    a = hip_backward_raw(hs, sc, cam, 3, kw, dense, bg=bg, options=dict(variant))
                    b = hip_backward_raw(hs, sc, cam, 3, kw, sparse, bg=bg, options=dict(variant))
                    for k in GRAD_KEYS + ("acc",):
                        assert torch.equal(a[k], b[k]), (keep, variant, k)
                if keep == (0, 0, 0, 0):
                    assert all(not to_np(b[k]).any() for k in GRAD_KEYS)
    if not fast_exp:      # the colour-only loss against the oracle (explicit zeros on the oracle's side)
        cam, skw, P, bg = cases[0]
        sc = scenes.make_scene(P, cam, **skw)
        kw = scene_kwargs(sc, True, False)
        full = scenes.make_output_grads(cam, seed=3)
        os_ = oracle_forward(oracle, sc, cam, 3, kw)
        ob = oracle.backward(os_, full[0].numpy(), *[np.zeros_like(g.numpy()) for g in full[1:]])
        hs = hip_forward(sc, cam, 3, kw)
        hb = hip_backward_raw(hs, sc, cam, 3, kw, [full[0], None, None, None])
        err = np.abs(to_np(hb["acc"]).astype(np.float64) - ob["acc"])
        assert (err <= 4e-5 * ob["accabs"] + 1e-30).all()
        fin = oracle.finish_backward(os_, to_np(hb["acc"]))
        for k in GRAD_KEYS:
            assert_bits_equal(to_np(hb[k]).reshape(fin[k].shape), fin[k], k)


def test_colour_only_loss_through_autograd_materialises_no_zero_planes():
    """GaussianRasterizer with a loss on the colour image alone: autograd hands the Function None for the three unused outputs
    (set_materialize_grads(False)); they travel as ABSENT tensors to the C ABI (NULL) -- gradients bit-equal to the same loss
    with explicit zero gradients on the other outputs, and equal through the fused (raw-parameter) Function too."""
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = scenes.make_camera(200, 120)
    sc = scenes.make_scene(3000, cam, seed=8, sigma_px_median=2.5)
    dev = "cuda"
    gcol = scenes.make_output_grads(cam, seed=2)[0].to(dev)
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    got = []
    for explicit in (False, True):
        leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
        m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
        out = GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], shs=leaves["shs"],
                                     scales=leaves["scales"], rotations=leaves["rotations"])
        if explicit:
            torch.autograd.backward([out[0], out[2], out[3], out[4]], [gcol] + [torch.zeros_like(out[i]) for i in (2, 3, 4)])
        else:
            (out[0] * gcol).sum().backward()
        got.append({k: v.grad.clone() for k, v in leaves.items()} | {"means2D": m2.grad.clone()})
    for k in got[0]:
        assert torch.equal(got[0][k], got[1][k]), k
    assert float(got[0]["shs"].abs().sum()) > 0


def test_autograd_function_end_to_end(oracle):
    """Through GaussianRasterizer / _RasterizeGaussians (the interface gaustudio/renderers/base.py uses):
    9-tuple ordering of backward, CPU `bg`, grad carrier means2D."""
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = scenes.make_camera(200, 120)
    sc = scenes.make_scene(3000, cam, seed=8, sigma_px_median=2.5)
    dev = "cuda"
    leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    color, radii, depth, median, opac = GaussianRasterizer(rs)(
        means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
        scales=leaves["scales"], rotations=leaves["rotations"])
    assert color.shape == (3, 120, 200) and radii.dtype == torch.int32 and median.shape == (3, 120, 200)
    grads = [g.to(dev) for g in scenes.make_output_grads(cam)]
    torch.autograd.backward([color, depth, median, opac], grads)
    kw = scene_kwargs(sc, True, False)
    os_ = oracle_forward(oracle, sc, cam, 3, kw)
    ob = oracle.backward(os_, *[g.cpu().numpy() for g in grads])
    pairs = dict(means3D="dL_dmeans3D", scales="dL_dscales", rotations="dL_drotations", opacities="dL_dopacity", shs="dL_dsh")
    for k, ok in pairs.items():
        a = to_np(leaves[k].grad); b = ob[ok].reshape(a.shape)
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max(), k
    a = to_np(means2D.grad); b = ob["dL_dmeans2D"]
    assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max()


def test_backward_is_run_to_run_deterministic():
    """The reference's float atomicAdd makes its gradients order-dependent.  Here every sum has a fixed order: FMAs
    and a DPP butterfly inside a lane group, the quarters of a wave meeting in an instance through LDS adds that the
    wave issues in program order, the four waves of a tile in four planes added in fixed order, per-Gaussian rows
    added in ascending tile order -> bit-identical gradients on every run."""
    cam = scenes.make_camera(800, 800)
    sc = scenes.make_scene(300000, cam, seed=2)
    kw = scene_kwargs(sc, True, False)
    grads = scenes.make_output_grads(cam)
    hs = hip_forward(sc, cam, 3, kw)
    a = hip_backward_raw(hs, sc, cam, 3, kw, grads)
    b = hip_backward_raw(hs, sc, cam, 3, kw, grads)
    hs2 = hip_forward(sc, cam, 3, kw)
    c = hip_backward_raw(hs2, sc, cam, 3, kw, grads)
    for k in GRAD_KEYS:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k


@pytest.mark.parametrize("P,sigma", [(8000, 6.0), (30000, 3.0)])
def test_backward_long_lists_row_flag_regime(oracle, P, sigma):
    """Average tile list > 1024 entries: the backward runs with per-row validity bytes (rows of entries the walk never
    reaches are neither cleared nor read) and the bucketed long-list sort; same two checks as everywhere else."""
    cam = scenes.make_camera(64, 48)
    sc = scenes.make_scene(P, cam, seed=21, sigma_px_median=sigma)
    kw = scene_kwargs(sc, True, False)
    os_ = oracle_forward(oracle, sc, cam, 2, kw)
    r = os_["ranges"]
    assert os_["num_rendered"] > 1024 * r.shape[0] and int((r[:, 1] - r[:, 0]).max()) > 1024
    assert int(os_["n_contrib"].max()) < int((r[:, 1] - r[:, 0]).max())          # some entries are never reached
    _check(oracle, sc, cam, 2, kw, seed=3)


@pytest.mark.parametrize("name,P,W,H,D", [("C2", 300_000, 800, 800, 3), ("C3", 1_000_000, 1920, 1080, 3),
                                            ("C4-view", 5_000_000, 1297, 840, 3), ("C5-view", 2_500_000, 3840, 2160, 3)])
def test_backward_baseline_configs(oracle, name, P, W, H, D):
    """BASELINE configs C2, full-size C3 and one view of the C4 / C5 sizes (the long-list regime with row validity
    flags; a 4K frame): composite-stage sums inside the rigorous fp32 summation bound, the
    per-Gaussian stage bit-exact given the same sums, end-to-end gradients within 2e-4 of each tensor's scale."""
    cam = scenes.make_camera(W, H)
    sc = scenes.make_scene(P, cam, seed=0)
    _check(oracle, sc, cam, D, scene_kwargs(sc, True, False))


@pytest.mark.parametrize("kind", ["needles", "blobs", "threshold", "borders"])
def test_backward_on_adversarial_scenes(oracle, kind):
    """The backward's per-quarter lists come from the forward's block masks: on scenes built to stress the conservative
    culls the sums must still meet the oracle's summation bounds (a wrongly culled live pixel would drop a term)."""
    from test_gpu_forward import _adversarial_scene
    cam = scenes.make_camera(331, 203)
    sc = _adversarial_scene(5000 if kind != "blobs" else 1200, cam, seed=33, kind=kind)
    # needles and blobs make the per-Gaussian covariance derivatives ill-conditioned: the composite sums still have to
    # meet their rigorous bounds and the per-Gaussian stage is still bit-exact given the sums; only the last,
    # end-to-end comparison with the double-precision sums is relative to conditioning and gets a looser limit
    _check(oracle, sc, cam, 2, scene_kwargs(sc, True, False), e2e_tol=2e-2)


def test_backward_variants_of_the_kernel_agree():
    """composite_bwd carries dead pixels through arithmetically (G masked to 0, 1/(1-0) == 1); the variant that keeps
    an explicit select on T must give the same bits.  The per-wave (8x8) kernel of round 2a (variant bit 1) and a run
    on the reference's square binning (the extra instances contribute exact zeros, but shift the 64-instance batches
    and with them the order in which the quarters of a wave meet in an instance) add the same terms in another order:
    equal within the fp32 summation tolerance of every tensor's scale."""
    from gaustudio_amd import _C
    cam = scenes.make_camera(640, 360)
    sc = scenes.make_scene(120000, cam, seed=12)
    kw = scene_kwargs(sc, True, False)
    grads = scenes.make_output_grads(cam)
    hs = hip_forward(sc, cam, 3, kw)
    a = hip_backward_raw(hs, sc, cam, 3, kw, grads)
    old = _C.get_option("bwd_variant")
    try:
        _C.set_option("bwd_variant", 1)
        b = hip_backward_raw(hs, sc, cam, 3, kw, grads)
        w0 = w1 = None
        if ab_variants():
            _C.set_option("bwd_variant", 2)
            w0 = hip_backward_raw(hs, sc, cam, 3, kw, grads)
            _C.set_option("bwd_variant", 3)
            w1 = hip_backward_raw(hs, sc, cam, 3, kw, grads)
        else:         # the shipped build refuses the per-wave kernel loudly
            _C.set_option("bwd_variant", 2)
            with pytest.raises(RuntimeError, match="not in this build"):
                hip_backward_raw(hs, sc, cam, 3, kw, grads)
    finally:
        _C.set_option("bwd_variant", old)
    for k in GRAD_KEYS + ("acc",):
        assert torch.equal(a[k], b[k]), k
        assert w0 is None or torch.equal(w0[k], w1[k]), k
    try:
        _C.set_option("tight_binning", 0)
        hs0 = hip_forward(sc, cam, 3, kw)
        c = hip_backward_raw(hs0, sc, cam, 3, kw, grads)
    finally:
        _C.set_option("tight_binning", 1)
    assert hs0["num_binned"] > hs["num_binned"]
    # the default backward takes its per-block cull from the masks composite_fwd left behind the lists; after a forward
    # that leaves none (per-wave walk) it computes its own, slightly different conservative masks: same live terms
    d = None
    if ab_variants():
        try:
            _C.set_option("fwd_variant", 1)
            hs1 = hip_forward(sc, cam, 3, kw)
            d = hip_backward_raw(hs1, sc, cam, 3, kw, grads)
        finally:
            _C.set_option("fwd_variant", 0)
    for other in [o for o in (w0, c, d) if o is not None]:
        for k in GRAD_KEYS + ("acc",):
            x, y = a[k].double(), other[k].double()
            if x.numel() == 0:
                continue
            scale = float(y.abs().max())
            assert float((x - y).abs().max()) <= 2e-5 * scale + 1e-30, (k, float((x - y).abs().max()), scale)


@pytest.mark.parametrize("fast_exp", [0, 1], ids=["exact", "fast_exp"])
def test_banded_backward_equals_the_one_call_backward_bit_for_bit(fast_exp):
    """Round 6: gsr_backward_ex in two BANDS (GSR_BWD_PART_BAND_FIRST / _SECOND: compositing backward of the tile
    rows above / below a cut + the per-Gaussian stage of the Gaussians that end above it / the others) followed by the SH stage
    == the one-call backward, every output and the accumulator rows bit for bit -- short-list and long-list (row flags) regimes,
    cuts at 0, inside the image and at its end, colour-only and full losses; gsr_band_classes partitions the visible Gaussians."""
    import gaustudio_amd
    from gaustudio_amd import _C
    cases = [(scenes.make_camera(333, 211), dict(seed=4, sigma_px_median=3.0), 6000, None),
             (scenes.make_camera(96, 64), dict(seed=5, sigma_px_median=9.0), 9000, torch.tensor([1.0, 0.5, 0.25]))]   # long lists: row flags
    keys = GRAD_KEYS + ("acc",)
    for cam, skw, P, bg in cases:
        sc = scenes.make_scene(P, cam, **skw)
        kw = scene_kwargs(sc, True, False)
        full = scenes.make_output_grads(cam, seed=3)
        gy = (cam.height + 15) // 16
        with gaustudio_amd.options(fast_exp=fast_exp):
            hs = hip_forward(sc, cam, 3, kw, bg=bg)
        opt = dict(fast_exp=fast_exp)
        for grads in (full, [full[0], None, None, None]):
            want = hip_backward_raw(hs, sc, cam, 3, kw, grads, bg=bg, options=opt)
            for S in (0, 1, gy // 2, gy - 1, gy, gy + 5):
                first, second = _C.band_classes(hs["radii"], hs["geom"], S)
                vis = hs["radii"] > 0
                assert torch.equal((first + second) > 0, vis) and int((first * second).sum()) == 0
                if S >= gy:
                    assert int(second.sum()) == 0
                if S == 0:
                    assert int((first.bool() & (hs["tiles_touched"] > 0)).sum()) == 0
                a = hip_backward_raw(hs, sc, cam, 3, kw, grads, bg=bg, options=opt, parts=1 | 16, sh_g0=S)
                if 0 < S < gy:       # after the first band the Gaussians of class 1 are final, the others untouched (NaN-poisoned)
                    m = ~second.bool()
                    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations"):
                        assert torch.equal(a[k][m], want[k][m]) and bool(torch.isnan(a[k][second.bool()]).all()), (S, k)
                a = hip_backward_raw(hs, sc, cam, 3, kw, grads, bg=bg, options=opt, parts=1 | 32, sh_g0=S, reuse=a)
                a = hip_backward_raw(hs, sc, cam, 3, kw, grads, bg=bg, options=opt, parts=2, reuse=a)
                for k in keys:
                    assert torch.equal(a[k], want[k]), (S, k, "colour-only" if grads[1] is None else "full")
    # a band call with the SH part, or both bands at once, is refused
    with pytest.raises(RuntimeError, match="band call runs GSR_BWD_PART_MAIN only"):
        hip_backward_raw(hs, sc, cam, 3, kw, full, bg=bg, options=opt, parts=1 | 2 | 16, sh_g0=2)
    with pytest.raises(RuntimeError, match="two calls"):
        hip_backward_raw(hs, sc, cam, 3, kw, full, bg=bg, options=opt, parts=1 | 16 | 32, sh_g0=2)
