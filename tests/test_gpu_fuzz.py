"""-m gpu: seeded random configurations (sizes, image shapes, SH degree, footprints, modifiers, input variants) through
the same two checks as the dedicated tests: forward BIT-EXACT against the oracle, backward sums inside the fp32
summation bound and the per-Gaussian stage bit-exact given the sums."""
import numpy as np
import pytest
import torch

from gaustudio_amd import scenes

from test_gpu_backward import _check
from util import compare_forward_exact, hip_forward, oracle_forward, scene_kwargs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_random_configuration(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.integers(17, 420))
    H = int(rng.integers(9, 300))
    P = int(rng.choice([37, 300, 2500, 9000, 20000]))
    D = int(rng.integers(0, 4))
    sigma = float(rng.choice([0.7, 1.5, 4.0, 12.0]))
    use_sh = bool(rng.random() < 0.75)
    use_cov = bool(rng.random() < 0.25)
    mod = float(rng.choice([1.0, 1.0, 0.6, 1.9]))
    bg = torch.tensor(rng.random(3), dtype=torch.float32) if rng.random() < 0.5 else None
    cam = scenes.make_camera(W, H, fovx_deg=float(rng.choice([35.0, 60.0, 95.0])))
    sc = scenes.make_scene(P, cam, seed=seed, sigma_px_median=sigma, zmin=float(rng.choice([0.15, 2.0])))   # 0.15: near-plane culls
    kw = scene_kwargs(sc, use_sh, use_cov)
    Dk = D if use_sh else 0
    os_ = oracle_forward(oracle, sc, cam, Dk, kw, mod, bg)
    hs = hip_forward(sc, cam, Dk, kw, mod, bg)
    compare_forward_exact(hs, os_)
    if os_["num_rendered"] > 0:
        _check(oracle, sc, cam, Dk, kw, scale_modifier=mod, bg=bg, seed=seed)


@pytest.mark.parametrize("D", [0, 1, 2, 3])
@pytest.mark.parametrize("P", [1000, 4097])
def test_sh_rows_holding_exactly_the_active_degree(oracle, D, P):
    """shs with M = (D+1)^2 coefficients (row widths 3, 12, 27, 48 floats -- 27 is not a multiple of four) take the
    LDS-staged cooperative SH kernels in the backward; P = 4097 leaves a one-row last wave."""
    cam = scenes.make_camera(200, 120)
    sc = scenes.make_scene(P, cam, seed=40 + D, sigma_px_median=2.5)
    sc = sc._replace(shs=sc.shs[:, :(D + 1) ** 2, :].contiguous())
    kw = scene_kwargs(sc, True, False)
    os_ = oracle_forward(oracle, sc, cam, D, kw)
    hs = hip_forward(sc, cam, D, kw)
    compare_forward_exact(hs, os_)
    _check(oracle, sc, cam, D, kw, seed=D)


@pytest.mark.parametrize("D", [0, 1, 2, 3])
def test_fused_split_storage_at_every_degree(D):
    """f_dc [P,1,3] + f_rest [P,(D+1)^2-1,3] (staged row widths 0, 9, 24, 45 floats) through the fused interface against
    torch activations + the standard operator, outputs and raw-attribute gradients."""
    import torch.nn.functional as Fn
    from gaustudio_amd.fused import FusedGaussianRasterizer
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    cam = scenes.make_camera(200, 120)
    sc = scenes.make_scene(3001, cam, seed=60 + D, sigma_px_median=2.5)
    M = (D + 1) ** 2
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), D, cam.campos.to(dev), False, False)
    grads = [g.to(dev) for g in scenes.make_output_grads(cam, seed=5)]

    def raw():
        r = dict(xyz=sc.means3D.clone(), f_dc=sc.shs[:, :1, :].clone(), f_rest=sc.shs[:, 1:M, :].clone(),
                 opacity=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)), scale=torch.log(sc.scales), rot=sc.rotations * 1.7)
        return {k: v.contiguous().to(dev).requires_grad_(True) for k, v in r.items()}

    a = raw()
    out_a = GaussianRasterizer(rs)(means3D=a["xyz"], means2D=torch.zeros_like(a["xyz"]), opacities=torch.sigmoid(a["opacity"]),
                                   shs=torch.cat((a["f_dc"], a["f_rest"]), dim=1), scales=torch.exp(a["scale"]),
                                   rotations=Fn.normalize(a["rot"]))
    torch.autograd.backward([out_a[0], out_a[2], out_a[3], out_a[4]], grads)
    b = raw()
    out_b = FusedGaussianRasterizer(rs)(means3D=b["xyz"], means2D=torch.zeros_like(b["xyz"]), raw_opacities=b["opacity"],
                                        f_dc=b["f_dc"], f_rest=b["f_rest"], raw_scales=b["scale"], raw_rotations=b["rot"])
    torch.autograd.backward([out_b[0], out_b[2], out_b[3], out_b[4]], grads)
    for i in (0, 2, 4):
        d = (out_a[i] - out_b[i]).detach().abs()
        assert float((d > 1e-5).float().mean()) < 1e-3 and float(d.max()) < 2e-2, i
    for k in a:
        if a[k].numel() == 0:
            continue
        ga, gb = a[k].grad, b[k].grad
        assert gb is not None and gb.shape == ga.shape, k
        assert float((ga - gb).abs().max()) <= 5e-4 * float(ga.abs().max()) + 1e-12, (k, float((ga - gb).abs().max()), float(ga.abs().max()))
