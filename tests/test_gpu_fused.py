"""-m gpu: fused parameter activations against the unfused path -- torch activations exactly
as VanillaPointCloud.get_attribute / get_features apply them (gaustudio/models/vanilla_sg.py:58-63,103-106), then the
standard operator -- for outputs and for the gradients of the RAW attributes; test_fused_matches_unfused also anchors
the comparison on the CPU oracle directly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gaustudio_amd import scenes

pytestmark = pytest.mark.gpu


def _raw_params(sc, dev):
    g = torch.Generator().manual_seed(3)
    raw = dict(xyz=sc.means3D.clone(), f_dc=sc.shs[:, :1, :].reshape(-1, 3).clone(), f_rest=sc.shs[:, 1:, :].reshape(-1, 45).clone(),
               opacity=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)), scale=torch.log(sc.scales),
               rot=sc.rotations * (0.5 + torch.rand(sc.rotations.shape[0], 1, generator=g) * 2.0))   # NOT unit length
    return {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}


def _assert_images_equal_up_to_attributed_events(sc, cam, D, a, out_a, out_b, oracle=None):
    """With `oracle`: the unfused forward on the torch-activated parameters is first shown bit-equal to the CPU ORACLE's
    (every output and intermediate), so what follows is the fused mode's own comparison with the oracle, not only with the
    library.  The in-kernel activations (expf / sqrtf of the device library) differ from torch's in the last bit, so the two
    paths see parameters a few 1e-8 apart: images agree to 1e-5 except where that moves an alpha / transmittance across a
    threshold or a radius across a ceil() -- and every such pixel must be ATTRIBUTED by the replay of tests/attribution.py
    (decoded state of the unfused forward), exactly as against the reference kernels.  No blanket flip budget."""
    import attribution
    from util import hip_forward
    sc_act = scenes.Scene(a["xyz"].detach().cpu(), torch.exp(a["scale"]).detach().cpu(), F.normalize(a["rot"]).detach().cpu(),
                          torch.sigmoid(a["opacity"]).detach().cpu(),
                          torch.cat((a["f_dc"].reshape(-1, 1, 3), a["f_rest"].reshape(-1, 15, 3)), dim=1).detach().cpu())
    st = hip_forward(sc_act, cam, D, dict(shs=sc_act.shs, scales=sc_act.scales, rotations=sc_act.rotations))
    assert torch.equal(st["color"], out_a[0]) and torch.equal(st["radii"], out_a[1])      # the same forward, decoded
    if oracle is not None:
        from util import compare_forward_exact, oracle_forward
        compare_forward_exact(st, oracle_forward(oracle, sc_act, cam, D, dict(shs=sc_act.shs, scales=sc_act.scales, rotations=sc_act.rotations)))
    assert float((out_a[1] != out_b[1]).float().mean()) < 1e-4                            # radii: a ceil() on the other side at most
    ia = {k: out_a[i].detach().cpu().numpy() for i, k in ((0, "color"), (2, "depth"), (4, "opacity"))}
    ib = {k: out_b[i].detach().cpu().numpy() for i, k in ((0, "color"), (2, "depth"), (4, "opacity"))}
    rep = attribution.attribute_images(st, cam.width, cam.height, ia, ib, tol=1e-5, depth_scale=20.0,
                                       radii_b=out_b[1].detach().cpu().numpy())
    assert not rep["unattributed"], (rep["flagged"], rep["unattributed"][:3])
    return rep


def test_fused_matches_unfused_at_c3_size():
    """The headline size (1 M Gaussians, 1920x1080, SH 3), forward: fused vs unfused, every differing pixel attributed."""
    from gaustudio_amd.fused import FusedGaussianRasterizer
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    cam = scenes.make_camera(1920, 1080)
    sc = scenes.make_scene(1_000_000, cam, seed=0)
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    a = _raw_params(sc, dev)
    with torch.no_grad():
        shs = torch.cat((a["f_dc"].reshape(-1, 1, 3), a["f_rest"].reshape(-1, 15, 3)), dim=1)
        out_a = GaussianRasterizer(rs)(means3D=a["xyz"], means2D=None, opacities=torch.sigmoid(a["opacity"]), shs=shs,
                                       scales=torch.exp(a["scale"]), rotations=F.normalize(a["rot"]))
        out_b = FusedGaussianRasterizer(rs)(means3D=a["xyz"], means2D=None, raw_opacities=a["opacity"], f_dc=a["f_dc"],
                                            f_rest=a["f_rest"], raw_scales=a["scale"], raw_rotations=a["rot"])
    rep = _assert_images_equal_up_to_attributed_events(sc, cam, 3, a, out_a, out_b)
    assert rep["flagged"] < 2000            # a handful of events per 2 M pixels, not a systematic difference


@pytest.mark.parametrize("D", [0, 3])
def test_fused_matches_unfused(oracle, D):
    from gaustudio_amd.fused import FusedGaussianRasterizer
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    cam = scenes.make_camera(320, 200)
    sc = scenes.make_scene(20000, cam, seed=12, sigma_px_median=2.0)
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), D, cam.campos.to(dev), False, False)
    grads = [g.to(dev) for g in scenes.make_output_grads(cam, seed=4)]

    a = _raw_params(sc, dev)                                  # unfused: activations by torch
    shs = torch.cat((a["f_dc"].reshape(-1, 1, 3), a["f_rest"].reshape(-1, 15, 3)), dim=1)
    out_a = GaussianRasterizer(rs)(means3D=a["xyz"], means2D=torch.zeros_like(a["xyz"], requires_grad=True),
                                   opacities=torch.sigmoid(a["opacity"]), shs=shs, scales=torch.exp(a["scale"]),
                                   rotations=F.normalize(a["rot"]))
    torch.autograd.backward([out_a[0], out_a[2], out_a[3], out_a[4]], grads)

    b = _raw_params(sc, dev)                                  # fused: raw attributes straight into the operator
    out_b = FusedGaussianRasterizer(rs)(means3D=b["xyz"], means2D=torch.zeros_like(b["xyz"], requires_grad=True),
                                        raw_opacities=b["opacity"], f_dc=b["f_dc"], f_rest=b["f_rest"],
                                        raw_scales=b["scale"], raw_rotations=b["rot"])
    torch.autograd.backward([out_b[0], out_b[2], out_b[3], out_b[4]], grads)

    _assert_images_equal_up_to_attributed_events(sc, cam, D, a, out_a, out_b, oracle)
    for k in a:
        ga, gb = a[k].grad, b[k].grad
        assert gb is not None and gb.shape == ga.shape, k
        assert float((ga - gb).abs().max()) <= 2e-4 * float(ga.abs().max()) + 1e-12, (k, float((ga - gb).abs().max()), float(ga.abs().max()))
    assert float(b["rot"].grad.abs().max()) > 0
    assert (float(b["f_rest"].grad.abs().max()) > 0) == (D > 0)        # coefficients above the active degree get zeros


def test_fused_no_activation_flags_equals_standard_operator():
    """activation flags 0 + split SH storage must reproduce the standard operator bit for bit."""
    from gaustudio_amd.fused import FusedGaussianRasterizer
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    cam = scenes.make_camera(160, 120)
    sc = scenes.make_scene(5000, cam, seed=2)
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    with torch.no_grad():
        ref = GaussianRasterizer(rs)(means3D=sc.means3D.to(dev), means2D=None, opacities=sc.opacities.to(dev),
                                     shs=sc.shs.to(dev), scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
        got = FusedGaussianRasterizer(rs, activations=0)(
            means3D=sc.means3D.to(dev), means2D=None, raw_opacities=sc.opacities.to(dev), f_dc=sc.shs[:, :1].to(dev),
            f_rest=sc.shs[:, 1:].to(dev), raw_scales=sc.scales.to(dev), raw_rotations=sc.rotations.to(dev))
    for x, y in zip(ref, got):
        assert torch.equal(x, y)
