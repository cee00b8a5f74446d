"""-m gpu: fused parameter activations against the unfused path -- torch activations exactly
as VanillaPointCloud.get_attribute / get_features apply them (gaustudio/models/vanilla_sg.py:58-63,103-106), then the
standard operator -- for outputs and for the gradients of the RAW attributes."""
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


@pytest.mark.parametrize("D", [0, 3])
def test_fused_matches_unfused(D):
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

    assert torch.equal(out_a[1], out_b[1]) or (out_a[1] != out_b[1]).float().mean() < 1e-4      # radii
    for i, name in ((0, "color"), (2, "depth"), (4, "opacity")):
        d = (out_a[i] - out_b[i]).detach().abs()
        assert float((d > 1e-5).float().mean()) < 1e-4 and float(d.max()) < 6e-3, name            # flip budget as in test_gpu_ref
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
