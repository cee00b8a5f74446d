"""-m gpu: the operator driven exactly the way GauStudio's renderers drive it.

The reference checkout is not present on the GPU box, so the call sequence of
gaustudio/renderers/base.py:10-63 (BaseRenderer.render) is replayed here step by step -- keyword-constructed
GaussianRasterizationSettings, a CPU `bg` tensor (vanilla_renderer.py:23, pcd_renderer.py:20), the
`zeros_like(xyz, requires_grad=True) + 0` screen-space carrier with retain_grad(), `sh_degree = active degree if shs
is given else 1`, keyword call of the rasterizer with None for the absent inputs, `median[2:3].int()`, `radii > 0`
-- with the property sets VanillaRenderer.get_gaussians_properties (vanilla_renderer.py:28-51: activated attributes of
the point cloud, models/vanilla_sg.py:33-37) and PCDRenderer.get_gaussians_properties (pcd_renderer.py:24-36:
opacity = ones_like(xyz) i.e. [P,3], scales = ones * kernel_size, identity rotations, colors_precomp = rgb / 255)
produce.  Results are checked against the CPU oracle.  (On a machine that has /root/reference, the import-level test
in test_api_surface.py additionally loads the unmodified renderer classes against this module.)

The replay itself (tests/caller_replay.py) is PINNED to the real classes: tests/golden/py_render_calls.json is a recording of
what the operator receives when the unmodified VanillaRenderer / PCDRenderer run (dev container, recording rasterizer), and
`test_replay_equals_the_recorded_reference_calls` asserts that the replay produces the identical record on this box -- then
pushes the very same calls through the real operator.
"""
import math

import numpy as np
import pytest
import torch

from gaustudio_amd import scenes
from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

import caller_replay
from util import to_np

pytestmark = pytest.mark.gpu


from caller_replay import render_like_base_renderer as _render_like_base_renderer      # base.py:10-63, call for call


def _Camera(cam: scenes.Cam, dev):
    """The attributes of gaustudio.datasets.Camera that BaseRenderer.render reads (datasets/__init__.py:138-183), with the
    strides the reference's Camera produces: a transposed-view world_view_transform, a sliced camera_center."""
    view = cam.viewmatrix.t().contiguous().t()
    return caller_replay.Camera(cam.width, cam.height, 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy), view.to(dev),
                                cam.projmatrix.to(dev), torch.inverse(view)[3][:3].to(dev))


def test_vanilla_renderer_call_pattern_forward_and_backward(oracle):
    """VanillaRenderer: raw point-cloud attributes -> torch activations -> the op, inside a larger autograd graph;
    active SH degree 2 of 16 stored coefficients; CPU background; gradients back to the RAW attributes."""
    dev = torch.device("cuda")
    cam = scenes.make_camera(320, 200)
    sc = scenes.make_scene(6000, cam, seed=31, sigma_px_median=2.5)
    # raw attributes such that the activations reproduce the scene (vanilla_sg.py:33-37: exp / sigmoid / normalize)
    raw = dict(xyz=sc.means3D.clone(), scale=torch.log(sc.scales), rot=sc.rotations * 1.7,
               opacity=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)), f_dc=sc.shs[:, :1].clone(), f_rest=sc.shs[:, 1:].clone())
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}

    def properties():   # vanilla_renderer.py:28-51 with convert_SHs_python = compute_cov3D_python = False
        return (leaves["xyz"], torch.cat((leaves["f_dc"], leaves["f_rest"]), dim=1), None, torch.sigmoid(leaves["opacity"]),
                torch.exp(leaves["scale"]), torch.nn.functional.normalize(leaves["rot"]), None)

    bg = torch.tensor([0, 0, 0], dtype=torch.float32)                  # CPU tensor, as the renderer builds it
    pkg = _render_like_base_renderer(properties(), _Camera(cam, dev), active_sh_degree=2, bg_color=bg)
    assert pkg["rendered_median_id"].dtype == torch.int32 and pkg["visibility_filter"].dtype == torch.bool
    grads = [g.to(dev) for g in scenes.make_output_grads(cam, seed=3)]
    median_full = torch.cat([pkg["rendered_median_depth"], pkg["rendered_median_weight"],
                             torch.zeros_like(pkg["rendered_median_depth"])], 0)
    loss = (pkg["render"] * grads[0]).sum() + (pkg["rendered_depth"] * grads[1]).sum() + \
           (median_full * grads[2]).sum() + (pkg["rendered_final_opacity"] * grads[3]).sum()
    loss.backward()

    # oracle on the activated values (computed by torch in float32, exactly what the op received)
    with torch.no_grad():
        act = dict(means3D=leaves["xyz"], shs=torch.cat((leaves["f_dc"], leaves["f_rest"]), 1),
                   opacities=torch.sigmoid(leaves["opacity"]), scales=torch.exp(leaves["scale"]),
                   rotations=torch.nn.functional.normalize(leaves["rot"]))
        act = {k: v.cpu().numpy() for k, v in act.items()}
    st = oracle.forward(act["means3D"], act["opacities"], cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                        cam.campos.numpy(), cam.width, cam.height, cam.tanfovx, cam.tanfovy, sh_degree=2,
                        shs=act["shs"], scales=act["scales"], rotations=act["rotations"], tight=True)
    assert np.array_equal(to_np(pkg["render"]), st["color"]) and np.array_equal(to_np(pkg["rendered_depth"]), st["depth"])
    assert np.array_equal(to_np(pkg["rendered_median_depth"]), st["median"][0:1])
    assert np.array_equal(to_np(pkg["rendered_median_id"]), st["median"][2:3].astype(np.int32))
    assert np.array_equal(to_np(pkg["radii"]), st["radii"]) and np.array_equal(to_np(pkg["visibility_filter"]), st["radii"] > 0)
    gm = grads[2].cpu().numpy().copy()
    gm[2] = 0                                                       # the loss does not touch the id channel
    bw = oracle.backward(st, grads[0].cpu().numpy(), grads[1].cpu().numpy(), gm, grads[3].cpu().numpy())
    vp = pkg["viewspace_points"].grad                               # retained on the non-leaf carrier (densification statistics)
    assert vp is not None and np.abs(to_np(vp) - bw["dL_dmeans2D"]).max() <= 2e-4 * np.abs(bw["dL_dmeans2D"]).max()
    assert np.abs(to_np(leaves["xyz"].grad) - bw["dL_dmeans3D"]).max() <= 2e-4 * np.abs(bw["dL_dmeans3D"]).max()
    # chain rules of the activations, applied by torch to the op's gradients
    sh_g = np.concatenate([to_np(leaves["f_dc"].grad), to_np(leaves["f_rest"].grad)], 1)
    assert np.abs(sh_g - bw["dL_dsh"]).max() <= 2e-4 * np.abs(bw["dL_dsh"]).max()
    assert not sh_g[:, 9:].any()                                    # coefficients above the active degree get zero gradient
    want_scale = bw["dL_dscales"] * act["scales"]
    assert np.abs(to_np(leaves["scale"].grad) - want_scale).max() <= 2e-4 * np.abs(want_scale).max()
    op = act["opacities"]
    want_op = bw["dL_dopacity"] * op * (1 - op)
    assert np.abs(to_np(leaves["opacity"].grad) - want_op).max() <= 2e-4 * np.abs(want_op).max()
    assert leaves["rot"].grad is not None and torch.isfinite(leaves["rot"].grad).all()


def test_pcd_renderer_call_pattern(oracle):
    """PCDRenderer: colours precomputed, shs=None with sh_degree 1, opacity = ones_like(xyz) ([P,3]: the op reads its
    first P values), scales = ones * kernel_size, identity rotations; white background on the CPU."""
    dev = torch.device("cuda")
    cam = scenes.make_camera(256, 192)
    sc = scenes.make_scene(20000, cam, seed=33)
    xyz = sc.means3D.to(dev)
    rgb = (torch.rand(xyz.shape[0], 3, generator=torch.Generator().manual_seed(1)) * 255).floor().to(dev)
    for kernel_size in (0.0, 0.01):
        opacity = torch.ones_like(xyz, device=xyz.device)
        scales = torch.ones_like(xyz, device=xyz.device) * kernel_size
        rotations = torch.zeros((xyz.shape[0], 4), device=xyz.device)
        rotations[:, 0] = 1
        rotations = torch.nn.functional.normalize(rotations)
        props = (xyz, None, rgb / 255, opacity, scales, rotations, None)
        bg = torch.tensor([1, 1, 1], dtype=torch.float32)
        with torch.no_grad():
            pkg = _render_like_base_renderer(props, _Camera(cam, dev), active_sh_degree=3, bg_color=bg)
        st = oracle.forward(sc.means3D.numpy(), np.ones((xyz.shape[0], 1), np.float32), cam.viewmatrix.numpy(),
                            cam.projmatrix.numpy(), cam.campos.numpy(), cam.width, cam.height, cam.tanfovx, cam.tanfovy,
                            sh_degree=1, colors_precomp=to_np(rgb / 255), scales=to_np(scales), rotations=to_np(rotations),
                            bg=bg.numpy(), tight=True)
        assert np.array_equal(to_np(pkg["render"]), st["color"])
        assert np.array_equal(to_np(pkg["rendered_depth"]), st["depth"])
        assert np.array_equal(to_np(pkg["rendered_final_opacity"]), st["opacity"])
        assert np.array_equal(to_np(pkg["rendered_median_id"]), st["median"][2:3].astype(np.int32))
        assert np.array_equal(to_np(pkg["radii"]), st["radii"]) and int(pkg["visibility_filter"].sum()) > 1000


def test_renderer_debug_flag_and_mark_visible():
    """debug=True (the renderers' `debug` option) runs the same call with per-stage synchronisation and dumps; the
    rasterizer's markVisible is what gaustudio's densification code calls on the same settings object."""
    dev = torch.device("cuda")
    cam = scenes.make_camera(128, 96)
    sc = scenes.make_scene(3000, cam, seed=35)
    props = (sc.means3D.to(dev), sc.shs.to(dev), None, sc.opacities.to(dev), sc.scales.to(dev), sc.rotations.to(dev), None)
    bg = torch.tensor([0, 0, 0], dtype=torch.float32)
    with torch.no_grad():
        a = _render_like_base_renderer(props, _Camera(cam, dev), 3, bg, debug=False)
        b = _render_like_base_renderer(props, _Camera(cam, dev), 3, bg, debug=True)
    for k in ("render", "rendered_depth", "rendered_median_depth", "rendered_final_opacity", "radii"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("used", [(0,), (0, 2), (3,), (2, 4)])
def test_loss_on_a_subset_of_the_outputs(used):
    """Outputs the loss does not use get no materialised zero gradient from autograd (set_materialize_grads(False), radii
    marked non-differentiable): the operator fills in the zeros itself, and the parameter gradients equal those of a
    backward that was handed explicit zeros for the unused outputs."""
    dev = "cuda"
    cam = scenes.make_camera(200, 136)
    sc = scenes.make_scene(4000, cam, seed=7)
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    gr = [g.to(dev) for g in scenes.make_output_grads(cam, seed=9)]          # for outputs 0, 2, 3, 4
    slot = {0: 0, 2: 1, 3: 2, 4: 3}

    def run(explicit_zeros):
        P = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        m2 = torch.zeros_like(P["means3D"], requires_grad=True)
        out = GaussianRasterizer(rs)(means3D=P["means3D"], means2D=m2, opacities=P["opacities"], shs=P["shs"],
                                     scales=P["scales"], rotations=P["rotations"])
        assert not out[1].requires_grad
        if explicit_zeros:
            idx = (0, 2, 3, 4)
            torch.autograd.backward([out[i] for i in idx], [gr[slot[i]] if i in used else torch.zeros_like(out[i]) for i in idx])
        else:
            torch.autograd.backward([out[i] for i in used], [gr[slot[i]] for i in used])
        return [P[k].grad for k in P] + [m2.grad]

    for a, b in zip(run(False), run(True)):
        assert a is not None and torch.equal(a, b)


def test_covariance_gradient_only_when_covariances_were_supplied():
    """dL_dcov3D has a reader only when the caller passed cov3D_precomp: then the autograd path returns it (equal to the
    C-ABI's), otherwise the binding passes NULL and the library writes 24 B per Gaussian less; a NULL dL_dcov3D together
    with a cov3D_precomp is an argument error; everything else is bit-equal either way."""
    from util import hip_backward_raw, hip_forward, scene_kwargs
    dev = "cuda"
    cam = scenes.make_camera(200, 136)
    sc = scenes.make_scene(4000, cam, seed=11)
    grads = [g.to(dev) for g in scenes.make_output_grads(cam, seed=5)]
    rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    # (1) covariances supplied
    kw = scene_kwargs(sc, True, True)
    hs = hip_forward(sc, cam, 3, kw)
    raw = hip_backward_raw(hs, sc, cam, 3, kw, grads)
    cov = kw["cov3D_precomp"].to(dev).requires_grad_(True)
    xyz = sc.means3D.to(dev).requires_grad_(True)
    out = GaussianRasterizer(rs)(means3D=xyz, means2D=torch.zeros_like(xyz, requires_grad=True), opacities=sc.opacities.to(dev),
                                 shs=sc.shs.to(dev), cov3D_precomp=cov)
    torch.autograd.backward([out[0], out[2], out[3], out[4]], grads)
    assert cov.grad is not None and torch.equal(cov.grad, raw["dL_dcov3D"]) and float(cov.grad.abs().max()) > 0
    assert torch.equal(xyz.grad, raw["dL_dmeans3D"])
    with pytest.raises(Exception):
        hip_backward_raw(hs, sc, cam, 3, kw, grads, no_dcov=True)
    # (2) scale / rotation: NULL dL_dcov3D, the other seven outputs unchanged
    kw = scene_kwargs(sc, True, False)
    kw["rotations"] = sc.rotations
    hs = hip_forward(sc, cam, 3, kw)
    a = hip_backward_raw(hs, sc, cam, 3, kw, grads)
    b = hip_backward_raw(hs, sc, cam, 3, kw, grads, no_dcov=True)
    for k in a:
        if k != "dL_dcov3D":
            assert torch.equal(a[k], b[k]), k
    assert bool(torch.isnan(b["dL_dcov3D"]).all())       # untouched (the helper poisons its buffers with NaN)


def _fixture_cases():
    import json
    import os
    import sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import render_call_record as rcr
    return rcr, json.load(open(os.path.join(gold, "py_render_calls.json")))["cases"]


def test_replay_equals_the_recorded_reference_calls():
    """Every recorded case (8: VanillaRenderer training / no_grad / white bg + modifier + debug / cov3D in Python / SH in
    Python / 2-column scales; PCDRenderer default / kernel + white bg): the replay on the GPU gives the record the
    unmodified reference classes gave -- settings tuple field by field, keyword set, None-ness, shapes, dtypes, strides'
    contiguity, requires_grad / leaf / retained grad, grad mode, the CPU `bg` -- and the package it returns has the recorded
    keys, dtypes and shapes.  Then the same properties go through the REAL operator."""
    import json
    rcr, fixture = _fixture_cases()
    assert sorted(fixture) == sorted(c["name"] for c in rcr.CASES)
    for case in rcr.CASES:
        want = fixture[case["name"]]
        got = caller_replay.replay_case(case, "cuda")
        assert json.loads(json.dumps(rcr.comparable(got))) == rcr.comparable(want), case["name"]
        assert got["returns"] == want["returns"] and got["bg_is_the_renderers_cpu_tensor"] and want["bg_is_the_renderers_cpu_tensor"]
        assert want["torch_factory_calls_with_device_cuda"][-1] == ["zeros_like", "cuda"]          # base.py:13: the carrier is asked for on "cuda"
        # the real operator on the same call
        raw = rcr.raw_attributes(case, "cuda")
        c = scenes.make_camera(96, 64)
        cam = _Camera(c, "cuda")
        bg, modifier, debug, conf = caller_replay.renderer_state(case["renderer"], case["config"])
        with torch.set_grad_enabled(not case.get("no_grad", False)):
            raw["xyz"] = raw["xyz"] * 0.5 + torch.tensor([0.0, 0.0, 5.0], device="cuda")        # in front of the camera
            props = caller_replay.properties_like_reference(case["renderer"], conf, raw, case["active_sh_degree"], cam)
            pkg = _render_like_base_renderer(props, cam, case["active_sh_degree"], bg, modifier, debug)
        for k, v in want["returns"].items():
            assert list(pkg[k].shape) == v["shape"] and str(pkg[k].dtype) == v["dtype"], (case["name"], k)
            assert bool(torch.isfinite(pkg[k].float()).all())
        assert int(pkg["visibility_filter"].sum()) > 0, case["name"]
        if torch.is_grad_enabled() and not case.get("no_grad", False) and pkg["render"].requires_grad:
            pkg["render"].sum().backward()
            assert pkg["viewspace_points"].grad is not None
