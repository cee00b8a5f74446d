"""TEST INFRASTRUCTURE: the call sequence of the reference's renderers, replayed by hand for boxes that have no reference
checkout (the GPU box).  `render_like_base_renderer` restates gaustudio/renderers/base.py:10-63 call for call;
`properties_like_reference` restates VanillaRenderer.get_gaussians_properties (vanilla_renderer.py:28-51, with the
activations of models/vanilla_sg.py:58-63,99-106) and PCDRenderer.get_gaussians_properties (pcd_renderer.py:24-36).

A restatement can drift from what it restates, so it is PINNED: tests/golden/py_render_calls.json records what the
operator receives when the UNMODIFIED classes run (tests/golden/make_ref_py_fixtures.py, dev container), and
tests/test_gpu_caller.py / tests/test_api_surface.py assert that this replay, given the same seeded point cloud and a
recording rasterizer, produces the identical record (settings tuple, keywords, None-ness, shapes, dtypes, requires_grad,
leaf-ness, retained grads)."""
import math

import torch

RENDERER_DEFAULTS = {      # vanilla_renderer.py:10-17, pcd_renderer.py:7-13
    "vanilla_renderer": {"kernel_size": 0.0, "scaling_modifier": 1.0, "white_background": False, "convert_SHs_python": False,
                         "compute_cov3D_python": False, "debug": False},
    "pcd_renderer": {"kernel_size": 0.0, "scaling_modifier": 1.0, "white_background": False, "debug": False, "convert_SHs_python": True},
}


class Camera:
    """The attributes of gaustudio.datasets.Camera that BaseRenderer.render reads (datasets/__init__.py:138-183)."""

    def __init__(self, width, height, fovx, fovy, viewmatrix, projmatrix, campos):
        self.image_height, self.image_width = height, width
        self.FoVx, self.FoVy = fovx, fovy
        self.world_view_transform, self.full_proj_transform, self.camera_center = viewmatrix, projmatrix, campos


def renderer_state(name, config):
    """What VanillaRenderer / PCDRenderer.__init__ derive from their config: (bg_color on the CPU, scaling_modifier, debug, conf)."""
    conf = {**RENDERER_DEFAULTS[name], **config}
    bg = torch.tensor([1, 1, 1], dtype=torch.float32) if conf["white_background"] else torch.tensor([0, 0, 0], dtype=torch.float32)
    return bg, conf["scaling_modifier"], conf["debug"], conf


def _sh_basis(d, D):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
    C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)
    b = [torch.full_like(x, C0)]
    if D > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if D > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if D > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def _covariance(scales, modifier, rot):
    """models/utils.py:44-97 build_covariance_from_scaling_rotation: L = R(q / |q|) diag(modifier * s), Sigma = L L^T, upper triangle."""
    q = rot / rot.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                     torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                     torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    L = R @ torch.diag_embed(modifier * scales)
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)


def properties_like_reference(renderer, conf, raw, active_sh_degree, camera):
    """-> (xyz, shs, colors_precomp, opacity, scales, rotations, cov3D_precomp) as get_gaussians_properties returns them."""
    xyz = raw["xyz"]
    if renderer == "pcd_renderer":                                   # pcd_renderer.py:24-36
        opacity = torch.ones_like(xyz, device=xyz.device)
        scales = torch.ones_like(xyz, device=xyz.device) * conf["kernel_size"]
        rotations = torch.zeros((xyz.shape[0], 4), device=xyz.device)
        rotations[:, 0] = 1
        rotations = torch.nn.functional.normalize(rotations)
        return xyz, None, raw["rgb"] / 255, opacity, scales, rotations, None
    opacity = torch.sigmoid(raw["opacity"])                          # vanilla_sg.py:58-63 with the default activations (:27-31)
    scales = rotations = cov3D_precomp = None
    if conf["compute_cov3D_python"]:
        cov3D_precomp = _covariance(torch.exp(raw["scale"]), conf["scaling_modifier"], raw["rot"])     # vanilla_sg.py:99-100
    else:
        scales = torch.exp(raw["scale"])
        if scales.shape[-1] == 2:                                     # vanilla_renderer.py:38-39
            scales = torch.cat([scales, torch.zeros_like(scales[:, :1]) + 1e-7], dim=-1)
        rotations = torch.nn.functional.normalize(raw["rot"])
    features = torch.cat((raw["f_dc"].reshape(len(raw["f_dc"]), -1, 3), raw["f_rest"].reshape(len(raw["f_dc"]), -1, 3)), dim=1)   # vanilla_sg.py:102-106
    shs = colors_precomp = None
    if conf["convert_SHs_python"]:                                   # vanilla_renderer.py:44-49
        dir_pp = xyz - camera.camera_center.repeat(features.shape[0], 1)
        dir_n = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        nc = (active_sh_degree + 1) ** 2
        sh2rgb = (_sh_basis(dir_n, active_sh_degree)[:, :, None] * features[:, :nc]).sum(1)
        colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
    else:
        shs = features
    return xyz, shs, colors_precomp, opacity, scales, rotations, cov3D_precomp


def render_like_base_renderer(props, camera, active_sh_degree, bg_color, scaling_modifier=1.0, debug=False, device="cuda",
                              Settings=None, Rasterizer=None):
    """base.py:10-63, call for call.  Settings / Rasterizer default to the drop-in module's classes."""
    if Settings is None:
        from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings as Settings, GaussianRasterizer as Rasterizer
    xyz, shs, colors_precomp, opacity, scales, rotations, cov3D_precomp = props
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:       # noqa: BLE001  (the reference's bare except: under no_grad the carrier has no grad_fn)
        pass
    raster_settings = Settings(
        image_height=int(camera.image_height), image_width=int(camera.image_width),
        tanfovx=math.tan(camera.FoVx * 0.5), tanfovy=math.tan(camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=camera.world_view_transform,
        projmatrix=camera.full_proj_transform, sh_degree=active_sh_degree if shs is not None else 1,
        campos=camera.camera_center, prefiltered=False, debug=debug)
    rasterizer = Rasterizer(raster_settings=raster_settings)
    image, radii, depth, median_map, final_opacity = rasterizer(
        means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": image, "rendered_depth": depth, "rendered_median_depth": median_map[0:1],
            "rendered_median_weight": median_map[1:2], "rendered_median_id": median_map[2:3].int(),
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "rendered_final_opacity": final_opacity, "radii": radii}


def replay_case(case, device):
    """One recorded case replayed with the recording rasterizer: -> the record in the fixture's format."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import numpy as np
    import render_call_record as rcr
    from gaustudio_amd import formats
    rec_mod = rcr.recording_module()
    raw = rcr.raw_attributes(case, device)
    # the camera the generator used: Camera(R = I, T = (0.1, -0.2, 4), FoV 60 x 40 degrees, 96 x 64)
    c = formats.CameraRecord(0, "replay", 96, 64, np.eye(3), np.array([0.1, -0.2, 4.0]), math.radians(60), math.radians(40)).cam
    # datasets/__init__.py:154-183 builds its matrices on the CPU: world_view_transform is a TRANSPOSED VIEW (non-contiguous),
    # camera_center a slice of torch.inverse's result (non-contiguous); `.to(device)` keeps those strides.  Same values here,
    # same strides: the operator must take them as they come (it calls .contiguous() on its inputs)
    view = c.viewmatrix.t().contiguous().t()
    campos = torch.inverse(view)[3][:3]
    cam = Camera(96, 64, math.radians(60), math.radians(40), view.to(device), c.projmatrix.to(device), campos.to(device))
    bg, modifier, debug, conf = renderer_state(case["renderer"], case["config"])
    with torch.set_grad_enabled(not case.get("no_grad", False)):
        props = properties_like_reference(case["renderer"], conf, raw, case["active_sh_degree"], cam)
        pkg = render_like_base_renderer(props, cam, case["active_sh_degree"], bg, modifier, debug, device=device,
                                        Settings=rec_mod.GaussianRasterizationSettings, Rasterizer=rec_mod.GaussianRasterizer)
    assert len(rec_mod.calls) == 1
    call = rec_mod.calls[0]
    call["bg_is_the_renderers_cpu_tensor"] = call.pop("_bg_obj") is bg and bg.device.type == "cpu"
    call["returns"] = rcr.describe_package(pkg)
    return call
