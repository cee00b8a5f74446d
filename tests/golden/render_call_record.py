"""TEST INFRASTRUCTURE shared by the fixture generator (dev container, real reference classes) and the tests (any box):
a RECORDING stand-in for `gaustudio_diff_gaussian_rasterization`, the renderer cases, and the seeded raw attributes of the
point clouds those cases render.  Nothing here computes an image."""
import types

import torch

# every case = one renderer configuration of the reference (renderers/vanilla_renderer.py:10-17, pcd_renderer.py:7-13) on one
# model state; `grad`: raw attributes require grad (training) / rendered under no_grad (the extraction scripts)
CASES = [
    dict(name="vanilla_train_deg2", renderer="vanilla_renderer", config={}, model="vanilla", active_sh_degree=2, grad=True),
    dict(name="vanilla_eval_deg3_no_grad", renderer="vanilla_renderer", config={}, model="vanilla", active_sh_degree=3, grad=False, no_grad=True),
    dict(name="vanilla_white_bg_modifier_debug", renderer="vanilla_renderer",
         config={"white_background": True, "scaling_modifier": 1.7, "debug": True}, model="vanilla", active_sh_degree=0, grad=True),
    dict(name="vanilla_cov3D_python", renderer="vanilla_renderer", config={"compute_cov3D_python": True, "scaling_modifier": 1.3},
         model="vanilla", active_sh_degree=1, grad=True),
    dict(name="vanilla_convert_SHs_python", renderer="vanilla_renderer", config={"convert_SHs_python": True}, model="vanilla",
         active_sh_degree=3, grad=True),
    dict(name="vanilla_2d_scales", renderer="vanilla_renderer", config={}, model="vanilla2d", active_sh_degree=3, grad=True),
    dict(name="pcd_default", renderer="pcd_renderer", config={}, model="general", active_sh_degree=None, grad=False, no_grad=True),
    dict(name="pcd_kernel_white", renderer="pcd_renderer", config={"kernel_size": 0.01, "white_background": True}, model="general",
         active_sh_degree=None, grad=False),
]

P = 64


def raw_attributes(case, device="cpu"):
    """Seeded raw (pre-activation) attributes of the case's point cloud, in the reference's storage shapes."""
    g = torch.Generator().manual_seed(77)
    r = lambda *s: torch.randn(*s, generator=g)
    if case["model"] == "general":
        raw = dict(xyz=r(P, 3), rgb=(torch.rand(P, 3, generator=g) * 255).floor(), normal=r(P, 3))
    else:
        raw = dict(xyz=r(P, 3), opacity=r(P, 1), f_dc=r(P, 1, 3), f_rest=r(P, 15, 3),
                   scale=(r(P, 2) if case["model"] == "vanilla2d" else r(P, 3)) * 0.3 - 2.0, rot=r(P, 4))
    raw = {k: v.to(device) for k, v in raw.items()}
    if case["grad"]:
        for v in raw.values():
            v.requires_grad_(True)
    return raw


def build_model(case, VanillaPointCloud, GeneralPointCloud):
    """The reference's own model class holding the seeded attributes (dev container)."""
    raw = raw_attributes(case)
    if case["model"] == "general":
        m = GeneralPointCloud({}, device="cpu")
        m.update(**raw)
        m._rgb = raw["rgb"]
        return m
    m = VanillaPointCloud({}, device="cpu")
    m.update(**raw)
    m.active_sh_degree = case["active_sh_degree"]
    return m


def describe_tensor(t, means3D=None):
    if t is None:
        return None
    d = {"shape": list(t.shape), "dtype": str(t.dtype), "requires_grad": bool(t.requires_grad), "is_leaf": bool(t.is_leaf),
         "contiguous": bool(t.is_contiguous())}
    if t.requires_grad and not t.is_leaf:
        d["retains_grad"] = bool(t.retains_grad)
    if means3D is not None:
        d["same_device_as_means3D"] = t.device == means3D.device
    return d


def describe_package(pkg):
    return {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in sorted(pkg.items())}


def recording_module():
    """A module object exporting GaussianRasterizationSettings (the real 12-field tuple of this repository == the reference's,
    tests/test_api_surface.py) and a GaussianRasterizer that records instead of rendering."""
    from gaustudio_amd.rasterizer import GaussianRasterizationSettings
    mod = types.ModuleType("gaustudio_diff_gaussian_rasterization")
    mod.calls = []

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, *args, **kw):
            rs = self.raster_settings
            m3 = kw.get("means3D")
            rec = {"positional_args": len(args), "keywords": sorted(kw), "grad_enabled": torch.is_grad_enabled(),
                   "settings": {}, "arguments": {k: describe_tensor(v, m3) for k, v in kw.items()}}
            for f in rs._fields:
                v = getattr(rs, f)
                if torch.is_tensor(v):
                    rec["settings"][f] = {"tensor": describe_tensor(v, m3 if f != "bg" else None), "values": [float(x) for x in v.flatten().tolist()] if f == "bg" else None}
                else:
                    rec["settings"][f] = {"type": type(v).__name__, "value": v}
            rec["_bg_obj"] = rs.bg
            mod.calls.append(rec)
            n = m3.shape[0]
            H, W = int(rs.image_height), int(rs.image_width)
            z = lambda *s: torch.zeros(*s, device=m3.device)
            return z(3, H, W), torch.zeros(n, dtype=torch.int32, device=m3.device), z(1, H, W), z(3, H, W), z(1, H, W)

    mod.GaussianRasterizationSettings = GaussianRasterizationSettings
    mod.GaussianRasterizer = GaussianRasterizer
    return mod


def comparable(call):
    """The part of a recorded call that must be IDENTICAL between the reference's classes on the CPU box and a replay on the
    GPU box: everything except the two fields that describe the box."""
    import copy
    c = copy.deepcopy({k: v for k, v in call.items() if k not in ("_bg_obj", "torch_factory_calls_with_device_cuda", "returns", "bg_is_the_renderers_cpu_tensor")})
    # camera_center is a 3-element SLICE of torch.inverse's result (datasets/__init__.py:182-183): non-contiguous where it was
    # created (CPU), densified by Camera.to(device) (:197-205) on its way to a GPU -- its contiguity describes the box.  (The
    # transposed-view world_view_transform keeps its strides through .to(): compared.)
    c["settings"]["campos"]["tensor"].pop("contiguous", None)
    return c
