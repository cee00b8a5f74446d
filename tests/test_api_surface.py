"""CPU (-m "not gpu"): the drop-in boundary.  Interface facts are taken from the reference source text
($RAST/gaustudio_diff_gaussian_rasterization/__init__.py, rasterize_points.h, rasterizer.h)."""
import ctypes
import inspect
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_import_name_and_public_names():
    import gaustudio_diff_gaussian_rasterization as g
    for name in ("GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_RasterizeGaussians", "_C"):
        assert hasattr(g, name), name
    for fn in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):
        assert callable(getattr(g._C, fn))


def test_settings_tuple_field_order():
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings as S
    assert S._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                         "projmatrix", "sh_degree", "campos", "prefiltered", "debug")      # __init__.py:160-172
    s = S(4, 5, 0.1, 0.2, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3, torch.zeros(3), False, True)
    assert s.image_width == 5 and s.debug is True


def test_operator_signatures_match_reference():
    import gaustudio_diff_gaussian_rasterization as g
    fwd = list(inspect.signature(g.GaussianRasterizer.forward).parameters)
    assert fwd == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]
    assert list(inspect.signature(g.rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp", "raster_settings"]
    # the reference's positional parameters (rasterize_points.h:18-65), then optional trailing ones a reference-style positional
    # call never reaches: `options` (the per-call options of include/gsrast.h gsr_options) and, for the backward, the image size
    # (only needed when every upstream gradient is absent -- round 5: an absent gradient is zero and is not read)
    def positional(fn):
        ps = inspect.signature(fn).parameters
        names = list(ps)
        i = names.index("options")
        assert ps["options"].default is None and names[i + 1:] in ([], ["image_height", "image_width"])
        assert all(ps[n].default == -1 for n in names[i + 1:])
        return names[:i]
    assert positional(g._C.rasterize_gaussians) == [
        "background", "means3D", "colors", "opacity", "scales", "rotations", "scale_modifier", "cov3D_precomp", "viewmatrix",
        "projmatrix", "tan_fovx", "tan_fovy", "image_height", "image_width", "sh", "degree", "campos", "prefiltered", "debug"]
    assert positional(g._C.rasterize_gaussians_backward) == [
        "background", "means3D", "radii", "colors", "scales", "rotations", "scale_modifier", "cov3D_precomp", "viewmatrix",
        "projmatrix", "tan_fovx", "tan_fovy", "dL_dout_color", "dL_dout_depth", "dL_dout_median_depth",
        "dL_dout_final_opacity", "sh", "degree", "campos", "geomBuffer", "R", "binningBuffer", "imageBuffer", "debug"]
    assert list(inspect.signature(g._C.mark_visible).parameters) == ["means3D", "viewmatrix", "projmatrix"]
    assert hasattr(g.GaussianRasterizer, "markVisible")


def _rasterizer():
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings as S, GaussianRasterizer
    return GaussianRasterizer(S(8, 8, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False))


def test_input_validation_messages():
    r = _rasterizer()
    m, o = torch.zeros(2, 3), torch.zeros(2, 1)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(m, m, o, scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), colors_precomp=torch.zeros(2, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=torch.zeros(2, 1, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4), cov3D_precomp=torch.zeros(2, 6))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3))


def test_no_cpu_fallback_and_shape_error():
    """The product path must fail loudly without a ROCm device: no oracle, no torch fallback."""
    r = _rasterizer()
    m, o = torch.zeros(2, 3), torch.zeros(2, 1)
    with pytest.raises(RuntimeError, match="ROCm devices only"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):     # rasterize_points.cu:57-59
        r(torch.zeros(2, 4), m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    with pytest.raises(RuntimeError, match="ROCm devices only"):
        r.markVisible(m)


def test_a_forward_that_raises_early_leaves_no_stale_grad_mode_note(monkeypatch):
    """the wrappers note the caller's grad mode in a one-shot thread-local that Function.forward consumes.  A forward that
    raises BEFORE consuming it (here: the options lookup fails) under torch.no_grad() must not leave `False` behind -- the next direct
    `_RasterizeGaussians.apply` on this thread would silently become forward_only and its backward be refused."""
    import sys
    from gaustudio_amd import rasterizer as gr
    go = sys.modules["gaustudio_amd.options"]          # (the package re-exports the class `options` under the module's name)
    r = _rasterizer()
    m, o = torch.zeros(2, 3), torch.zeros(2, 1)

    def boom(_needs):
        raise RuntimeError("options lookup failed")
    monkeypatch.setattr(gr, "_options_for_forward", boom)
    with torch.no_grad(), pytest.raises(RuntimeError, match="options lookup failed"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    assert go.take_grad_mode() is True          # nothing noted: a direct Function.apply counts as "grad enabled"
    monkeypatch.undo()
    with torch.no_grad(), pytest.raises(RuntimeError, match="ROCm devices only"):     # consumed inside: nothing left either
        r(m, m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    assert go.take_grad_mode() is True


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gaustudio_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src.replace("the CPU oracle", "").replace("CPU oracle", "").replace("the oracle", "") \
                    or "import" not in "".join(l for l in src.splitlines() if "oracle" in l and "import" in l), f
    for f in ("gaustudio_amd/_C.py", "gaustudio_amd/rasterizer.py", "gaustudio_amd/parallel.py", "gaustudio_amd/__init__.py",
              "gaustudio_diff_gaussian_rasterization/__init__.py"):
        src = open(os.path.join(ROOT, f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_c_abi_library_exports_every_declared_symbol():
    """libgsrast.so loads without a GPU and exports exactly what include/gsrast.h declares."""
    hdr = open(os.path.join(ROOT, "include", "gsrast.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", hdr)) - {"gsr_alloc_fn"}
    assert {"gsr_forward", "gsr_backward", "gsr_mark_visible", "gsr_abi_version", "gsr_last_error",
            "gsr_backward_scratch_bytes", "gsr_inspect_geometry", "gsr_inspect_binning", "gsr_inspect_image",
            "gsr_inspect_backward_sums", "gsr_set_profiling", "gsr_last_forward_ms", "gsr_last_backward_ms"} <= names
    from gaustudio_amd import _C
    L = _C.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gsrast.h but not exported by libgsrast.so"
    assert L.gsr_abi_version() == 6
    L.gsr_backward_scratch_bytes.restype = ctypes.c_size_t
    assert L.gsr_backward_scratch_bytes(ctypes.c_int(1000), ctypes.c_int(5000)) >= 5000 * 49


@pytest.mark.skipif(not os.path.isdir("/root/reference/gaustudio/renderers"), reason="reference checkout not present")
def test_reference_renderers_load_our_op_unchanged():
    """`gaustudio/renderers` (unmodified, imported from the read-only reference checkout) must import and
    register against our package.  plyfile/trimesh/open3d/skimage are not installed here and are irrelevant
    to the operator: they are stubbed."""
    import types

    class _Any:
        """Attribute sink for the stubbed third-party packages (annotations like o3d.geometry.PointCloud)."""
        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Any()

    stubs = {}
    for name in ("plyfile", "trimesh", "open3d", "skimage", "skimage.measure", "omegaconf"):
        if name not in sys.modules:
            stubs[name] = _Stub(name)
    sys.modules.update(stubs)
    sys.path.insert(0, "/root/reference")
    try:
        import gaustudio.renderers as R
        import gaustudio.renderers.base as base
        import gaustudio_diff_gaussian_rasterization as ours
        assert base.GaussianRasterizer is ours.GaussianRasterizer
        assert base.GaussianRasterizationSettings is ours.GaussianRasterizationSettings
        assert "vanilla_renderer" in R.renderers and "pcd_renderer" in R.renderers
        r = R.make({"name": "vanilla_renderer"}) if hasattr(R, "make") else None
        if r is not None:
            assert r.bg_color.device.type == "cpu"          # the CPU `bg` our op has to accept
    finally:
        sys.path.remove("/root/reference")
        for k in list(sys.modules):
            if k == "gaustudio" or k.startswith("gaustudio."):
                del sys.modules[k]
        for k in stubs:
            sys.modules.pop(k, None)


def test_replay_of_the_renderer_calls_equals_the_recording():
    """tests/caller_replay.py (the hand-written restatement of renderers/base.py:10-63 + get_gaussians_properties that the GPU
    tests drive the operator with) against tests/golden/py_render_calls.json (what the UNMODIFIED classes hand the operator,
    recorded in the dev container): identical records for all cases, on the CPU."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import caller_replay
    import render_call_record as rcr
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "py_render_calls.json")))["cases"]
    assert sorted(fixture) == sorted(c["name"] for c in rcr.CASES) and len(fixture) == 8
    for case in rcr.CASES:
        got = caller_replay.replay_case(case, "cpu")
        want = fixture[case["name"]]
        assert json.loads(json.dumps(rcr.comparable(got))) == rcr.comparable(want), case["name"]
        assert got["returns"] == want["returns"]
    # facts of the recording the operator depends on (SURVEY.md s8b "Device conventions", "Absent-input convention")
    v = fixture["vanilla_train_deg2"]
    assert v["positional_args"] == 0 and v["settings"]["bg"]["tensor"]["shape"] == [3] and v["bg_is_the_renderers_cpu_tensor"]
    assert v["settings"]["viewmatrix"]["tensor"]["contiguous"] is False          # a transposed view: the op must .contiguous() it
    assert v["arguments"]["means2D"]["is_leaf"] is False and v["arguments"]["means2D"]["retains_grad"] is True
    assert v["arguments"]["colors_precomp"] is None and v["arguments"]["cov3D_precomp"] is None
    p = fixture["pcd_default"]
    assert p["arguments"]["opacities"]["shape"] == [64, 3] and p["arguments"]["shs"] is None and p["settings"]["sh_degree"]["value"] == 1


@pytest.mark.skipif(not os.path.isdir("/root/reference/gaustudio/renderers"), reason="reference checkout not present")
def test_recorded_renderer_calls_are_current():
    """Re-runs the recording against the reference checkout (dev container) and compares with the committed fixture."""
    import json
    import subprocess
    import tempfile
    gold = os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as d:
        code = ("import sys, os; sys.path.insert(0, %r); import make_ref_py_fixtures as m; m.HERE = %r; m.make_render_calls_fixture(); m.make_ply_fixture()" % (gold, d))
        subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, timeout=300)
        assert json.load(open(os.path.join(d, "py_render_calls.json"))) == json.load(open(os.path.join(gold, "py_render_calls.json")))
        import numpy as np
        a, b = np.load(os.path.join(d, "py_ply.npz")), np.load(os.path.join(gold, "py_ply.npz"))
        assert sorted(a.files) == sorted(b.files) and all(np.array_equal(a[k], b[k]) for k in a.files)
