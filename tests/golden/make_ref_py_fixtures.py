#!/usr/bin/env python
"""Generates, by EXECUTING the reference's unmodified Python classes in the dev container (tests/golden/ref_env.py):

  tests/golden/py_ply.npz            models/vanilla_sg.py:144-181 `export` + models/base.py:73-105 `load` + :102-106
                                     `get_features` + the activations of get_attribute (:58-63), on seeded clouds, through a
                                     minimal `plyfile` stand-in (one vertex element, binary little endian)
  tests/golden/py_render_calls.json  renderers/base.py:10-63 BaseRenderer.render driven through the real VanillaRenderer
                                     (vanilla_renderer.py) and PCDRenderer (pcd_renderer.py) with a RECORDING rasterizer in
                                     the place of `gaustudio_diff_gaussian_rasterization`: the settings tuple and every call
                                     argument (keyword, None-ness, shape, dtype, device relation, requires_grad, leaf-ness) and
                                     what render() returns (keys, dtypes, shapes)

    python tests/golden/make_ref_py_fixtures.py        # needs /root/reference; the fixtures travel, the reference does not
"""
import json
import math
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_env  # noqa: E402
import render_call_record as rcr  # noqa: E402


def seeded_cloud(P, M, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(xyz=r(P, 3), f_dc=r(P, 1, 3), f_rest=r(P, M - 1, 3), opacity=r(P, 1), scale=r(P, 3) * 0.3 - 2.0, rot=r(P, 4))


def make_ply_fixture():
    out = {}
    with ref_env.reference_modules():
        from gaustudio.models.vanilla_sg import VanillaPointCloud
        for tag, (P, M, seed) in {"deg3": (41, 16, 3), "deg1": (23, 4, 5)}.items():
            raw = seeded_cloud(P, M, seed)
            m = VanillaPointCloud({}, device="cpu")
            m.update(**raw)                                           # models/base.py:57-61
            names = m.construct_list_of_attributes()                  # vanilla_sg.py:161-181
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "a.ply")
                m.export(path)                                        # vanilla_sg.py:144-159
                blob = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
                rec = ref_env.PlyData.read(path)["vertex"].data
                m2 = VanillaPointCloud({}, device="cpu")
                m2.load(path)                                         # base.py:73-105
                path2 = os.path.join(d, "b.ply")
                m2.export(path2)                                      # export of LOADED (2-D) tensors: the transpose is applied again
                rec2 = ref_env.PlyData.read(path2)["vertex"].data
            for k, v in raw.items():
                out[f"{tag}_in_{k}"] = v.numpy()
            out[f"{tag}_names"] = np.array(json.dumps(names))
            out[f"{tag}_record_names"] = np.array(json.dumps(list(rec.dtype.names)))
            out[f"{tag}_records"] = np.stack([rec[n] for n in rec.dtype.names], 1)
            assert all(rec.dtype[n] == np.float32 for n in rec.dtype.names)
            out[f"{tag}_file"] = blob
            out[f"{tag}_reexport_records"] = np.stack([rec2[n] for n in rec2.dtype.names], 1)
            for k in ("xyz", "opacity", "f_dc", "f_rest", "scale", "rot"):
                out[f"{tag}_loaded_{k}"] = getattr(m2, "_" + k).numpy()
            out[f"{tag}_get_features"] = m2.get_features.numpy()          # vanilla_sg.py:102-106 (the reshape quirk)
            out[f"{tag}_features_before_export"] = m.get_features.numpy()
            for k in ("opacity", "scale", "rot", "xyz"):
                out[f"{tag}_activated_{k}"] = m2.get_attribute(k).numpy()   # vanilla_sg.py:58-63
    path = os.path.join(HERE, "py_ply.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_render_calls_fixture():
    rec_mod = rcr.recording_module()
    out = {}
    with ref_env.reference_modules(rasterizer_module=rec_mod):
        import gaustudio.renderers as R
        import gaustudio.renderers.base as base
        from gaustudio.models.vanilla_sg import VanillaPointCloud
        from gaustudio.models.general_pcd import GeneralPointCloud
        from gaustudio.datasets import Camera
        proxy = ref_env.CudaToCpuTorch()
        base.torch = proxy                           # renderers/base.py:13 asks for device="cuda": created on the CPU here, request recorded
        import gaustudio.models.utils as mutils
        mutils.torch = proxy                         # models/utils.py:54,68 (build_rotation / build_scaling_rotation) hard-code device="cuda" too
        cam = Camera(R=np.eye(3), T=np.array([0.1, -0.2, 4.0]), FoVx=math.radians(60), FoVy=math.radians(40), image_width=96, image_height=64)
        for case in rcr.CASES:
            model = rcr.build_model(case, VanillaPointCloud, GeneralPointCloud)
            renderer = R.make({"name": case["renderer"], **case["config"]})
            rec_mod.calls.clear()
            del proxy.requested[:]
            with torch.set_grad_enabled(not case.get("no_grad", False)):     # the extraction scripts render under no_grad (extract_mesh.py:97)
                pkg = renderer.render(cam, model)
            assert len(rec_mod.calls) == 1
            call = rec_mod.calls[0]
            call["bg_is_the_renderers_cpu_tensor"] = call.pop("_bg_obj") is renderer.bg_color and renderer.bg_color.device.type == "cpu"
            call["torch_factory_calls_with_device_cuda"] = [list(x) for x in proxy.requested]
            call["returns"] = rcr.describe_package(pkg)
            out[case["name"]] = call
    path = os.path.join(HERE, "py_render_calls.json")
    with open(path, "w") as f:
        json.dump({"what": "renderers/base.py:10-63 executed (unmodified) through VanillaRenderer / PCDRenderer with a recording rasterizer: "
                           "the exact settings tuple and call arguments the operator receives; generated by tests/golden/make_ref_py_fixtures.py",
                   "cases": out}, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    make_ply_fixture()
    make_render_calls_fixture()
