#!/usr/bin/env python
"""Generates tests/golden/py_sh_cov.npz by IMPORTING the reference's own Python helpers in this
container (they cannot travel to the GPU box, /root/reference does not exist there):

  /root/reference/gaustudio/utils/sh_utils.py:57-112   eval_sh   (SH -> RGB, same constants as auxiliary.h:22-38)
  /root/reference/gaustudio/models/utils.py:44-97      build_covariance_from_scaling_rotation

These are the only pieces of the reference that restate hot-path math outside the CUDA kernels
(SURVEY.md s4); the fixture pins the oracle's SH evaluation and cov3D construction against them.
models/utils.py hard-codes device='cuda' in torch.zeros; it is executed here with a torch proxy whose
zeros() drops the device argument (the source file itself is read from the reference, not copied)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/gaustudio"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gaustudio_amd import scenes  # noqa: E402


def load_sh_utils():
    spec = importlib.util.spec_from_file_location("ref_sh_utils", os.path.join(REF, "utils", "sh_utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_model_utils():
    class TorchProxy(types.ModuleType):
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def zeros(*a, device=None, **kw):
            return torch.zeros(*a, **kw)

    src = open(os.path.join(REF, "models", "utils.py")).read()
    ns = {"__name__": "ref_model_utils"}
    proxy = TorchProxy("torch")
    real_import = __import__

    def fake_import(name, *a, **kw):
        if name == "torch":
            return proxy
        return real_import(name, *a, **kw)

    ns["__builtins__"] = dict(vars(__import__("builtins")), __import__=fake_import)
    exec(compile(src, "models/utils.py", "exec"), ns)
    return ns


def main():
    shu = load_sh_utils()
    mu = load_model_utils()
    cam = scenes.make_camera(640, 480)
    sc = scenes.make_scene(4096, cam, seed=11)
    campos = torch.tensor([0.3, -0.2, 0.1])
    dirs = sc.means3D - campos[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = dict(means3D=sc.means3D.numpy(), shs=sc.shs.numpy(), campos=campos.numpy(), scales=sc.scales.numpy(),
               rotations=sc.rotations.numpy())
    shs_view = sc.shs.transpose(1, 2)                         # [P,3,16] as vanilla_renderer.py:45 builds it
    for deg in range(4):
        rgb = shu.eval_sh(deg, shs_view.double(), dirs.double())
        out[f"rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()      # vanilla_renderer.py:49
    for mod in (1.0, 1.7):
        cov = mu["build_covariance_from_scaling_rotation"](sc.scales.double(), mod, sc.rotations.double())
        out[f"cov3D_mod{mod}"] = cov.numpy()
    np.savez_compressed(os.path.join(HERE, "py_sh_cov.npz"), **out)
    print("wrote", os.path.join(HERE, "py_sh_cov.npz"), os.path.getsize(os.path.join(HERE, "py_sh_cov.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
