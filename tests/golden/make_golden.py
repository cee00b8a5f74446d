#!/usr/bin/env python
"""Generates the golden fixtures tests/golden/ref_*.npz by running the REFERENCE's own kernels
(oracle/_ref/libgsref.so, built by oracle/build_ref.sh from /root/reference with hipify-perl) on an MI355X.

    gpurun -- 'python tests/golden/make_golden.py gpurun_out/golden'   # on the GPU box
    cp gpurun_out/golden/ref_*.npz tests/golden/                        # back in the repo

Each fixture stores the exact float32 inputs, the camera, the upstream gradients, and everything the
reference returned: the five outputs, radii, num_rendered and the eight gradient tensors.
The reference publishes no vectors of its own; these are "outputs of the reference
itself run here"."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaustudio_amd import scenes  # noqa: E402
import ref_util  # noqa: E402
from util import scene_kwargs  # noqa: E402

CASES = {
    # name: (scene builder, camera builder, D, use_sh, use_cov, scale_modifier, bg)
    "ref_a_sh3": dict(P=2000, W=128, H=96, D=3, use_sh=True, use_cov=False, mod=1.0, bg=(0, 0, 0), sig=3.0, seed=0),
    "ref_b_precomp_whitebg": dict(P=3000, W=160, H=100, D=0, use_sh=False, use_cov=True, mod=1.0, bg=(1, 1, 1), sig=2.5, seed=1),
    "ref_c_ring_sh2_mod": dict(P=4000, W=200, H=120, D=2, use_sh=True, use_cov=False, mod=1.5, bg=(0, 0, 0), sig=None, seed=2),
}


def build(c):
    if c["sig"] is None:
        sc = scenes.make_ball_scene(c["P"], radius=3.0, seed=c["seed"], sigma=0.05)
        cam = scenes.ring_cameras(5, c["W"], c["H"], radius=8.0)[1]
    else:
        cam = scenes.make_camera(c["W"], c["H"])
        sc = scenes.make_scene(c["P"], cam, seed=c["seed"], sigma_px_median=c["sig"])
    return sc, cam


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    for name, c in CASES.items():
        sc, cam = build(c)
        kw = scene_kwargs(sc, c["use_sh"], c["use_cov"])
        grads = scenes.make_output_grads(cam, seed=7)
        bg = torch.tensor(c["bg"], dtype=torch.float32)
        ref = ref_util.run(sc, cam, c["D"], kw, grads, scale_modifier=c["mod"], bg=bg)
        d = dict(means3D=sc.means3D.numpy(), opacities=sc.opacities.numpy(), D=np.int32(c["D"]),
                 scale_modifier=np.float32(c["mod"]), bg=bg.numpy(), width=np.int32(cam.width),
                 height=np.int32(cam.height), tanfovx=np.float64(cam.tanfovx), tanfovy=np.float64(cam.tanfovy),
                 viewmatrix=cam.viewmatrix.numpy(), projmatrix=cam.projmatrix.numpy(), campos=cam.campos.numpy(),
                 grad_color=grads[0].numpy(), grad_depth=grads[1].numpy(), grad_median=grads[2].numpy(),
                 grad_opacity=grads[3].numpy())
        for k, v in kw.items():
            d["in_" + k] = v.numpy()
        for k, v in ref.items():
            d["ref_" + k] = np.asarray(v) if not torch.is_tensor(v) else v.numpy()
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, "R =", ref["num_rendered"], os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/golden")
