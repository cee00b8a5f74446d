import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from gaustudio_amd import scenes
from oracle import pyoracle as po
import ref_util
from util import scene_kwargs, oracle_forward, hip_forward, to_np
for (P, W, H, D, sig) in [(2000, 128, 96, 3, 3.0), (10000, 400, 400, 3, 1.5), (300000, 800, 800, 3, 1.5)]:
    cam = scenes.make_camera(W, H); sc = scenes.make_scene(P, cam, seed=0, sigma_px_median=sig)
    kw = scene_kwargs(sc, True, False); grads = scenes.make_output_grads(cam)
    ref = ref_util.run(sc, cam, D, kw, grads)
    os_ = oracle_forward(po, sc, cam, D, kw)
    ob = po.backward(os_, *[g.numpy() for g in grads])
    print(f"--- P={P} {W}x{H} R ref/oracle {ref['num_rendered']}/{os_['num_rendered']}  radii mismatches {(ref['radii'].numpy() != os_['radii']).sum()}")
    for k in ("color", "depth", "median", "opacity"):
        a = ref[k].numpy(); b = os_[k]
        if k == "median":
            for c, nm in enumerate(("median_depth", "median_weight", "median_id")):
                d = np.abs(a[c] - b[c]); print(f"  {nm:14s} max {d.max():.3e}  n>1e-5 {(d > 1e-5).sum()} of {d.size}")
        else:
            d = np.abs(a - b); print(f"  {k:14s} max {d.max():.3e}  n>1e-5 {(d > 1e-5).sum()} of {d.size}  mean {d.mean():.2e}")
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        a = ref[k].numpy().reshape(ob[k].shape); b = ob[k]
        d = np.abs(a - b); print(f"  {k:14s} max abs {d.max():.3e} scale {np.abs(b).max():.3e} rel {d.max()/np.abs(b).max():.2e}")
