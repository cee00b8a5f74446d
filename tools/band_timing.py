#!/usr/bin/env python
"""Where does a banded backward spend its time?  C3, through the C ABI (tests/util.hip_backward_raw): the one-call backward against
band FIRST + band SECOND + SH, each stage timed with a device synchronisation around it (diagnostic, not a benchmark)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaustudio_amd import scenes
from util import hip_forward, hip_backward_raw, scene_kwargs

cam = scenes.make_camera(1920, 1080)
sc = scenes.make_scene(1_000_000, cam, seed=0)
kw = {k: v.cuda() for k, v in scene_kwargs(sc, True, False).items()}
sc = scenes.Scene(*[t.cuda() for t in sc])
grads = [g.cuda() for g in scenes.make_output_grads(cam)]
hs = hip_forward(sc, cam, 3, kw)
gy = (cam.height + 15) // 16
S = gy // 2


def timed(label, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = hip_backward_raw(hs, sc, cam, 3, kw, grads, options={}, **k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"{label:28s} {dt:8.3f} ms (host wall, includes the inspect kernel + allocations)", flush=True)
    return out


for rep in range(3):
    timed("one call (parts 3)")
    a = timed("band FIRST (1|16)", parts=1 | 16, sh_g0=S)
    a = timed("band SECOND (1|32)", parts=1 | 32, sh_g0=S, reuse=a)
    a = timed("SH (2)", parts=2, reuse=a)
