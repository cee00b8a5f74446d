"""Diagnostic: where the waves of composite_bwd spend their life at C3 (s_memtime per phase, per wave).
Needs the instrumented build:  make -C gaustudio_amd/csrc BWD_EXTRA=-DGSR_BWD_TIMING  (rebuild without it afterwards);
result of round 3: profiles/r03_composite_bwd_phases.txt."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustudio_amd import scenes
from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaustudio_amd", "libgsrast.so"))
dev = "cuda"
cam = scenes.make_camera(1920, 1080)
sc = scenes.make_scene(1_000_000, cam, seed=0)
rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                   cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
P = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
grads = [g.to(dev) for g in scenes.make_output_grads(cam, seed=4)]
def step():
    out = GaussianRasterizer(rs)(means3D=P["means3D"], means2D=torch.zeros_like(P["means3D"], requires_grad=True), opacities=P["opacities"],
                                 shs=P["shs"], scales=P["scales"], rotations=P["rotations"])
    torch.autograd.backward([out[0], out[2], out[3], out[4]], grads)
for _ in range(3): step()
import numpy as np
SL = 40000
buf = (ctypes.c_ulonglong * (SL * 12))()
assert lib.gsr_debug_bwd_phase_ticks(buf, 1) == 0
N = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): step()
e1.record(); torch.cuda.synchronize()
print("ms per fwd+bwd step", e0.elapsed_time(e1) / N)
assert lib.gsr_debug_bwd_phase_ticks(buf, 0) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(SL, 12).astype(np.float64) / N
live = a.sum(1) > 0
names = ["pixel state arrives", "first barrier (bmax)", "zero-rows loop", "(first ids arrive / loop tail -> top)", "wait top-of-round barrier", "staging LDS writes (wait records)",
         "issue next loads + zero planes + list init", "wait second barrier", "median + list building", "walk", "wait before flush", "flush"]
tot = a.sum()
for k, n in enumerate(names):
    print(f"{n:45s} {a[:, k].sum():16.0f} {100.0 * a[:, k].sum() / tot:6.1f} %   per wave {a[live, k].mean():9.0f}")
print("waves", int(live.sum()), "ticks per wave", tot / live.sum())
