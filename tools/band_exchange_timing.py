#!/usr/bin/env python
"""Diagnostic: host-side timeline of one banded factored step at C3 with a process group of ONE rank over RCCL (every collective call
of the N > 1 step, nothing on the wire): wall time of each hook and of exchange(), and the step with / without bands."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaustudio_amd import scenes, parallel, GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
BACKEND = os.environ.get("BAND_BACKEND", "nccl")
if BACKEND != "none":
    dist.init_process_group(BACKEND, rank=0, world_size=1, **({"device_id": dev} if BACKEND == "nccl" else {}))
    parallel.FORCE_COLLECTIVES = True
W, H, P, D = 1920, 1080, 1_000_000, 3
cam = scenes.make_camera(W, H)
sc = scenes.make_scene(P, cam, seed=0)
params = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
m2 = torch.zeros_like(params["means3D"])
rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, cam.viewmatrix.to(dev),
                                   cam.projmatrix.to(dev), D, cam.campos.to(dev), False, False)
rast = GaussianRasterizer(rs)
grads = [g.to(dev) for g in scenes.make_output_grads(cam)]
campos = cam.campos.to(dev)[None]

SEQ = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1,2,1,2').split(',')]
for bands in SEQ:
    COMPACT = {"none": False, "view": "view", "view+geometry": "view+geometry"}[os.environ.get("BAND_COMPACT", "view")]
    fx = parallel.FactoredGradExchange(params, views_per_rank=1, compact=COMPACT, bands=bands, band_split=((H + 15) // 16) // 2 if bands == 2 else None)
    log = []

    def wrap(name):
        fn = getattr(fx, name)

        def inner(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            log.append((name, (time.perf_counter() - t0) * 1e3))
            return r
        setattr(fx, name, inner)
    for n in ("_on_classes", "_on_band_ready", "_on_colors_ready", "exchange", "visible"):
        wrap(n)

    evs = []

    def step():
        t_a = time.perf_counter()
        for p in params.values():
            p.grad = None
        fx.arm(0, sh_degree=D)
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t_b = time.perf_counter()
        out = rast(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
        fx.visible(0, out[1])
        t0 = time.perf_counter()
        torch.autograd.backward([out[0], out[2], out[3], out[4]], grads)
        log.append(("backward (host)", (time.perf_counter() - t0) * 1e3))
        log.append(("arm (host)", (t_b - t_a) * 1e3))
        log.append(("forward+visible (host)", (t0 - t_b) * 1e3))
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        fx.exchange(campos, sh_degree=D)
        e2 = torch.cuda.Event(enable_timing=True); e2.record()
        evs.append((e0, e1, e2))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from gaustudio_amd import _C
    if os.environ.get("BAND_PROF") == "1":
        _C.set_profiling(1)
    N = 10
    for block in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        del log[:]
        t0 = time.perf_counter()
        for _ in range(N):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N * 1e3
        dev_fb = sum(a.elapsed_time(b) for a, b, c in evs[-N:]) / N
        dev_ex = sum(b.elapsed_time(c) for a, b, c in evs[-N:]) / N
        if os.environ.get("BAND_PROF") == "1":
            print("      stage ms:", _C.last_forward_ms(), _C.last_backward_ms(), flush=True)
        print(f"   bands={bands} block {block}: wall {dt:.3f} ms; device fwd+bwd {dev_fb:.3f}, exchange {dev_ex:.3f}", flush=True)
    agg = {}
    for k, v in log:
        agg.setdefault(k, []).append(v)
    print("   exchange sections (host ms, summed over all steps):", {k: round(v, 2) for k, v in getattr(fx, "_dbg", {}).items()})
    print(f"bands={bands}: step {dt:.3f} ms; host ms per call: " + "; ".join(f"{k} {sum(v) / len(v):.3f}" for k, v in agg.items()), flush=True)
if BACKEND != 'none':
    dist.destroy_process_group()
