#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native Gaussian rasterizer.

Metric (BASELINE.json): Mpixels/s forward+backward at 1 M Gaussians, 1920x1080 (config C3), per-kernel
fraction of the HBM roofline, 1 -> 8 GPU scaling.  A "step" = one forward + one backward of the
operator over one camera per GPU (through GaussianRasterizer / the autograd Function, i.e. the path
gaustudio/renderers/base.py takes), inputs resident in HBM; for N > 1 each rank renders its own camera
of the replicated scene and the step ends with ONE RCCL all-reduce of the flat per-Gaussian gradient
buffer (gaustudio_amd/parallel.py).  value = N * H * W / step_time.

    python bench.py                                  # N=1, C3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the dominant north-star kernel (composite_fwd):
algorithmic bytes R*44 + T*8 + H*W*32 (+ H*W*8 training aux) over its mean duration,
measured with HIP events on the launch stream during the timed steps.  `cpu_baseline` is the CPU
oracle (a C restatement of the reference kernels -- the reference itself has no CPU renderer) timed on
the host cores on a bounded sample of the same frame.
"""
import argparse
import json
import math
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL needs this for N > 1 (already exported on the GPU boxes; kept here so a
# bare `torch.distributed.run bench.py` from another shell behaves the same)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (P, W, H, SH degree, description)
    "C1": (10_000, 400, 400, 0, "C1: 10k synthetic Gaussians, 400x400, SH degree 0, fwd+bwd"),
    "C2": (300_000, 800, 800, 3, "C2: 300k synthetic Gaussians (lego stand-in), 800x800, SH degree 3, fwd+bwd"),
    "C3": (1_000_000, 1920, 1080, 3,
           "C3: 1M synthetic Gaussians, 1920x1080, SH degree 3, fwd+bwd (colour+depth+median+opacity consumed)"),
    "C3D0": (1_000_000, 1920, 1080, 0,
             "C3 at SH degree 0: 1M synthetic Gaussians, 1920x1080, fwd+bwd (SURVEY s8d names D = 0 and D = 3)"),
    # per-GPU shares of the multi-GPU configurations (one view per GPU)
    "C4": (5_000_000, 1297, 840, 3, "C4 share: 5M synthetic Gaussians (garden stand-in), one 1297x840 view, SH degree 3, fwd+bwd"),
    "C5": (2_500_000, 3840, 2160, 3, "C5 share: 2.5M synthetic Gaussians (Truck stand-in), one 3840x2160 view, SH degree 3, fwd+bwd"),
}


def rank_camera(scenes, W, H, rank, world):
    """Rank r looks down +z from the origin, yawed by 3 degrees per rank around the scene's axis so every
    rank renders a different view of the same (replicated) Gaussians with the same amount of work."""
    a = math.radians(3.0) * (rank - (world - 1) / 2.0)
    R = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    return scenes.make_camera(W, H, R=R)


def cpu_baseline(sc, cam, D, grads, budget_s=20.0):
    """Times the CPU oracle (fwd+bwd) on every `tile_step`-th tile of the same frame; per-Gaussian
    stages run in full.  tile_step is chosen from a pilot so that the run takes roughly budget_s."""
    from oracle import pyoracle as po   # test infrastructure, used here ONLY as the timed CPU baseline
    po.build()
    threads = po.num_threads()
    kw = dict(sh_degree=D, shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
    args = (sc.means3D.numpy(), sc.opacities.numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
            cam.campos.numpy(), cam.width, cam.height, cam.tanfovx, cam.tanfovy)
    g = [t.numpy() for t in grads]

    def run(step):
        t0 = time.perf_counter()
        st = po.forward(*args, tile_step=step, **kw)
        po.backward(st, *g, tile_step=step, want_abs=False)
        return time.perf_counter() - t0

    pilot_step = 32
    t_pilot = run(pilot_step)
    step = 1
    while step < pilot_step and t_pilot * pilot_step / step > budget_s:
        step *= 2
    t = run(step) if step != pilot_step else t_pilot
    T = ((cam.width + 15) // 16) * ((cam.height + 15) // 16)
    tiles = len(range(0, T, step))
    pixels = cam.width * cam.height * tiles / T
    return {"value": round(pixels / t / 1e6, 4), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "seconds": round(t, 2),
            "sample": f"CPU restatement of the reference kernels (oracle/gsr_oracle.c, OpenMP x{threads}): fwd+bwd on "
                      f"every {step}th 16x16 tile of the same frame ({tiles}/{T} tiles), per-Gaussian stages in full"}


def stage_roofline(P, P_vis, R, T, H, W, D, fwd_ms, bwd_ms, fwd_only):
    """Achieved algorithmic GB/s of every stage against the 8 TB/s HBM roofline (BASELINE.md s4 byte counts; the
    records are 64 B here instead of the reference's 48 B of SoA fields, counted as written).  R = instances binned."""
    sh = 12 * (D + 1) ** 2
    HW = H * W
    stages = {}
    if fwd_ms:
        b = {"preprocess": P * (12 + 12 + 16 + 4 + sh) + P_vis * (64 + 4 + 4),
             "scatter": R * 8 + P_vis * 64, "sort": R * 12,
             "composite": R * 44 + T * 8 + HW * 32 + (0 if fwd_only else HW * 8)}
        for k, nbytes in b.items():
            if fwd_ms.get(k):
                gbs = nbytes / (fwd_ms[k] * 1e-3) / 1e9
                stages[k + "_fwd" if k in ("preprocess", "composite") else k] = {
                    "ms": round(fwd_ms[k], 4), "algorithmic_bytes": nbytes, "GB/s": round(gbs, 1), "frac_hbm": round(gbs / 8000.0, 4)}
    if bwd_ms:
        b = {"composite_bwd": R * 44 + HW * 32 + R * 48,
             "preprocess_bwd": R * 48 + P_vis * (12 + 4 + sh + 64 + 12 + 16 + 8) + P * (12 + 4 + 12 + 12 + 24 + 12 * 16 + 12 + 16)}
        for k, nbytes in b.items():
            if bwd_ms.get(k):
                gbs = nbytes / (bwd_ms[k] * 1e-3) / 1e9
                stages[k] = {"ms": round(bwd_ms[k], 4), "algorithmic_bytes": nbytes, "GB/s": round(gbs, 1),
                             "frac_hbm": round(gbs / 8000.0, 4)}
    return stages


def committed_counters(workload):
    """Per-kernel PMC measurements committed under profiles/ (newest round first): HBM traffic and VALU instruction
    counts per launch.  PMC counters cannot be read in-process; they are recaptured with rocprofv3 whenever a kernel
    changes (tools/profile_session.sh) and the file records the commit-time kernel durations they belong to."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        t = json.load(open(f))
        if t.get("workload") == workload:
            t["_file"] = os.path.relpath(f, ROOT)
            # do the committed counters belong to the kernels being run?  (fingerprint written by tools/make_traffic_json.py)
            import hashlib
            h = hashlib.sha1()
            try:
                for src in ("gsr_kernels_fwd.hip", "gsr_kernels_bwd.hip", "gsr_common.h", "Makefile"):
                    h.update(open(os.path.join(ROOT, "gaustudio_amd", "csrc", src), "rb").read())
                t["_matches_kernel_sources"] = t.get("_kernel_sources_sha1") == h.hexdigest()
            except OSError:
                t["_matches_kernel_sources"] = None
            return t
    return None


def valu_issue(counters, kernel, ms):
    """VALU issue fraction of a kernel: (wave64 VALU instructions per launch from the committed SQ_INSTS_VALU pass) x
    (mean cycles per instruction of that kernel's instruction mix, from the issue-rate probe: profiles/r01_issue_rates.txt,
    DESIGN.md s4) / (SIMD-cycles available in the measured duration: 1024 SIMDs x ~2.3 GHz sustained)."""
    if not counters or kernel not in counters or not ms or "valu_insts" not in counters[kernel]:
        return None
    k = counters[kernel]
    cpi = k.get("valu_cycles_per_inst_model", 3.0)
    avail = ms * 1e-3 * 2.3e9 * 1024
    return {"valu_insts": k["valu_insts"], "cycles_per_inst_model": cpi, "issue_frac": round(k["valu_insts"] * cpi / avail, 3),
            "source": counters.get("_file")}


def list_histogram(ranges):
    n = (ranges[:, 1] - ranges[:, 0]).float()
    q = torch.quantile(n, torch.tensor([0.5, 0.9, 0.99], device=n.device)).tolist()
    return {"mean": round(float(n.mean()), 1), "p50": int(q[0]), "p90": int(q[1]), "p99": int(q[2]), "max": int(n.max()),
            "empty_tiles": int((n == 0).sum())}


def reference_ab(sc, cam, D, grads_cpu, dev, steps=5):
    """Optional GPU A/B: the reference's OWN kernels (hipified test-only into oracle/_ref/libgsref.so by
    oracle/build_ref.sh; checker infrastructure, never part of the product) timed on the same inputs, including the
    zero-fills the reference's torch glue performs (rasterize_points.cu:68-81,160-169)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import ref_util
        if not ref_util.available():
            return None
        import ctypes
        L = ref_util.lib()
        h = ctypes.c_void_p(L.ref_create())
        p, cf = ref_util._p, ctypes.c_float
        P, W, H = sc.means3D.shape[0], cam.width, cam.height
        t = {k: getattr(sc, k).to(dev).contiguous() for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        view, proj, cpos = cam.viewmatrix.to(dev).contiguous(), cam.projmatrix.to(dev).contiguous(), cam.campos.to(dev)
        bgd = torch.zeros(3, device=dev)
        grads = [g.to(dev) for g in grads_cpu]
        fo = dict(dtype=torch.float32, device=dev)

        def step():
            out = [torch.empty((3, H, W), **fo), torch.empty((1, H, W), **fo), torch.empty((3, H, W), **fo), torch.empty((1, H, W), **fo)]
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            R = L.ref_forward(h, P, D, 16, p(bgd), W, H, p(t["means3D"]), p(t["shs"]), p(None), p(t["opacities"]), p(t["scales"]),
                              cf(1.0), p(t["rotations"]), p(None), p(view), p(proj), p(cpos), cf(cam.tanfovx), cf(cam.tanfovy), 0,
                              *[p(o) for o in out], p(radii))
            z = lambda *s: torch.zeros(*s, **fo)
            G = [z(P, 3), z(P, 4), z(P, 1), z(P, 3), z(P), z(P, 3), z(P, 6), z(P, 16, 3), z(P, 3), z(P, 4)]
            rc = L.ref_backward(h, P, D, 16, p(bgd), W, H, p(t["means3D"]), p(t["shs"]), p(None), p(t["scales"]), cf(1.0),
                                p(t["rotations"]), p(None), p(view), p(proj), p(cpos), cf(cam.tanfovx), cf(cam.tanfovy), p(radii),
                                *[p(g) for g in grads], *[p(g) for g in G])
            assert R > 0 and rc == 0
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        L.ref_destroy(h)
        return round(ms, 3)
    except Exception as e:  # the A/B is optional: never let it break the benchmark line
        return f"unavailable: {type(e).__name__}: {e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--fwd-only", action="store_true", help="time the no_grad forward only (inference paths)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-ab", action="store_true", help="skip timing the hipified reference kernels (optional A/B)")
    ap.add_argument("--overlap-chunks", type=int, default=4,
                    help="N > 1: SH-gradient ranges reduced from inside the backward (0/1 = one all-reduce after it)")
    ap.add_argument("--traffic", type=float, default=None,
                    help="measured HBM bytes per composite_fwd launch from a rocprofv3 --pmc pass; default: the "
                         "committed measurement in profiles/r*_traffic.json for this workload, else null")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    # GSR_BENCH_BACKEND=gloo + fewer GPUs than ranks is a plumbing test of the N > 1 path on a 1-GPU box
    # (ranks share device 0, the collective goes through gloo); the measured configuration is nccl = RCCL.
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from gaustudio_amd import _C, parallel, runtime, scenes
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    # one-time process initialisation (not a benchmark step): load the gfx950 code objects with a 512-Gaussian
    # call and reserve an allocator pool, as a long-running trainer / server would at start-up
    runtime.warm_start(dev)

    P, W, H, D, desc = WORKLOADS[a.workload]
    cam0 = scenes.make_camera(W, H)
    sc = scenes.make_scene(P, cam0, seed=0)                 # identical on every rank (replicated parameters)
    cam = rank_camera(scenes, W, H, rank, world)
    grads_cpu = scenes.make_output_grads(cam, seed=1)

    params = {k: getattr(sc, k).to(dev).requires_grad_(not a.fwd_only)
              for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=not a.fwd_only)
    grads = [g.to(dev) for g in grads_cpu]
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), D, cam.campos.to(dev),
                                       False, False)
    rasterizer = GaussianRasterizer(rs)
    bucket = None if a.fwd_only else parallel.FlatGradBucket(list(params.values()), roles=params)
    state = {}

    comm_ev = []       # (backward enqueued, reduction finished) events of the timed steps, N > 1 only

    def step(timed=False):
        if a.fwd_only:
            with torch.no_grad():
                out = rasterizer(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                 shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
            state["out"] = out
            return
        for p in params.values():
            p.grad = None
        means2D.grad = None
        if world > 1:
            # gradients are born in the flat all-reduce buffer; the SH ranges are reduced while the backward runs
            bucket.arm(a.overlap_chunks)
        color, radii, depth, median, opac = rasterizer(
            means3D=params["means3D"], means2D=means2D, opacities=params["opacities"], shs=params["shs"],
            scales=params["scales"], rotations=params["rotations"])
        torch.autograd.backward([color, depth, median, opac], grads)
        if world > 1 and timed:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        parallel.allreduce_gaussian_grads(bucket)          # the tail of the one logical reduction (no-op at N=1)
        if world > 1 and timed:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            comm_ev.append((e0, e1))
        state["out"] = (color, radii, depth, median, opac)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    _C.set_profiling(True)                                   # HIP events on the launch stream, no host syncs
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(timed=True)
    barrier()
    dt = time.perf_counter() - t0
    fwd_ms = _C.last_forward_ms()
    bwd_ms = _C.last_backward_ms()
    _C.set_profiling(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # instance counts of this rank's view (they drive every composite-stage byte count): the reference-defined
    # num_rendered and the instances actually binned; per-tile list lengths
    with torch.no_grad():
        out = _C.rasterize_gaussians(rs.bg, params["means3D"], torch.Tensor([]), params["opacities"], params["scales"],
                                     params["rotations"], 1.0, torch.Tensor([]), rs.viewmatrix, rs.projmatrix,
                                     rs.tanfovx, rs.tanfovy, H, W, params["shs"], D, rs.campos, False, False)
        R_ref = out[0]
        counts = _C.inspect_counts(out[8], W, H)
        R = counts["num_binned"]
        _, ranges = _C.inspect_binning(out[7], out[8], R, W, H)
        lists = list_histogram(ranges)
        vis = int((state["out"][1] > 0).sum().item())
        del out

    if rank == 0:
        T = ((W + 15) // 16) * ((H + 15) // 16)
        ms_per_step = dt / a.steps * 1e3
        value = world * H * W / (dt / a.steps) / 1e6
        comp_ms = fwd_ms["composite"] if fwd_ms else None
        alg_bytes = R * 44 + T * 8 + H * W * 32 + (0 if a.fwd_only else H * W * 8)
        counters = committed_counters(a.workload)
        traffic = a.traffic
        if traffic is None and counters and "composite_fwd" in counters:
            traffic = counters["composite_fwd"].get("traffic_bytes")
        roof = None
        if comp_ms:
            achieved = alg_bytes / (comp_ms * 1e-3) / 1e9
            roof = {"kernel": "composite_fwd", "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                    "traffic_upper": (counters or {}).get("composite_fwd", {}).get("traffic_upper_bytes"),
                    "traffic_counters_match_kernel_sources": (counters or {}).get("_matches_kernel_sources"),
                    "algorithmic_bytes": alg_bytes, "avg_ms": round(comp_ms, 4),
                    "algorithmic_bytes_with_reference_R": R_ref * 44 + T * 8 + H * W * 32 + (0 if a.fwd_only else H * W * 8),
                    "valu": valu_issue(counters, "composite_fwd", comp_ms),
                    "note": "algorithmic bytes use the instances actually staged (tight binning); `traffic` = FETCH_SIZE + "
                            "WRITE_SIZE of the committed PMC passes with the access-pattern calibration of "
                            "profiles/r02_fetch_write_calibration.txt (`traffic_upper` = 2*FETCH + WRITE); the kernel is "
                            "VALU-issue bound (256 pixel evaluations per staged 48-B record): `valu.issue_frac` is the "
                            "fraction of the SIMDs' issue cycles its VALU instructions fill, see DESIGN.md s4"}
        line = {
            "metric": "Mpixels/s fwd+bwd @1M Gaussians 1920x1080" if a.workload == "C3" and not a.fwd_only
                      else f"Mpixels/s {'fwd' if a.fwd_only else 'fwd+bwd'} @{a.workload}",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "gaussians": P, "visible": vis, "width": W, "height": H, "sh_degree": D,
                       "num_rendered": R_ref, "instances_binned": R, "tiles": T, "tile_list_length": lists,
                       "views_per_step": world,
                       "parallelism": f"one camera per GPU x{world}, 1 logical RCCL all-reduce of "
                                      f"{0 if bucket is None else bucket.nbytes} B/rank "
                                      f"({a.overlap_chunks if a.overlap_chunks > 1 else 1} chunk(s) overlapped with the backward)"
                                      if world > 1 else "single GPU"},
            "stage_ms": {"forward": fwd_ms, "backward": bwd_ms},
            "stage_roofline": stage_roofline(P, vis, R, T, H, W, D, fwd_ms, bwd_ms, a.fwd_only),
            "roofline": roof,
        }
        if bwd_ms:
            line["valu_composite_bwd"] = valu_issue(counters, "composite_bwd", bwd_ms.get("composite_bwd"))
        if world > 1 and comm_ev:
            torch.cuda.synchronize()
            exposed = sum(e0.elapsed_time(e1) for e0, e1 in comm_ev) / len(comm_ev)
            comp_total = sum(v for k, v in (fwd_ms or {}).items() if k != "calls") + sum(v for k, v in (bwd_ms or {}).items() if k != "calls")
            line["comm"] = {"payload_bytes_per_rank": bucket.nbytes, "chunks_per_step": bucket.stats["chunks"] / max(1, a.steps + a.warmup),
                            "chunk_bytes_per_step": bucket.stats["chunk_bytes"] / max(1, a.steps + a.warmup),
                            "tail_bytes_per_step": bucket.stats["tail_bytes"] / max(1, a.steps + a.warmup),
                            "compute_ms": round(comp_total, 4), "comm_exposed_ms": round(exposed, 4),
                            "note": "compute_ms = sum of the operator's kernel stages (HIP events); comm_exposed_ms = time from the "
                                    "end of the backward's enqueue to the end of the reduction on rank 0's stream (what the "
                                    "collective adds to the step after overlap)"}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sc, cam, D, grads_cpu)
        else:
            line["cpu_baseline"] = None
        if world == 1 and not a.no_ref_ab and not a.fwd_only:
            line["reference_hipified_ms"] = reference_ab(sc, cam, D, grads_cpu, dev)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
