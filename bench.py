#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native Gaussian rasterizer.

Metric (BASELINE.json): Mpixels/s forward+backward at 1 M Gaussians, 1920x1080 (config C3), per-kernel
fraction of the HBM roofline, 1 -> 8 GPU scaling.  A "step" = one forward + one backward of the
operator over `--views-per-rank` cameras per GPU (default 1; through GaussianRasterizer / the autograd Function,
i.e. the path gaustudio/renderers/base.py takes), inputs resident in HBM; for N > 1 each rank renders its own
cameras of the replicated scene and the step ends with ONE logical gradient exchange (`--exchange dense`: one RCCL
all-reduce of the flat 236 B/Gaussian buffer; `--exchange factored`, the default at N > 1: all-gather of the per-view
colour gradients + all-reduce of the 44 B/Gaussian geometry block, gaustudio_amd/parallel.py).
value = N * V * H * W / step_time.

    python bench.py                                  # N=1, C3: static camera (headline) + rotating-camera block + fast_exp block
    python bench.py --workload C3-extract            # BASELINE config 3 as worded: depth+normal render (gs-extract-mesh path)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W [--views-per-rank V]

Prints ONE JSON line on rank 0.  `roofline` is for the dominant north-star kernel (composite_fwd):
algorithmic bytes R*44 + T*8 + H*W*32 (+ H*W*8 training aux) over its mean duration,
measured with HIP events on the launch stream during the timed steps.  `cpu_baseline` is the CPU
oracle (a C restatement of the reference kernels -- the reference itself has no CPU renderer) timed on
the host cores on the same full frame.  `rotating`: the same step with a different camera every step (ring of K) and
an Adam update of all parameters between the steps (excluded from the time, but its 2.8 GB of traffic evicts the
caches as a training loop's would); `variants`: the opt-in fast_exp mode.
"""
import argparse
import json
import math
import os
import subprocess
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
    # a NON-UNIFORM scene: C2's size, the structure of a trained capture -- clustered Gaussians in a ball seen by
    # an inward ring camera, > 20 % empty tiles, a 1 % tail of 50-60 px splats; per-tile lists p50 ~ 10, p99 ~ 11 k, max ~ 58 k
    "C2-clustered": (300_000, 800, 800, 3, "C2-clustered: 300k Gaussians in 48 clusters inside a ball (Zipf populations, 1 % of the splats with sigma "
                                           "50-60 px), inward ring camera at 11 units, 800x800, SH degree 3, fwd+bwd"),
    # C4 with the VISIBILITY of a real 360-degree capture (BASELINE config 4 names MipNeRF-360 garden: the cameras stand
    # INSIDE the scene): 5 M Gaussians in a ball of radius 6, a ring of cameras at radius 2.5 looking through the centre -- a view
    # sees 0.14-0.19 of the Gaussians (the rest is behind the camera or outside the frustum), the union of 8 views 0.5-0.6
    "C4-inside": (5_000_000, 1297, 840, 3, "C4-inside: 5M Gaussians in a ball of radius 6 (garden stand-in), one 1297x840 view from a camera "
                                           "INSIDE the scene (ring of radius 2.5 looking through the centre), SH degree 3, fwd+bwd"),
}
CLUSTERED = {"C2-clustered": dict(cam_distance=11.0, ring=5, view=1)}
INSIDE = {"C4-inside": dict(ball_radius=6.0, cam_radius=2.5, ring=8, sigma=0.008)}


def rank_camera(scenes, W, H, rank, world):
    """Rank r looks down +z from the origin, yawed by 3 degrees per rank around the scene's axis so every
    rank renders a different view of the same (replicated) Gaussians with the same amount of work."""
    a = math.radians(3.0) * (rank - (world - 1) / 2.0)
    R = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    return scenes.make_camera(W, H, R=R)


def cpu_baseline(sc, cam, D, grads, budget_s=30.0):
    """Times the CPU oracle (fwd+bwd) on the same FULL frame (every tile; per-Gaussian stages in full).  Only if a pilot
    on every 32nd tile predicts more than budget_s seconds is the frame sub-sampled (and the line says so)."""
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
    what = "the same full frame" if step == 1 else f"every {step}th 16x16 tile of the same frame ({tiles}/{T} tiles), per-Gaussian stages in full"
    return {"value": round(pixels / t / 1e6, 4), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "seconds": round(t, 2), "full_frame": step == 1,
            "sample": f"CPU restatement of the reference kernels (oracle/gsr_oracle.c, OpenMP x{threads}): fwd+bwd on {what}"}


def stage_roofline(P, P_vis, R, T, H, W, D, fwd_ms, bwd_ms, fwd_only, R_staged=None):
    """Achieved algorithmic GB/s of every stage against the 8 TB/s HBM roofline (BASELINE.md s4 byte counts; the
    records are 64 B here instead of the reference's 48 B of SoA fields, counted as written).  R = instances binned."""
    sh = 12 * (D + 1) ** 2
    jac = 36 if D > 0 else 0          # d(rgb)/d(view direction), left by preprocess_fwd for the SH backward (round 4)
    HW = H * W
    stages = {}
    if fwd_ms:
        Rs = R if R_staged is None else R_staged       # composite_fwd: the instances it actually staged (gsr_inspect_staged)
        # preprocess: the geometry inputs of every Gaussian, the SH row of the VISIBLE ones only (requested late since round 5)
        b = {"preprocess": P * (12 + 12 + 16 + 4) + P_vis * sh + P_vis * (64 + 4 + 4 + (0 if fwd_only else jac)),
             "scatter": R * 8 + P_vis * 64, "sort": R * 12,
             "composite": Rs * 44 + T * 8 + HW * 32 + (0 if fwd_only else HW * 8)}
        for k, nbytes in b.items():
            if fwd_ms.get(k):
                gbs = nbytes / (fwd_ms[k] * 1e-3) / 1e9
                stages[k + "_fwd" if k in ("preprocess", "composite") else k] = {
                    "ms": round(fwd_ms[k], 4), "algorithmic_bytes": nbytes, "GB/s": round(gbs, 1), "frac_hbm": round(gbs / 8000.0, 4)}
    if bwd_ms:
        survey = {"composite_bwd": R * 44 + HW * 32 + R * 40}      # one reduced update of 10 floats per instance (the design's rows are 48 B)
        b = {"composite_bwd": R * 44 + HW * 32 + R * 48,
             # rows, then per visible Gaussian mean / radius / jacobian (instead of the 192-B coefficient row) / record / scale /
             # rotation / row offsets, and the outputs (dL_dcov3D only exists when covariances were supplied: not here)
             "preprocess_bwd": R * 48 + P_vis * (12 + 4 + jac + 64 + 12 + 16 + 8) + P * (12 + 4 + 12 + 12 + sh + 12 + 16)}
        for k, nbytes in b.items():
            if bwd_ms.get(k):
                gbs = nbytes / (bwd_ms[k] * 1e-3) / 1e9
                stages[k] = {"ms": round(bwd_ms[k], 4), "algorithmic_bytes": nbytes, "GB/s": round(gbs, 1),
                             "frac_hbm": round(gbs / 8000.0, 4)}
                if k in survey:
                    stages[k]["survey_8d_bytes"] = survey[k]
                    stages[k]["frac_hbm_survey_8d_bytes"] = round(survey[k] / (bwd_ms[k] * 1e-3) / 1e9 / 8000.0, 4)
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
            "empty_tiles": int((n == 0).sum()), "max_over_mean": round(float(n.max() / n.mean().clamp(min=1e-9)), 2)}


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


def ring_of_cameras(scenes, W, H, K, rank=0, world=1):
    """K cameras for the rotating mode: yawed by 3 degrees per step around the scene's axis (the same kind of view as
    rank_camera's, so every step has about the same amount of work), offset per rank."""
    out = []
    for k in range(K):
        a = math.radians(3.0) * (k - (K - 1) / 2.0 + 0.37 * (rank - (world - 1) / 2.0))
        R = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
        out.append(scenes.make_camera(W, H, R=R))
    return out


def run_extract(a, dev, scenes, _C):
    """BASELINE config 3 as BASELINE.json words it: "1M synthetic Gaussians, 1920x1080, depth+normal render (gs-extract-mesh
    path)".  One step = what gs-extract-mesh does per view (gaustudio/scripts/extract_mesh.py:95-115 with
    gaustudio/datasets/__init__.py:307-380): no_grad forward -> opacity mask on the median depth -> depth2point (world)
    + depth2normal -> TSDF integrate, every stage on the GPU; K ring cameras are cycled (each view of a real extraction is
    integrated once; the first pass over the ring allocates the volume's blocks during warm-up)."""
    from gaustudio_amd import postprocess as pp
    from gaustudio_amd.tsdf import TSDFVolume
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    P, W, H, D, desc = WORKLOADS["C3"]
    cam0 = scenes.make_camera(W, H)
    sc = scenes.make_scene(P, cam0, seed=0)
    K = max(1, a.rotate_cameras or 8)
    cams = ring_of_cameras(scenes, W, H, K)
    params = {k: getattr(sc, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    m2 = torch.zeros_like(params["means3D"])
    rss, Ks, Es = [], [], []
    for c in cams:
        rss.append(GaussianRasterizationSettings(H, W, c.tanfovx, c.tanfovy, torch.zeros(3, device=dev), 1.0, c.viewmatrix.to(dev),
                                                 c.projmatrix.to(dev), D, c.campos.to(dev), False, False))
        f = W / (2 * c.tanfovx)
        Ks.append(torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]))
        Es.append(c.viewmatrix.t().contiguous())
    # the synthetic cloud has no surfaces: its median depth is rough, so a frame touches many more 8^3 blocks than a real
    # scene's would -- voxels of 4 cm (scene depth 2 .. 20) and 2^22 hash slots (16 GiB of the 288) hold it
    vol = TSDFVolume(voxel_size=0.04, sdf_trunc=0.16, space_carving=False, device=dev, capacity_blocks=1 << 22)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    stage_names = ("render", "mask", "points", "normals", "integrate")   # fused epilogue: mask = points = 0, everything under "normals"
    acc = {k: 0.0 for k in stage_names}
    marks = []

    def step(i, timed=False):
        rs, Kc, E, c = rss[i % K], Ks[i % K], Es[i % K], cams[i % K]
        e = [ev() for _ in range(6)] if timed else None
        if timed: e[0].record()
        with torch.no_grad():
            _, _, _, median, opacity = GaussianRasterizer(rs)(means3D=params["means3D"], means2D=m2, opacities=params["opacities"],
                                                               shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
        if timed: e[1].record()
        if a.epilogue_separate:
            invalid = opacity[0] < 0.5                                           # extract_mesh.py:104
            depth = median[0].masked_fill(invalid, 0.0)                          # :107
            if timed: e[2].record()
            pts = pp.depth_to_points(depth, Kc, E, "world")                     # :109 Camera.depth2point(..., 'world')
            if timed: e[3].record()
            nrm = pp.depth_to_normals(depth, Kc, E, coordinate="world")         # Camera.depth2normal (gs-extract-pcd / normal maps)
            if timed: e[4].record()
        else:
            # the three steps in one pass over the frame (gsr_depth_epilogue: mask applied while the tile is read, points and
            # normals written together; value for value the same -- tests/test_postprocess.py); the stage split reports it all
            # under "points + normals"
            if timed: e[2] = e[3] = e[1]                      # (no extra event records: each costs ~5 us of stream time)
            pts, nrm = pp.depth_epilogue(median[0], Kc, E, opacity=opacity[0], min_opacity=0.5, coordinate="world")
            invalid = None
            if timed: e[4].record()
        # :110 compacts `pts[~invalid]` for the CPU library; here a masked pixel's point IS the sensor origin (depth 0) and
        # the integrate kernel skips zero-length rays, so the whole [H*W,3] map goes in: no nonzero / gather pass, no host
        # synchronisation for the count (tests/test_tsdf.py: identical volume)
        # (the [H,W,3] map goes in as a map: the kernel then works in 32 x 32 pixel patches; `--tsdf-linear` = as a flat list)
        vol.integrate(pts.view(-1, 3) if a.tsdf_linear else pts, c.campos)     # :115 vdb_volume.integrate
        if timed:
            e[5].record()
            marks.append(e)
        return nrm, invalid

    for i in range(max(0, a.settle) + max(a.warmup, K)):                        # --settle (sustained clock), then the warm-up: at least one
        step(i)                                                                 # full ring, which also allocates the blocks
    torch.cuda.synchronize()
    # the timed frames carry the two events around composite_fwd only; the per-stage events (six torch events + the library's
    # six per frame: ~7 % of a 0.55-ms frame) go into 12 further, untimed, frames behind the timed region
    _C.set_profiling(2)
    t0 = time.perf_counter()
    inv = None
    for i in range(a.steps):
        _, inv = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fwd_timed = _C.last_forward_ms()
    _C.set_profiling(1)
    for i in range(12):
        step(a.steps + i, timed=True)
    torch.cuda.synchronize()
    if inv is None:                                          # fused epilogue: recompute the mask of the last frame outside the clock
        with torch.no_grad():
            rs = rss[(a.steps - 1) % K]
            inv = GaussianRasterizer(rs)(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                                         scales=params["scales"], rotations=params["rotations"])[4][0] < 0.5
    n_valid = int((~inv).sum().item()) * a.steps             # (after the clock: the first reduction loads a torch code object)
    fwd_ms = _C.last_forward_ms()
    _C.set_profiling(False)
    if fwd_ms and fwd_timed and fwd_timed.get("composite"):
        fwd_ms["composite_in_stage_pass"] = fwd_ms["composite"]
        fwd_ms["composite"], fwd_ms["calls"], fwd_ms["stage_pass_steps"] = fwd_timed["composite"], fwd_timed["calls"], 12
    for e in marks:
        for k, (x, y) in zip(stage_names, zip(e[:-1], e[1:])):
            acc[k] += x.elapsed_time(y)
    stage_ms = {k: round(v / len(marks), 4) for k, v in acc.items()}
    status = int(vol.status.item())
    blocks = int((vol.keys != -1).sum().item())
    HW = H * W
    if a.epilogue_separate:
        epi = {"points": {"ms": stage_ms["points"], "algorithmic_bytes": HW * 16},
               "normals": {"ms": stage_ms["normals"], "algorithmic_bytes": HW * 16}}
    else:   # one pass: depth + opacity read, points + normals written
        epi = {"mask+points+normals (fused)": {"ms": stage_ms["normals"], "algorithmic_bytes": HW * 32}}
    for v in epi.values():
        v["GB/s"] = round(v["algorithmic_bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
        v["frac_hbm"] = round(v["GB/s"] / 8000.0, 4)
    # CPU baseline: the reference's own epilogue is torch on the host tensors it is given (datasets/__init__.py:307-380);
    # timed here as its numpy restatement (oracle/post_oracle.py, pinned to the reference's outputs) on one frame
    cpu = None
    if not a.no_cpu_baseline:
        from oracle import post_oracle as po
        with torch.no_grad():
            _, _, _, median, opacity = GaussianRasterizer(rss[0])(means3D=params["means3D"], means2D=m2, opacities=params["opacities"],
                                                                   shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
        d = median[0].masked_fill(opacity[0] < 0.5, 0.0).cpu().numpy()
        Kn, En = Ks[0].numpy(), Es[0].numpy()
        t1 = time.perf_counter()
        po.depth2point(d, Kn, En)
        po.depth2normal(d, Kn, En)
        tc = time.perf_counter() - t1
        cpu = {"value": round(HW / tc / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port", "seconds": round(tc, 2),
               "sample": "numpy restatement (oracle/post_oracle.py, pinned to outputs of the reference's Camera class) of depth2point + "
                         "depth2normal on one 1920x1080 frame; the render and the TSDF fusion have no CPU path in the reference "
                         "(GPU rasterizer; vdbfusion is an un-vendored C++ dependency)"}
    comp_ms = (fwd_ms or {}).get("composite")
    T = ((W + 15) // 16) * ((H + 15) // 16)
    line = {"metric": "Mpixels/s depth+normal render + TSDF integrate @1M Gaussians 1920x1080 (gs-extract-mesh path)",
            "value": round(HW * a.steps / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": 1, "steps": a.steps, "warmup": max(a.warmup, K),
            "settle_steps": max(0, a.settle),
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3-extract: 1M synthetic Gaussians, 1920x1080, SH degree 3, no_grad render -> median-depth mask -> "
                                   "depth2point + depth2normal -> TSDF integrate (BASELINE config 3, gs-extract-mesh path)",
                       "gaussians": P, "width": W, "height": H, "sh_degree": D, "cameras_cycled": K,
                       "valid_points_per_frame": n_valid // max(1, a.steps), "tsdf": {"voxel_size": 0.04, "sdf_trunc": 0.16,
                       "hash_slots": vol.capacity, "occupied_blocks": blocks, "status": status}},
            "stage_ms": {"pipeline": stage_ms, "forward": fwd_ms},
            "roofline": None if not comp_ms else {
                "kernel": "composite_fwd", "bound": "hbm", "peak": 8000.0, "unit": "GB/s", "avg_ms": round(comp_ms, 4),
                "note": "instances per frame vary with the camera: see the C3 line for the byte model; the epilogue kernels "
                        "(HBM-bound streams) are in epilogue_roofline", "achieved": None, "frac": None, "traffic": None},
            "epilogue_roofline": epi, "cpu_baseline": cpu}
    print(json.dumps(line), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-runs this command line as N ranks under torch.distributed.run
    (one node, rendezvous on 127.0.0.1, a free port) and exits with its return code.  Under torchrun (WORLD_SIZE set)
    bench.py never comes here."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = str(so.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.call(cmd, env=env))


def other_workloads(names, steps=10, warmup=3, settle=30, exact=False, fast=False):
    """{name: {ms_per_step, value, unit, steps, stage_ms, roofline_frac}} of `python bench.py --workload <name> --no-extras ...` run as
    child processes (one at a time, after the headline's timed region; the parent's GPU work is finished and synchronised)."""
    out = {}
    for name in names:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", str(warmup),
               "--settle", str(settle), "--no-extras", "--no-cpu-baseline", "--no-ref-ab"] + (["--exact"] if exact else []) + (["--fast-exp"] if fast else [])
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not rows:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(rows[-1])
            out[name] = {"ms_per_step": j["ms_per_step"], "value": j["value"], "unit": j["unit"], "steps": j["steps"], "warmup": j["warmup"],
                         "settle_steps": j.get("settle_steps"), "stage_ms": j.get("stage_ms"),
                         "composite_fwd_frac_of_hbm": (j.get("roofline") or {}).get("frac"),
                         "instances_binned": j["config"].get("instances_binned"), "visible": j["config"].get("visible"),
                         "workload": j["config"]["workload"], "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:  # noqa: BLE001  (never take the headline down)
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=100,
                    help="untimed steps of the same workload run ONCE, in front of the --warmup steps, to bring the device to its "
                         "sustained clock (after the idle seconds of scene generation the first ~30 steps of the two VALU-bound "
                         "compositing kernels run 5-6 %% slower: --steps 20 --warmup 5 measured 1.076 ms, the same 20 steps after "
                         "50 or 200 warm-up steps 1.030-1.033 ms; profiles/r04_warmup_ramp.txt); 0 = off")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS) + ["C3-extract"])
    ap.add_argument("--fwd-only", action="store_true", help="time the no_grad forward only (inference paths)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-ab", action="store_true", help="skip timing the hipified reference kernels (optional A/B)")
    ap.add_argument("--overlap-chunks", type=int, default=4,
                    help="N > 1, --exchange dense: SH-gradient ranges reduced from inside the backward (0/1 = one all-reduce after it)")
    ap.add_argument("--exchange", default="factored", choices=["dense", "factored"],
                    help="N > 1: dense = ONE all-reduce of the flat 236 B/Gaussian gradient buffer; factored = all-gather of the "
                         "per-view colour gradients (12 B/Gaussian/view) + all-reduce of the 44 B/Gaussian geometry block, the SH "
                         "gradient rebuilt locally (gaustudio_amd/parallel.py)")
    ap.add_argument("--compact", default="none", choices=["none", "view", "view+geometry", "union"],
                    help="N > 1, --exchange factored: none = dense 12 B/Gaussian colour slots; view = per-view packed messages (header + "
                         "12 B per VISIBLE Gaussian, counts agreed off the critical path); view+geometry = also the geometry all-reduce "
                         "on the union of the step's views; union = the round-3 form (mask all-reduce + compacted buffers)")
    ap.add_argument("--bands", type=int, default=1, choices=[1, 2],
                    help="N > 1, --exchange factored --compact view|view+geometry: 2 = every backward runs BANDED (cut at the middle tile row): the "
                         "colour rows of the Gaussians that end above the cut leave while the lower half is still being composited "
                         "(FactoredGradExchange(bands=2); bit-identical gradients)")
    ap.add_argument("--epilogue-separate", action="store_true", help="C3-extract: mask, depth_to_points and depth_to_normals as separate steps (torch ops + two kernels) instead of the fused gsr_depth_epilogue")
    ap.add_argument("--tsdf-linear", action="store_true", help="C3-extract: integrate the point map as a flat list (256 consecutive points per workgroup) instead of 32x32 patches")
    ap.add_argument("--views-per-rank", type=int, default=1, help="cameras rendered (and accumulated) per rank and step")
    ap.add_argument("--rotate-cameras", type=int, default=None,
                    help="K: cycle a ring of K cameras (a different one every step) and apply an Adam update of every parameter "
                         "between the steps; default: the headline is the static camera and a K=8 block is reported beside it")
    ap.add_argument("--fast-exp", action="store_true", help="run the whole benchmark in the fast_exp mode (the library default since round 4; explicit here)")
    ap.add_argument("--exact", action="store_true", help="run the whole benchmark in the bit-exact mode (reproducible polynomial exp: the CPU oracle's bits; "
                                                         "the test suite's mode) instead of the library default")
    ap.add_argument("--loss", default="all", choices=["all", "color"],
                    help="all (the headline, BASELINE's C3: colour + depth + median + opacity gradients consumed) or color: a loss on the "
                         "colour image alone, the usual 3DGS training call -- the three unused outputs' gradients are ABSENT (NULL through "
                         "the C ABI: no zero planes, no loads, the colour-only compositing backward); reported beside the headline as "
                         "`variants.loss_color`")
    ap.add_argument("--no-extras", action="store_true", help="skip the rotating-camera and fast_exp blocks of the default line")
    ap.add_argument("--traffic", type=float, default=None,
                    help="measured HBM bytes per composite_fwd launch from a rocprofv3 --pmc pass; default: the "
                         "committed measurement in profiles/r*_traffic.json for this workload, else null")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N` (the way the driver starts N = 1): become the launcher -- one process
        # per GPU under torch.distributed.run on this node, the same argv; rank 0 of the children prints the JSON line
        return self_launch(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus} "
                         f"(or unset WORLD_SIZE: `python bench.py --gpus {a.gpus}` launches its own ranks)")
    # GSR_BENCH_BACKEND=gloo + fewer GPUs than ranks is a plumbing test of the N > 1 path on a 1-GPU box
    # (ranks share device 0, the collective goes through gloo); the measured configuration is nccl = RCCL.
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    # GSR_BENCH_FORCE_PG=1: run the N > 1 code path (process group, armed buckets, chunk hooks, all-gather / all-reduce,
    # comm block) even at world_size 1 -- every collective call of the multi-GPU step then meets RCCL on a single GPU
    # (tests/test_distributed.py::test_nccl_world1_*); a sum over one rank is the identity, so the numbers are the N = 1 ones
    force_pg = os.environ.get("GSR_BENCH_FORCE_PG") == "1"
    multi = world > 1 or force_pg
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import gaustudio_amd
    from gaustudio_amd import _C, parallel, runtime, scenes
    from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    # one-time process initialisation (not a benchmark step): load the gfx950 code objects with a 512-Gaussian
    # call and reserve an allocator pool, as a long-running trainer / server would at start-up
    runtime.warm_start(dev)
    parallel.FORCE_COLLECTIVES = force_pg
    if a.workload == "C3-extract":
        if world != 1:
            raise SystemExit("C3-extract is a single-GPU workload")
        return run_extract(a, dev, scenes, _C)

    P, W, H, D, desc = WORKLOADS[a.workload]
    V = max(1, a.views_per_rank)
    if a.workload in CLUSTERED:
        c = CLUSTERED[a.workload]
        sc = scenes.make_clustered_scene(P, W, cam_distance=c["cam_distance"], seed=0)
        ring = scenes.ring_cameras(max(c["ring"], world * V + c["view"]), W, H, radius=c["cam_distance"])
        all_cams = [ring[(c["view"] + g) % len(ring)] for g in range(world * V)]
    elif a.workload in INSIDE:
        c = INSIDE[a.workload]
        sc = scenes.make_ball_scene(P, radius=c["ball_radius"], seed=0, sigma=c["sigma"])
        ring = scenes.ring_cameras(max(c["ring"], world * V), W, H, radius=c["cam_radius"])
        all_cams = [ring[g % len(ring)] for g in range(world * V)]
    else:
        cam0 = scenes.make_camera(W, H)
        sc = scenes.make_scene(P, cam0, seed=0)                 # identical on every rank (replicated parameters)
        # this rank's V cameras of the world * V views of a step (view v of rank r = global view r * V + v)
        all_cams = [rank_camera(scenes, W, H, g, world * V) for g in range(world * V)]
    cams = all_cams[rank * V:(rank + 1) * V]
    cam = cams[0]
    grads_cpu = scenes.make_output_grads(cam, seed=1)

    params = {k: getattr(sc, k).to(dev).requires_grad_(not a.fwd_only)
              for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=not a.fwd_only)
    grads = [g.to(dev) for g in grads_cpu]

    def settings(c):
        return GaussianRasterizationSettings(H, W, c.tanfovx, c.tanfovy, torch.zeros(3, device=dev), 1.0,
                                             c.viewmatrix.to(dev), c.projmatrix.to(dev), D, c.campos.to(dev), False, False)
    rs_list = [settings(c) for c in cams]
    rs = rs_list[0]
    rasterizers = [GaussianRasterizer(r) for r in rs_list]
    factored = multi and a.exchange == "factored" and not a.fwd_only   # (step() reads it at call time)
    bucket = None if a.fwd_only else parallel.FlatGradBucket(list(params.values()), roles=params)
    fx = None
    if factored:
        fx = parallel.FactoredGradExchange(params, views_per_rank=V, compact={"none": False, "union": True}.get(a.compact, a.compact),
                                           bands=a.bands, band_split=((H + 15) // 16) // 2 if a.bands == 2 else None)
        campos_all = torch.stack([c.campos for c in all_cams]).to(dev)
    state = {}
    comm_ev = []       # (backward enqueued, exchange finished) events of the timed steps, N > 1 only
    # compositing mode: the library default (fast_exp = v_exp_f32 unless GSR_FAST_EXP=0) or the one asked for
    if a.fast_exp and a.exact:
        raise SystemExit("--fast-exp and --exact exclude each other")
    if a.bands == 2 and not (a.exchange == "factored" and a.compact in ("view", "view+geometry")):
        raise SystemExit("--bands 2 needs --exchange factored --compact view|view+geometry")
    fast_mode = True if a.fast_exp else (False if a.exact else bool(_C.get_option("fast_exp")))
    mode = gaustudio_amd.options(fast_exp=fast_mode)

    def render(i, rasterizer):
        out = rasterizer(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"], shs=params["shs"],
                         scales=params["scales"], rotations=params["rotations"])
        return out

    def step(timed=False, rasts=None):
        state["steps_run"] = state.get("steps_run", 0) + 1
        mode_key = "steps_factored" if factored else "steps_dense"
        state[mode_key] = state.get(mode_key, 0) + 1
        rasts = rasterizers if rasts is None else rasts
        if a.fwd_only:
            with torch.no_grad():
                for r in rasts:
                    state["out"] = render(0, r)
            return
        for p in params.values():
            p.grad = None
        means2D.grad = None
        for v, r in enumerate(rasts):
            last = v == len(rasts) - 1
            if factored:
                fx.arm(v, sh_degree=D)                      # colour gradients of view v go to their all-gather slot; the gather starts inside the backward
            elif multi and len(rasts) == 1:
                # gradients are born in the flat all-reduce buffer; the SH ranges are reduced while the backward runs
                bucket.arm(a.overlap_chunks)
            color, radii, depth, median, opac = render(v, r)
            if factored:
                fx.visible(v, radii)                        # --compact view: header + count gather right after the forward (no-op otherwise)
            if state.get("loss", a.loss) == "color":
                torch.autograd.backward([color], grads[:1])    # the other three outputs unused: their gradients ABSENT (NULL), not zeros
            else:
                torch.autograd.backward([color, depth, median, opac], grads)
            if last:
                state["out"] = (color, radii, depth, median, opac)
        if multi and timed:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        if factored:
            fx.exchange(campos_all, sh_degree=D)
        else:
            parallel.allreduce_gaussian_grads(bucket)      # the tail of the one logical reduction (no-op at N=1)
        if multi and timed:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            comm_ev.append((e0, e1))

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    settled = {"done": False}
    STAGE_STEPS = 12

    def timed_run(fn, steps, warmup, timed_level=2):
        if not settled["done"]:                              # once per process: the device at its sustained clock (see --settle)
            settled["done"] = True
            for _ in range(max(0, a.settle)):
                fn(False)
        for _ in range(warmup):
            fn(False)
        barrier()
        # The K timed steps carry HIP events around composite_fwd only (profiling level 2: two records per step on the launch
        # stream, no host synchronisation): that is the live duration `roofline` is computed from.  Events at every stage
        # boundary (nine per step) cost 3.1 % of a C3 step (1.014 against 0.983 ms, profiles/r04_warmup_ramp.txt), so the stage
        # breakdown `stage_ms` comes from STAGE_STEPS further, untimed, steps behind the timed region; its `composite` entry is
        # the timed steps' own.
        _C.set_profiling(timed_level)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(True)
        barrier()
        dt = time.perf_counter() - t0
        f_timed = _C.last_forward_ms()
        _C.set_profiling(1)
        for _ in range(STAGE_STEPS):
            fn(False)
        barrier()
        f, b = _C.last_forward_ms(), _C.last_backward_ms()
        _C.set_profiling(False)
        if f and f_timed and f_timed.get("composite"):
            f["composite_in_stage_pass"] = f["composite"]
            f["composite"] = f_timed["composite"]
            f["calls"] = f_timed["calls"]
            f["stage_pass_steps"] = STAGE_STEPS
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if multi:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), f, b

    rotating_headline = a.rotate_cameras is not None and a.rotate_cameras > 0

    # N > 1, factored exchange: one untimed step first.  If anything in it raises on this backend (it has only ever run over
    # gloo and on two ranks sharing one GPU: no multi-GPU node was available to the builder), every rank falls back to the
    # dense all-reduce together instead of losing the scaling run; the line says so (`comm.exchange_fallback`).
    if factored:
        ok = 1
        try:
            with mode:
                step(False)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            ok = 0
            state["fallback_reason"] = f"{type(e).__name__}: {e}"[:300]
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            factored = False
            _C.set_grad_arena([])
            state.setdefault("fallback_reason", "another rank failed")

    # ---- rotating mode: K cameras cycled, Adam update of every parameter between the steps ----
    def make_rotating(K):
        ring = ring_of_cameras(scenes, W, H, K, rank, world)
        rrs = [GaussianRasterizer(settings(c)) for c in ring]
        opt = None if a.fwd_only else torch.optim.Adam(list(params.values()), lr=1e-7, foreach=True)
        opt_ev = []
        it = {"i": 0}

        def fn(timed):
            i = it["i"]; it["i"] += 1
            step(timed, [rrs[(i * V + v) % K] for v in range(V)])
            if opt is not None:
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record()
                opt.step()                                   # params + 2 moments + gradients: 16 x 59 floats per Gaussian of traffic
                if timed:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record()
                    opt_ev.append((e0, e1))
        return fn, opt_ev

    with mode:
        if rotating_headline:
            rot_fn, opt_ev = make_rotating(a.rotate_cameras)
            dt, fwd_ms, bwd_ms = timed_run(rot_fn, a.steps, a.warmup)
            torch.cuda.synchronize()
            opt_ms = sum(x.elapsed_time(y) for x, y in opt_ev) / max(1, len(opt_ev)) if opt_ev else 0.0
            dt -= opt_ms * 1e-3 * a.steps                    # the optimizer is not part of the metric; its traffic is the point
        else:
            dt, fwd_ms, bwd_ms = timed_run(lambda timed: step(timed), a.steps, a.warmup)
            opt_ms = None

    extras = {}
    if multi and not a.fwd_only and not a.no_extras and not rotating_headline and not state.get("fallback_reason"):
        # N > 1: the OTHER exchange beside the headline, a few steps of the same views (north_star words the exchange as "a single
        # RCCL all-reduce": that is `dense`; the default headline is its factored form) -- both on one line.  Every rank takes
        # the same branch: `factored` / the fallback flag were agreed above.
        head_ev = list(comm_ev)
        del comm_ev[:]
        was_factored = factored
        vsteps, vwarm = max(3, min(a.steps, 10)), 3
        other, err = ("dense" if was_factored else "factored"), None
        try:
            _C.set_grad_arena([])
            for p in params.values():
                p.grad = None
            if was_factored:
                factored = False
            else:
                if fx is None:
                    fx = parallel.FactoredGradExchange(params, views_per_rank=V, compact={"none": False, "union": True}.get(a.compact, a.compact),
                                           bands=a.bands, band_split=((H + 15) // 16) // 2 if a.bands == 2 else None)
                    campos_all = torch.stack([c.campos for c in all_cams]).to(dev)
                factored = True
            with mode:
                vdt, vf, vb = timed_run(lambda timed: step(timed), vsteps, vwarm)
            torch.cuda.synchronize()
            vexp = sum(e0.elapsed_time(e1) for e0, e1 in comm_ev) / max(1, len(comm_ev))
        except Exception as e:  # noqa: BLE001  (the variant must never take the headline down with it)
            err = f"{type(e).__name__}: {e}"[:300]
        flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            extras["variants"] = {other: {
                "steps": vsteps, "warmup": vwarm, "ms_per_step": round(vdt / vsteps * 1e3, 4),
                "value": round(world * V * H * W / (vdt / vsteps) / 1e6, 3), "unit": "Mpixels/s",
                "comm_exposed_ms": round(vexp, 4),
                "payload_bytes_per_rank": bucket.nbytes if other == "dense" else fx.payload()["payload_bytes_per_rank"],
                "note": ("--exchange dense: ONE logical RCCL all-reduce of the flat 236 B/Gaussian gradient buffer (north_star's wording), "
                         f"its SH ranges reduced from inside the backward in {a.overlap_chunks if a.overlap_chunks > 1 and V == 1 else 1} chunk(s)")
                        if other == "dense" else
                        "--exchange factored: all-gather of the per-view colour gradients + all-reduce of the 44 B/Gaussian geometry block"}}
        else:
            extras["variants"] = {other: {"error": err or "another rank failed"}}
        _C.set_grad_arena([])
        for p in params.values():
            p.grad = None
        factored = was_factored
        del comm_ev[:]
        comm_ev.extend(head_ev)
    if not multi and not a.no_extras and not rotating_headline and a.workload in ("C3", "C3D0"):
        # beside the static-camera headline: (1) the rotating / optimizer-contended step (same mode), (2) the OTHER compositing mode
        K = 8
        rot_fn, opt_ev = make_rotating(K)
        with mode:
            rdt, rf, rb = timed_run(rot_fn, 24, 8)
        torch.cuda.synchronize()
        ro = sum(x.elapsed_time(y) for x, y in opt_ev) / max(1, len(opt_ev)) if opt_ev else 0.0
        rstep = rdt / 24 * 1e3 - ro
        extras["rotating"] = {"cameras": K, "steps": 24, "ms_per_step": round(rstep, 4), "optimizer_ms_excluded": round(ro, 4),
                              "value": round(V * H * W / (rstep * 1e-3) / 1e6, 3), "unit": "Mpixels/s",
                              "stage_ms": {"forward": rf, "backward": rb},
                              "note": "a different camera every step (ring of 8, 3 degrees apart) and an Adam update of all "
                                      "59 floats per Gaussian between the steps (timed with events and subtracted): nothing of "
                                      "the previous step survives in the Infinity Cache, the binning capacity is speculated "
                                      "from another view's count"}
        if not a.fwd_only:
            for p in params.values():
                p.grad = None
        with gaustudio_amd.options(fast_exp=not fast_mode):
            fdt, ff, fb = timed_run(lambda timed: step(timed), 24, 8)
        other = "bit_exact" if fast_mode else "fast_exp"
        extras["variants"] = {other: {"steps": 24, "ms_per_step": round(fdt / 24 * 1e3, 4),
                                      "value": round(V * H * W / (fdt / 24) / 1e6, 3), "unit": "Mpixels/s",
                                      "stage_ms": {"forward": ff, "backward": fb},
                                      "note": ("gaustudio_amd.options(fast_exp=False) / GSR_FAST_EXP=0: the reproducible 9-instruction exp, every output "
                                               "bit-identical to the CPU oracle (the test suite's mode)") if fast_mode else
                                              ("gaustudio_amd.options(fast_exp=True): v_exp_f32 in both compositing kernels (the library default)")}}
        # rounds 1-3 timed the steps WITH HIP events at every stage boundary (nine records per step); kept on the line so that rounds
        # stay comparable (housekeeping).  The other half of the r03 -> r04 protocol change, the 100 --settle steps,
        # cannot be undone inside a warm process: profiles/r04_warmup_ramp.txt has it (1.076 ms after 5 warm-up steps, 1.03 after 50+).
        with mode:
            ldt, _, _ = timed_run(lambda timed: step(timed), 24, 4, timed_level=1)
        extras["legacy_protocol_ms"] = {"ms_per_step": round(ldt / 24 * 1e3, 4), "steps": 24,
                                        "note": "the headline step timed as rounds 1-3 did: stage-boundary events (profiling level 1, nine records "
                                                "per step) inside the timed region; warm clock (see profiles/r04_warmup_ramp.txt for the --settle part)"}
        if not a.fwd_only and a.loss == "all":
            for p in params.values():
                p.grad = None
            state["loss"] = "color"
            with mode:
                cdt, cf, cb = timed_run(lambda timed: step(timed), 24, 8)
            state.pop("loss")
            extras["variants"]["loss_color"] = {
                "steps": 24, "ms_per_step": round(cdt / 24 * 1e3, 4), "value": round(V * H * W / (cdt / 24) / 1e6, 3), "unit": "Mpixels/s",
                "stage_ms": {"forward": cf, "backward": cb},
                "note": "the same step with a loss on the colour image alone (--loss color): the gradients of depth / median / opacity are "
                        "absent -- NULL through the C ABI, no zero planes materialised (the reference reads all four, backward.cu:476-483, "
                        "and its autograd fills 41 MB of zeros per 1080p step), nothing loaded for them, composite_bwd's colour-only instantiation"}

    if not multi and not a.no_extras and not rotating_headline and a.workload == "C3" and not a.fwd_only and a.loss == "all":
        # the 8-GPU configurations on the driver's clock too.  One view of C4 / C4-inside / C5 each, a handful of steps
        # behind the headline, each in a child process of this very script (its own scene, its own settle steps; `--no-extras` there)
        extras["other_workloads"] = other_workloads(("C4", "C4-inside", "C5"), exact=a.exact, fast=a.fast_exp)

    # instance counts of this rank's view (they drive every composite-stage byte count): the reference-defined
    # num_rendered and the instances actually binned; per-tile list lengths
    with torch.no_grad():
        out = _C.rasterize_gaussians(rs.bg, params["means3D"], torch.Tensor([]), params["opacities"], params["scales"],
                                     params["rotations"], 1.0, torch.Tensor([]), rs.viewmatrix, rs.projmatrix,
                                     rs.tanfovx, rs.tanfovy, H, W, params["shs"], D, rs.campos, False, False)
        R_ref = out[0]
        counts = _C.inspect_counts(out[8], W, H)
        R = counts["num_binned"]
        R_staged = _C.inspect_staged(out[8], W, H)            # what composite_fwd fetched before every pixel of a tile had saturated
        _, ranges = _C.inspect_binning(out[7], out[8], R, W, H)
        lists = list_histogram(ranges)
        vis = int((out[5] > 0).sum().item())
        del out

    if rank == 0:
        T = ((W + 15) // 16) * ((H + 15) // 16)
        ms_per_step = dt / a.steps * 1e3
        value = world * V * H * W / (dt / a.steps) / 1e6
        comp_ms = fwd_ms["composite"] if fwd_ms else None
        # SURVEY s8(d) bytes of composite_fwd on the instances the kernel actually STAGED (it stops fetching a tile's list, 256
        # entries at a time, once every pixel of the tile has saturated: on dense frames -- C4: 3.4 k entries per tile, 99 % of the
        # pixels saturated after a few hundred -- most of the binned instances are never read; dividing ALL of them by the
        # kernel time would print a bandwidth the kernel does not have)
        alg_bytes = R_staged * 44 + T * 8 + H * W * 32 + (0 if a.fwd_only else H * W * 8)
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
                    "traffic_by_mode": (counters or {}).get("by_mode"),
                    "algorithmic_bytes": alg_bytes, "avg_ms": round(comp_ms, 4), "instances_staged": R_staged,
                    "algorithmic_bytes_all_binned_instances": R * 44 + T * 8 + H * W * 32 + (0 if a.fwd_only else H * W * 8),
                    "algorithmic_bytes_with_reference_R": R_ref * 44 + T * 8 + H * W * 32 + (0 if a.fwd_only else H * W * 8),
                    "valu": valu_issue(counters, "composite_fwd", comp_ms),
                    "note": "algorithmic bytes use the instances the kernel actually staged (`instances_staged` of the `instances_binned`: "
                            "tight binning, and a tile stops fetching once all its pixels have saturated); `traffic` = FETCH_SIZE + "
                            "WRITE_SIZE of the committed PMC passes (a PMC pass cannot run inside this process) with the "
                            "access-pattern calibration of profiles/r02_fetch_write_calibration.txt (`traffic_upper` = 2*FETCH + "
                            "WRITE); the kernel is VALU-issue bound (256 pixel evaluations per staged 48-B record): `valu.issue_frac` "
                            "is the fraction of the SIMDs' issue cycles its VALU instructions fill, see DESIGN.md s4"}
        if multi:
            par = (f"{V} camera(s) per GPU x{world}, one logical gradient exchange per step: " +
                   (f"all-gather of {fx.color_bytes_per_rank} B/rank of colour gradients + all-reduce of {fx.geometry_bytes} B of "
                    f"geometry gradients (SH gradient rebuilt locally)" if factored else
                    f"1 RCCL all-reduce of {bucket.nbytes} B/rank ({a.overlap_chunks if a.overlap_chunks > 1 and V == 1 else 1} chunk(s) "
                    f"overlapped with the backward)"))
        else:
            par = "single GPU"
        line = {
            "metric": "Mpixels/s fwd+bwd @1M Gaussians 1920x1080" if a.workload == "C3" and not a.fwd_only
                      else f"Mpixels/s {'fwd' if a.fwd_only else 'fwd+bwd'} @{a.workload}",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "settle_steps": max(0, a.settle),
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "gaussians": P, "visible": vis, "width": W, "height": H, "sh_degree": D,
                       "num_rendered": R_ref, "instances_binned": R, "instances_staged_by_composite_fwd": R_staged, "tiles": T, "tile_list_length": lists,
                       "views_per_step": world * V, "views_per_rank": V,
                       "loss": "colour + depth + median + opacity gradients consumed" if a.loss == "all" else "colour gradient only (--loss color)",
                       "camera": (f"ring of {a.rotate_cameras}, a different one every step, Adam update between the steps "
                                  f"({round(opt_ms, 4)} ms, excluded)") if rotating_headline else "static (the same view every step)",
                       "mode": ("fast_exp (library default: v_exp_f32 in the compositing kernels; pinned to the reference's kernels and to the "
                                "CPU oracle by threshold-event attribution, tests/test_gpu_ref.py + tests/test_gpu_fastexp_oracle.py, "
                                "profiles/r04_parity.json)") if fast_mode else
                               "bit-exact (reproducible exp: every output bit-identical to the CPU oracle; --exact / GSR_FAST_EXP=0)",
                       "parallelism": par},
            "stage_ms": {"forward": fwd_ms, "backward": bwd_ms},
            "stage_roofline": stage_roofline(P, vis, R, T, H, W, D, fwd_ms, bwd_ms, a.fwd_only, R_staged),
            "roofline": roof,
        }
        line.update(extras)
        if bwd_ms:
            line["valu_composite_bwd"] = valu_issue(counters, "composite_bwd", bwd_ms.get("composite_bwd"))
        if multi and comm_ev:
            torch.cuda.synchronize()
            exposed = sum(e0.elapsed_time(e1) for e0, e1 in comm_ev) / len(comm_ev)
            comp_total = (sum((fwd_ms or {}).get(k, 0.0) for k in ("preprocess", "scan", "scatter", "sort", "composite")) +
                          sum((bwd_ms or {}).get(k, 0.0) for k in ("composite_bwd", "preprocess_bwd")))
            # every dense step of the process (settle, warm-up, timed, stage pass; the `variants` block included when it ran dense)
            n_it = max(1, state.get("steps_dense", a.steps + a.warmup))
            comm = {"exchange": "factored" if factored else "dense", "exchange_fallback": state.get("fallback_reason"),
                    "compute_ms": round(comp_total * V, 4), "comm_exposed_ms": round(exposed, 4),
                    "note": "compute_ms = sum of the operator's kernel stages (HIP events) x views per rank; comm_exposed_ms = time "
                            "from the end of the last backward's enqueue to the end of the exchange on rank 0's stream (what the "
                            "collectives add to the step after overlap)"}
            if factored:
                comm.update(fx.payload())
            else:
                comm.update({"payload_bytes_per_rank": bucket.nbytes, "dense_payload_bytes_per_rank": bucket.nbytes,
                             "chunks_per_step": bucket.stats["chunks"] / n_it,
                             "chunk_bytes_per_step": bucket.stats["chunk_bytes"] / n_it,
                             "tail_bytes_per_step": bucket.stats["tail_bytes"] / n_it})
            line["comm"] = comm
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sc, cam, D, grads_cpu)
        else:
            line["cpu_baseline"] = None
        if world == 1 and not a.no_ref_ab and not a.fwd_only:
            line["reference_hipified_ms"] = reference_ab(sc, cam, D, grads_cpu, dev)
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
