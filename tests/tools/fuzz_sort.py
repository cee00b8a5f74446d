#!/usr/bin/env python
"""Fuzz of the per-tile sort regimes on a GPU box (not part of the test suite; test infrastructure: imports oracle/): few
tiles, list lengths from 1 k to 400 k keys, depth distributions from uniform to piled (quantised to a few levels, all equal,
two thin sheets), with and without MANY long lists in the frame -- forward BIT-EXACT against the CPU oracle (point_list,
ranges, n_contrib, images), i.e. every list in (depth, id) order.
usage: python tests/tools/fuzz_sort.py [--n 60] [--first 0]"""
import argparse, os, sys, time, traceback
os.environ.setdefault("GSR_FAST_EXP", "0")   # the reproducible mode (as tests/conftest.py), set before libgsrast.so loads
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaustudio_amd import scenes
from oracle import pyoracle
from util import compare_forward_exact, hip_forward, oracle_forward, scene_kwargs

ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=60); ap.add_argument("--first", type=int, default=0)
a = ap.parse_args()
pyoracle.build()
fails, t0 = [], time.time()
for seed in range(a.first, a.first + a.n):
    rng = np.random.default_rng(9100 + seed)
    many = rng.random() < 0.25
    if many:                                    # > 1024 tiles, most of them with long lists
        W, H = 16 * int(rng.integers(33, 40)), 16 * int(rng.integers(32, 36))
        P = int(rng.choice([200000, 320000]))
        sig = float(rng.choice([4.0, 6.0]))
    else:
        W, H = int(rng.integers(16, 97)), int(rng.integers(16, 65))
        P = int(rng.choice([3000, 9000, 20000, 60000, 150000, 400000]))
        sig = float(rng.choice([3.0, 6.0, 12.0]))
    mode = str(rng.choice(["uniform", "levels", "equal", "sheets", "clump"]))
    cam = scenes.make_camera(W, H)
    sc = scenes.make_scene(P, cam, seed=seed, sigma_px_median=sig)
    m = sc.means3D.clone()
    z = m[:, 2].clone()
    g = torch.Generator().manual_seed(seed)
    if mode == "levels":
        z = torch.round(z * float(rng.choice([0.25, 1.0, 8.0]))) / float(rng.choice([0.25, 1.0, 8.0])) + 0.5
    elif mode == "equal":
        z = torch.full_like(z, 5.0)
    elif mode == "sheets":
        z = torch.where(torch.rand(P, generator=g) < 0.5, 4.0 + 1e-4 * torch.rand(P, generator=g), 9.0 + 1e-6 * torch.rand(P, generator=g))
    z = z.clamp_min(0.3)
    m[:, 0] *= z / m[:, 2]; m[:, 1] *= z / m[:, 2]; m[:, 2] = z
    if mode == "clump":                         # a fraction of the scene on one line of sight
        K = int(P * float(rng.choice([0.05, 0.3])))
        m[:K, 0] = m[:K, 2] * 0.01 * torch.randn(K, generator=g); m[:K, 1] = m[:K, 2] * 0.01 * torch.randn(K, generator=g)
    sc = sc._replace(means3D=m.contiguous())
    kw = scene_kwargs(sc, True, False)
    try:
        os_ = oracle_forward(pyoracle, sc, cam, 0, kw)
        n = os_["ranges"][:, 1].astype(np.int64) - os_["ranges"][:, 0]
        hs = hip_forward(sc, cam, 0, kw)
        compare_forward_exact(hs, os_)
        print(f"seed {seed}: ok  {W}x{H} P={P} {mode:8s} lists max {int(n.max())} long {int((n > 1024).sum())} of {len(n)}  [{time.time() - t0:.0f} s]", flush=True)
    except Exception as e:  # noqa: BLE001
        fails.append(seed)
        print(f"seed {seed}: FAIL {W}x{H} P={P} {mode}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=2)
print("failures:", fails)
sys.exit(1 if fails else 0)
