#!/usr/bin/env python
"""Extended campaign of tests/test_gpu_fastexp_oracle.py on a GPU box (not part of the test suite; test infrastructure: imports
oracle/): the library's DEFAULT compositing mode (fast_exp, v_exp_f32) against the bit-exact CPU oracle on many more seeded
random configurations -- everything in front of compositing bit-equal, images within 1e-5 except at pixels the float64 replay
attributes to a threshold event, integer state equal except there, backward inside the summation bound.
usage: python tests/tools/fuzz_fastexp.py [--n 150] [--first 100]"""
import argparse, os, sys, time, traceback
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GSR_FAST_EXP", "0")        # process default = reproducible mode; fast_vs_oracle selects fast_exp per call
from gaustudio_amd import scenes
from oracle import pyoracle
from util import scene_kwargs
from test_gpu_fastexp_oracle import fast_vs_oracle

ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=150); ap.add_argument("--first", type=int, default=100)
a = ap.parse_args()
pyoracle.build()
fails, t0, flagged, pixels = [], time.time(), 0, 0
for seed in range(a.first, a.first + a.n):
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(17, 420)), int(rng.integers(9, 300))
    P = int(rng.choice([37, 300, 2500, 9000, 20000, 60000]))
    D = int(rng.integers(0, 4))
    sigma = float(rng.choice([0.7, 1.5, 4.0, 12.0]))
    use_sh = bool(rng.random() < 0.75)
    use_cov = bool(rng.random() < 0.25)
    mod = float(rng.choice([1.0, 1.0, 0.6, 1.9]))
    bg = torch.tensor(rng.random(3), dtype=torch.float32) if rng.random() < 0.5 else None
    cam = scenes.make_camera(W, H, fovx_deg=float(rng.choice([35.0, 60.0, 95.0])))
    sc = scenes.make_scene(P, cam, seed=seed, sigma_px_median=sigma, zmin=float(rng.choice([0.15, 2.0])))
    kw = scene_kwargs(sc, use_sh, use_cov)
    try:
        st = fast_vs_oracle(pyoracle, sc, cam, D if use_sh else 0, kw, scale_modifier=mod, bg=bg, seed=seed)
        f = int((st.get("attribution") or {}).get("flagged", 0)) if isinstance(st, dict) else 0
        flagged += f; pixels += W * H
        print(f"seed {seed}: ok  {W}x{H} P={P} D={D} sigma={sigma} flagged pixels {f}  [{time.time() - t0:.0f} s]", flush=True)
    except Exception as e:  # noqa: BLE001
        fails.append(seed)
        print(f"seed {seed}: FAIL {W}x{H} P={P} D={D}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=2)
print(f"fast_exp campaign: {a.n} configurations from seed {a.first}, {len(fails)} failures {fails}, attributed threshold-event pixels {flagged} of {pixels}, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
