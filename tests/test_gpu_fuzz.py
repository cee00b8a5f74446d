"""-m gpu: seeded random configurations (sizes, image shapes, SH degree, footprints, modifiers, input variants) through
the same two checks as the dedicated tests: forward BIT-EXACT against the oracle, backward sums inside the fp32
summation bound and the per-Gaussian stage bit-exact given the sums."""
import numpy as np
import pytest
import torch

from gaustudio_amd import scenes

from test_gpu_backward import _check
from util import compare_forward_exact, hip_forward, oracle_forward, scene_kwargs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_random_configuration(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.integers(17, 420))
    H = int(rng.integers(9, 300))
    P = int(rng.choice([37, 300, 2500, 9000, 20000]))
    D = int(rng.integers(0, 4))
    sigma = float(rng.choice([0.7, 1.5, 4.0, 12.0]))
    use_sh = bool(rng.random() < 0.75)
    use_cov = bool(rng.random() < 0.25)
    mod = float(rng.choice([1.0, 1.0, 0.6, 1.9]))
    bg = torch.tensor(rng.random(3), dtype=torch.float32) if rng.random() < 0.5 else None
    cam = scenes.make_camera(W, H, fovx_deg=float(rng.choice([35.0, 60.0, 95.0])))
    sc = scenes.make_scene(P, cam, seed=seed, sigma_px_median=sigma, zmin=float(rng.choice([0.15, 2.0])))   # 0.15: near-plane culls
    kw = scene_kwargs(sc, use_sh, use_cov)
    Dk = D if use_sh else 0
    os_ = oracle_forward(oracle, sc, cam, Dk, kw, mod, bg)
    hs = hip_forward(sc, cam, Dk, kw, mod, bg)
    compare_forward_exact(hs, os_)
    if os_["num_rendered"] > 0:
        _check(oracle, sc, cam, Dk, kw, scale_modifier=mod, bg=bg, seed=seed)
