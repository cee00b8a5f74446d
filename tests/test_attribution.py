"""CPU (-m "not gpu"): the threshold-event attribution machinery of tests/attribution.py, checked against the CPU
oracle before the GPU tests rely on it."""
import numpy as np
import torch

from gaustudio_amd import scenes
from oracle import pyoracle as po

import attribution as at
from util import oracle_forward, scene_kwargs


def _oracle_state(P=4000, W=160, H=96, seed=3):
    cam = scenes.make_camera(W, H)
    sc = scenes.make_scene(P, cam, seed=seed, sigma_px_median=2.5)
    st = oracle_forward(po, sc, cam, 3, scene_kwargs(sc, True, False), tight=True)
    return st, cam


def test_replay_default_leaf_reproduces_the_oracle_pixel():
    """With no decision inside a window there is exactly one leaf, and it is the oracle's pixel (float64 replay vs
    fp32 accumulation: a few 1e-7)."""
    st, cam = _oracle_state()
    gx = (cam.width + 15) // 16
    rng = np.random.default_rng(0)
    n_single = 0
    for _ in range(150):
        x, y = int(rng.integers(0, cam.width)), int(rng.integers(0, cam.height))
        tile = (y // 16) * gx + x // 16
        ids = at.tile_list(st, tile)
        xy, co, rgb, dep = at._records(st, ids)
        leaves = at._explore(len(ids), *at._terms(xy, co, float(x), float(y)), rgb, dep, None)
        want = np.array([st["color"][0, y, x], st["color"][1, y, x], st["color"][2, y, x], st["depth"][0, y, x], st["opacity"][0, y, x]], np.float64)
        best = min(np.abs(v - want).max() for v, _ in leaves)
        assert best <= 4e-6 * max(1.0, np.abs(want).max()), (x, y, best)
        n_single += len(leaves) == 1
    assert n_single >= 120          # windows are narrow: nearly every pixel has a single leaf


def _hand_state(alpha_mid):
    """One tile, three Gaussians centred on pixel (5, 5); the middle one has opacity * G = alpha_mid there."""
    means2D = np.array([[5.0, 5.0]] * 3, np.float32)
    conic_opacity = np.array([[0.5, 0.0, 0.5, 0.6], [0.5, 0.0, 0.5, alpha_mid], [0.5, 0.0, 0.5, 0.7]], np.float32)
    rgb = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], np.float32)
    depths = np.array([1.0, 2.0, 3.0], np.float32)
    return dict(ranges=np.array([[0, 3]], np.uint32), point_list=np.array([0, 1, 2], np.uint32), means2D=means2D,
                conic_opacity=conic_opacity, rgb=rgb, depths=depths)


def _pixel(alphas, rgb, depths):
    T, c, d = 1.0, np.zeros(3), 0.0
    for a, col, dep in zip(alphas, rgb, depths):
        c += col * a * T
        d += dep * a * T
        T *= 1 - a
    return np.array([c[0], c[1], c[2], d, 1 - T])


def test_alpha_threshold_flip_is_attributed_and_a_wrong_value_is_not():
    a_mid = (1.0 / 255.0) * (1.0 + 2e-6)                     # inside WIN_ALPHA: either decision is legitimate
    st = _hand_state(a_mid)
    with_mid = _pixel([0.6, a_mid, 0.7], st["rgb"].astype(np.float64), st["depths"].astype(np.float64))
    without = _pixel([0.6, 0.7], st["rgb"][[0, 2]].astype(np.float64), st["depths"][[0, 2]].astype(np.float64))
    tol = np.full(5, 4e-6)
    res = at.attribute_pixel(st, 0, 5, 5, with_mid, without, tol, tol)
    assert res["attributed"] and res["kinds"] == ["alpha"] and res["events"][0][0] == 1 and res["events"][0][2] < at.WIN_ALPHA
    # a difference that no in-window decision explains (a wrong colour) must NOT be attributed
    wrong = without.copy(); wrong[1] += 3e-3
    assert not at.attribute_pixel(st, 0, 5, 5, with_mid, wrong, tol, tol)["attributed"]
    # the same alpha well outside the window: dropping that Gaussian is a bug, not a flip
    st2 = _hand_state((1.0 / 255.0) * 1.01)
    w2 = _pixel([0.6, (1.0 / 255.0) * 1.01, 0.7], st2["rgb"].astype(np.float64), st2["depths"].astype(np.float64))
    assert not at.attribute_pixel(st2, 0, 5, 5, w2, without, tol, tol)["attributed"]
    # identical values: nothing to attribute (no differing decision)
    assert not at.attribute_pixel(st, 0, 5, 5, with_mid, with_mid, tol, tol)["attributed"]


def test_done_threshold_and_membership_events():
    # T (1 - alpha) lands within WIN_T of 1e-4 at the third Gaussian: "done" or one more contribution
    a0, a1 = 0.99, 0.99
    T2 = (1 - a0) * (1 - a1)                                  # 1e-4
    st = _hand_state(a1)
    st["conic_opacity"][0, 3] = a0
    st["conic_opacity"][2, 3] = 2e-4                          # T2 * (1 - 2e-4) = 0.99980e-4: inside WIN_T below the threshold
    st["conic_opacity"][2, 3] = 0.004                         # alpha >= 1/255 needed to reach the T test
    al = [float(np.float32(a0)), float(np.float32(a1)), float(np.float32(0.004))]
    rgb, dep = st["rgb"].astype(np.float64), st["depths"].astype(np.float64)
    stop = _pixel(al[:2], rgb[:2], dep[:2])
    go = _pixel(al, rgb, dep)
    tol = np.full(5, 4e-6)
    test_T = (1 - al[0]) * (1 - al[1]) * (1 - al[2])
    assert abs(test_T / 1e-4 - 1) < 6e-3
    old = at.WIN_T
    try:
        at.WIN_T = 1e-2
        res = at.attribute_pixel(st, 0, 5, 5, stop, go, tol, tol)
    finally:
        at.WIN_T = old
    assert res["attributed"] and res["kinds"] == ["T"]
    # membership: Gaussian 1 is in the list of one implementation only (its radius differs)
    st3 = _hand_state(0.3)
    a = _pixel([0.6, 0.3, 0.7], rgb, dep)
    b = _pixel([0.6, 0.7], rgb[[0, 2]], dep[[0, 2]])
    assert not at.attribute_pixel(st3, 0, 5, 5, a, b, tol, tol)["attributed"]
    res = at.attribute_pixel(st3, 0, 5, 5, a, b, tol, tol, maybe_ids=[1])
    assert res["attributed"] and res["kinds"] == ["radius"]


def test_attribute_images_on_oracle_with_a_planted_flip():
    """End to end: B = the oracle's images with one pixel replaced by the replay leaf of a flipped in-window decision
    (planted by widening the window), plus one pixel with a plain error -> one attributed, one unattributed."""
    st, cam = _oracle_state()
    W, H = cam.width, cam.height
    imgs_a = dict(color=st["color"], depth=st["depth"], opacity=st["opacity"])
    imgs_b = {k: v.copy() for k, v in imgs_a.items()}
    st_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) and k in ("radii", "means2D") else v) for k, v in st.items()}
    gx = (W + 15) // 16
    old = at.WIN_ALPHA
    planted = None
    try:
        at.WIN_ALPHA = 0.05                                   # wide: find a pixel with a candidate to flip
        for y in range(8, H, 7):
            for x in range(8, W, 5):
                tile = (y // 16) * gx + x // 16
                ids = at.tile_list(st, tile)
                xy, co, rgb, dep = at._records(st, ids)
                leaves = at._explore(len(ids), *at._terms(xy, co, float(x), float(y)), rgb, dep, None)
                alt = [(v, e) for v, e in leaves if len(e) == 1 and e[0][1] == "alpha"]
                base = [v for v, e in leaves if not e]
                if alt and base and np.abs(alt[0][0] - base[0]).max() > 2e-5:
                    planted = (x, y, alt[0][0])
                    break
            if planted:
                break
        assert planted is not None
        x, y, v = planted
        imgs_b["color"][:, y, x] = v[:3]; imgs_b["depth"][0, y, x] = v[3]; imgs_b["opacity"][0, y, x] = v[4]
        imgs_b["color"][1, 3, 3] += 1e-3                      # a plain error
        rep = at.attribute_images(st_t, W, H, imgs_a, imgs_b, tol=1e-5, depth_scale=20.0)
    finally:
        at.WIN_ALPHA = old
    assert rep["flagged"] == 2 and rep["attributed"] == 1 and len(rep["unattributed"]) == 1
    assert rep["unattributed"][0]["pixel"] == [3, 3] and rep["by_kind"] == {"alpha": 1}


def test_replay_default_leaf_reproduces_the_oracle_median_channels():
    """Round 6: with the tile's ids handed over the leaf also carries (median depth, median weight, median id) -- forward.cu:366-374 --
    and the default leaf is the oracle's pixel in all eight channels; pixels that never cross T = 0.5 keep (15, 0, 0)."""
    dense, cam = _oracle_state()
    sparse, _ = _oracle_state(P=250)
    gx = (cam.width + 15) // 16
    rng = np.random.default_rng(1)
    crossed = 0
    for k in range(200):
        st = dense if k % 2 else sparse
        x, y = int(rng.integers(0, cam.width)), int(rng.integers(0, cam.height))
        tile = (y // 16) * gx + x // 16
        ids = at.tile_list(st, tile)
        xy, co, rgb, dep = at._records(st, ids)
        leaves = at._explore(len(ids), *at._terms(xy, co, float(x), float(y)), rgb, dep, None, ids)
        want = np.array([st["color"][0, y, x], st["color"][1, y, x], st["color"][2, y, x], st["depth"][0, y, x], st["opacity"][0, y, x],
                         st["median"][0, y, x], st["median"][1, y, x], st["median"][2, y, x]], np.float64)
        default = [v for v, e in leaves if not e]
        assert len(default) == 1
        d = np.abs(default[0] - want)
        assert d[:7].max() <= 4e-6 * max(1.0, np.abs(want[:7]).max()) and d[7] == 0.0, (x, y, default[0], want)
        crossed += want[6] > 0
        if want[6] == 0:
            assert want[5] == 15.0 and want[7] == 0.0
    assert 20 <= crossed <= 195          # the scene exercises both outcomes


def _pixel8(alphas, rgb, depths, ids, fire_at):
    v = _pixel(alphas, rgb, depths)
    T, med = 1.0, [15.0, 0.0, 0.0]
    for k, (a, dep, i) in enumerate(zip(alphas, depths, ids)):
        if k == fire_at:
            med = [dep, a * T, float(i)]
        T *= 1 - a
    return np.concatenate([v, med])


def test_median_crossing_inside_its_window_is_attributed_either_way():
    """T (1 - alpha) within WIN_MEDIAN of 0.5 at the first Gaussian: one implementation records the first Gaussian as the median,
    the other the second (whose T is that product).  Both are leaves; a median id with no such event is not attributed."""
    tol = np.concatenate([np.full(7, 4e-6), [0.5]])
    for a0 in (0.5 * (1 + 1.2e-7), 0.5 * (1 - 1.2e-7)):            # test_T just below / just above 0.5
        st = _hand_state(0.3)
        st["conic_opacity"][0, 3] = a0
        al = [float(st["conic_opacity"][k, 3]) for k in range(3)]
        assert abs((1 - al[0]) / 0.5 - 1) < at.WIN_MEDIAN
        rgb, dep = st["rgb"].astype(np.float64), st["depths"].astype(np.float64)
        first = _pixel8(al, rgb, dep, [0, 1, 2], 0)
        second = _pixel8(al, rgb, dep, [0, 1, 2], 1)
        third = _pixel8(al, rgb, dep, [0, 1, 2], 2)
        res = at.attribute_pixel(st, 0, 5, 5, first, second, tol, tol)
        assert res["attributed"] and res["kinds"] == ["median"] and res["events"][0][0] == 0 and res["events"][0][2] < at.WIN_MEDIAN
        assert at.attribute_pixel(st, 0, 5, 5, second, first, tol, tol)["attributed"]
        # the reference's hole (forward.cu:368): T (1 - alpha) == 0.5 exactly fires at neither contributor -> the pixel keeps (15, 0, 0)
        never = _pixel8(al, rgb, dep, [0, 1, 2], None)
        res = at.attribute_pixel(st, 0, 5, 5, first, never, tol, tol)
        assert res["attributed"] and set(res["kinds"]) <= {"median", "median_exact_half"} and "median_exact_half" in res["kinds"]
        assert not at.attribute_pixel(st, 0, 5, 5, first, third, tol, tol)["attributed"]      # the third Gaussian is never the median
        assert not at.attribute_pixel(st, 0, 5, 5, first, first, tol, tol)["attributed"]
    # well outside the window: the second Gaussian as median is a bug, not a flip
    st = _hand_state(0.3)
    st["conic_opacity"][0, 3] = 0.51
    al = [float(st["conic_opacity"][k, 3]) for k in range(3)]
    rgb, dep = st["rgb"].astype(np.float64), st["depths"].astype(np.float64)
    assert not at.attribute_pixel(st, 0, 5, 5, _pixel8(al, rgb, dep, [0, 1, 2], 0), _pixel8(al, rgb, dep, [0, 1, 2], 1), tol, tol)["attributed"]
    # the "forced next" branch with no next contributor: the median stays unset in that implementation
    st = _hand_state(0.3)
    st["conic_opacity"][0, 3] = 0.5 * (1 + 1.2e-7)
    st["ranges"] = np.array([[0, 1]], np.uint32)
    al = [float(st["conic_opacity"][0, 3])]
    a = _pixel8(al, rgb[:1], dep[:1], [0], 0)
    b = _pixel8(al, rgb[:1], dep[:1], [0], None)
    assert at.attribute_pixel(st, 0, 5, 5, a, b, tol, tol)["attributed"]


def test_attribute_images_with_median_flags_a_differing_median_id():
    st, cam = _oracle_state()
    W, H = cam.width, cam.height
    imgs_a = dict(color=st["color"], depth=st["depth"], opacity=st["opacity"], median=st["median"])
    imgs_b = {k: v.copy() for k, v in imgs_a.items()}
    st_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) and k in ("radii", "means2D") else v) for k, v in st.items()}
    ys, xs = np.nonzero(st["median"][1] > 0)
    y, x = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
    imgs_b["median"][2, y, x] += 1.0                          # a wrong median id, all other channels equal
    rep = at.attribute_images(st_t, W, H, imgs_a, imgs_b, tol=1e-5, depth_scale=20.0)
    assert rep["flagged"] == 1 and rep["attributed"] == 0 and rep["unattributed"][0]["pixel"] == [x, y]
    rep = at.attribute_images(st_t, W, H, imgs_a, imgs_a, tol=1e-5, depth_scale=20.0)
    assert rep["flagged"] == 0
