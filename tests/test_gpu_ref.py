"""-m gpu: the HIP path against the REFERENCE's own kernels (oracle/_ref/libgsref.so, hipified test-only from
/root/reference by oracle/build_ref.sh, prebuilt in the dev container and shipped with the snapshot) and
against the committed golden fixtures those kernels produced.

Tolerance (BASELINE.json north_star): 1e-5 abs on rendered RGB / depth / opacity ON EVERY PIXEL.  The reference and
this implementation use different exp() (ocml expf vs the explicit polynomial) and -- the reference source being
compiled by hipcc with its default contraction, not by nvcc: there is no NVIDIA binary to compare with -- possibly a
different FMA contraction, so a value whose alpha / T / power sits within rounding distance of a threshold
(alpha < 1/255, T (1 - alpha) < 1e-4, power > 0; forward.cu:341-356) can take the other branch and the pixel then
changes by up to alpha*T*c ~ 4e-3.  Such pixels are not merely counted: EVERY pixel that differs by more than 1e-5 must
be ATTRIBUTED to such an event by tests/attribution.py -- a float64 replay of the pixel's list in which only decisions
inside stated windows of their thresholds may be taken either way has to reproduce this implementation's value with
one set of decisions and the reference's value with another.  `unattributed == 0` is asserted; the counts go to
profiles/r06_parity.json as a report (r05_parity.json: round 5's).  Round 6: the median map's depth, weight and id
channels are compared and attributed like the other five (forward.cu:366-374, 392-394).

Both compositing modes of the library are held against the reference here: the bit-exact default (`fast_exp` = 0, the
mode the CPU oracle pins to the bit) and `fast_exp` = 1 (v_exp_f32; include/gsrast.h gsr_options.fast_exp) -- the same
scenes, the same attribution, the same gradient bounds (forward.cu:338-361, backward.cu:519-540 are what both restate).

THE REFERENCE'S OWN NOISE FLOOR.  oracle/build_ref.sh builds the reference sources twice (hipcc's default contraction;
-ffp-contract=off).  Two legitimate builds of the same kernels differ from each other at threshold events exactly as
this library differs from either, and two RUNS of one build differ in every gradient (float atomicAdd in no fixed order,
backward.cu:559-607).  `_noise_floor` measures both per configuration; the gradient tolerance of this library against
the reference is GRAD_K x that measured floor (not a constant chosen by the builder), and the flip counts are reported
next to the reference-vs-reference flip counts.
"""
import glob
import os

import numpy as np
import pytest
import torch

from gaustudio_amd import scenes

import attribution
import ref_util
from util import hip_backward_raw, hip_forward, scene_kwargs, to_np

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_FILES = sorted(glob.glob(os.path.join(GOLD, "ref_*.npz")))
GRAD_KEYS = ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
# gradient tolerance against the reference = GRAD_K x the reference's OWN floor, measured in the same test on the same
# inputs: how far the reference's two builds (and two runs of one build) are from each other, relative to each tensor's
# scale.  The MAXIMUM deviation is set by the handful of threshold flips of the frame (each build flips at different
# pixels, and a flip lands in whichever of the eight tensors its Gaussian weighs most), so the floor for the maximum is
# the largest build-vs-build deviation over the eight tensors of the configuration; the MEAN deviation is arithmetic
# noise and is compared per tensor.  Measured (profiles/r04_parity.json): floors 1.9e-4 (C1) .. 3.9e-3 (C5); this library
# sits at 0.1x .. 2.0x its configuration's floor in both modes.  Rounds 1-3 asserted a builder-chosen 5e-4.
# Round 5: the floor is the maximum over five reference runs (three of the default build, two of the -ffp-contract=off build;
# _reference_runs) instead of one build pair + one run pair; the measured ratios max_rel / floor of every configuration, mode and
# tensor, and the floor of every pair, are in profiles/r05_parity.json (`max_rel_over_config_floor`,
# `reference_floor_pairs_max_rel_any_tensor`).
GRAD_K = 3.0
# round 6: gradients of the Gaussians that are in no flagged pixel's list, with the median-depth gradient absent on both sides
# (_away_from_events), relative to each tensor's largest element.  Measured on the MI355X over C1-C5
# (profiles/r06_parity.json `grads_no_median_away_from_events`): at most 9.5e-6 in the reproducible mode (C2, dL_dmeans2D; flips below
# the 1e-5 image tolerance are not exempt) and 1.6e-6 in the default fast_exp mode, with 0-1.1 % of the Gaussians exempt; the same
# tensors reach 1e-5 .. 4e-4 INSIDE the event lists.  Bound = 3x the largest observation.
AWAY_TOL = float(os.environ.get("GSR_AWAY_TOL", "3e-5"))


def _compare(hs, ref, config, W, H):
    radii = to_np(hs["radii"])
    assert (radii != ref["radii"].numpy()).sum() <= 1e-5 * radii.size
    # a radius that lands on the other side of a ceil() changes that Gaussian's getRect area: the counts follow the radii
    assert abs(hs["num_rendered"] - ref["num_rendered"]) <= 1e-5 * ref["num_rendered"]
    stats = {"radii_differ": int((radii != ref["radii"].numpy()).sum()),
             "num_rendered": [int(hs["num_rendered"]), int(ref["num_rendered"])]}
    # round 6: the median map (depth, weight, Gaussian id; forward.cu:366-374, 392-394) is compared like colour / depth / opacity --
    # it is THE output gs-extract-mesh feeds to TSDF fusion (gaustudio/scripts/extract_mesh.py:104, renderers/base.py:52)
    ours = {k: to_np(hs[k]) for k in ("color", "depth", "opacity", "median")}
    theirs = {k: ref[k].numpy() for k in ("color", "depth", "opacity", "median")}
    for k in ("color", "depth", "opacity"):
        d = np.abs(ours[k].astype(np.float64) - theirs[k].astype(np.float64))
        stats[k] = {"over_1e-5": int((d > 1e-5).sum()), "values": int(d.size), "max_abs": float(d.max())}
    mid_a, mid_b = ours["median"][2], theirs["median"][2]
    same_id = mid_a == mid_b
    stats["median_id_differ"] = int((~same_id).sum())
    for ch, name in ((0, "median_depth"), (1, "median_weight")):
        d = np.abs(ours["median"][ch].astype(np.float64) - theirs["median"][ch].astype(np.float64))
        stats[name] = {"over_1e-5": int((d > 1e-5).sum()), "over_1e-5_where_ids_agree": int(((d > 1e-5) & same_id).sum()),
                       "values": int(d.size), "max_abs": float(d.max()), "max_abs_where_ids_agree": float(d[same_id].max(initial=0.0))}
    assert stats["median_id_differ"] <= max(4, 1e-4 * mid_a.size), "median id"
    # every pixel beyond 1e-5 in ANY of the eight channels (colour, depth in scene units, opacity, median depth / weight; a median id
    # that differs at all) must be a demonstrated threshold event: alpha vs 1/255, T (1 - alpha) vs 1e-4, power vs 0, a radius on the
    # other side of a ceil(), or T (1 - alpha) vs 0.5 for the median (attribution.WIN_MEDIAN)
    rep = attribution.attribute_images(hs, W, H, ours, theirs, tol=1e-5, depth_scale=20.0, radii_b=ref["radii"].numpy())
    stats["attribution"] = {k: rep[k] for k in ("flagged", "attributed", "by_kind", "max_margin")}
    stats["attribution"]["unattributed"] = len(rep["unattributed"])
    stats["attribution"]["windows"] = {"alpha_rel": attribution.WIN_ALPHA, "T_rel": attribution.WIN_T, "power_rel": attribution.WIN_POWER,
                                       "median_rel": attribution.WIN_MEDIAN}
    assert not rep["unattributed"], f"{config}: {len(rep['unattributed'])} of {rep['flagged']} differing pixels are NOT threshold events: {rep['unattributed'][:3]}"
    stats["_event_pixels"] = [tuple(e["pixel"]) for e in rep["events"]]
    return stats


def _away_from_events(hs, sc, cam, D, kw, grads, runs_ref, event_pixels, W, fast_exp):
    """Round 6: the gradients AWAY from the threshold events, without the median-depth gradient.  The two things that make the
    GRAD_K x floor bound above loose are (1) the frame's handful of threshold flips -- a flip at a pixel changes T for everything
    behind it there, i.e. the gradients of the Gaussians in that pixel's list -- and (2) the median-depth gradient, which the
    reference routes by a transmittance reconstructed by division (backward.cu:566-569, ill-conditioned at 0.5) and this library
    by the forward's own decision.  Take both away -- the median gradient absent in BOTH implementations, the
    Gaussians of the flagged pixels' lists exempt -- and what remains is arithmetic: fp32 sums in another order, ocml expf vs the
    polynomial / v_exp_f32, rcp vs divide.  Returns {tensor: (max_rel away from events, max_rel in event lists, exempt count)}."""
    g_nm = [grads[0], grads[1], torch.zeros_like(grads[2]), grads[3]]
    ref = ref_util.run(sc, cam, D, kw, g_nm)
    hb = hip_backward_raw(hs, sc, cam, D, kw, [g_nm[0], g_nm[1], None, g_nm[3]], options=dict(fast_exp=fast_exp))
    P = sc.means3D.shape[0]
    exempt = np.zeros(P, bool)
    gx = (W + 15) // 16
    nc = to_np(hs["n_contrib"]).astype(np.int64)
    for (x, y) in event_pixels:
        ids = attribution.tile_list(hs, (y // 16) * gx + x // 16)
        exempt[ids[:int(nc[y, x]) + 2]] = True          # (+2: the flipped contributor may be the one just behind the last one this side applied)
    out = {}
    for k in GRAD_KEYS:
        b = ref[k].numpy()
        if b.size == 0:
            continue
        a = to_np(hb[k]).reshape(P, -1).astype(np.float64)
        b = b.reshape(P, -1).astype(np.float64)
        scale = max(float(np.abs(b).max()), 1e-30)
        d = np.abs(a - b).max(1) / scale
        out[k] = {"max_rel_away_from_events": float(d[~exempt].max(initial=0.0)), "max_rel_in_event_lists": float(d[exempt].max(initial=0.0)),
                  "mean_rel": float(d.mean())}
    out["_exempt_gaussians"] = int(exempt.sum())
    return out


def _dump_parity(config, stats, section="configs"):
    """GSR_DUMP_PARITY=1: merge the measured report into gpurun_out/r06_parity.json (copied to profiles/ by hand)."""
    if os.environ.get("GSR_DUMP_PARITY") != "1":
        return
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "r06_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    data = json.load(open(out)) if os.path.exists(out) else {
        "what": "HIP path (both compositing modes) vs the reference's own kernels (oracle/_ref/libgsref.so, the reference source "
                "compiled by hipcc) on an MI355X: values differing by more than 1e-5 abs, each attributed to a threshold event "
                "(tests/attribution.py); `reference_vs_reference` = the same measurement between two builds of the reference "
                "(default contraction vs -ffp-contract=off) and between two runs of one build (atomic order)",
        "configs": {}, "reference_vs_reference": {}}
    data.setdefault(section, {})[config] = stats
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)


_REF_RUNS = {}       # config -> reference runs of the configuration under test (one entry: the C4 / C5 tensors are large)


def _reference_runs(config, sc, cam, D, kw, grads):
    """The reference's kernels on this scene: the default build twice (two runs: the backward's atomics land in a different
    order) and the -ffp-contract=off build once."""
    if config not in _REF_RUNS:
        _REF_RUNS.clear()
        runs = {"a": ref_util.run(sc, cam, D, kw, grads), "a2": ref_util.run(sc, cam, D, kw, grads),
                "b": ref_util.run(sc, cam, D, kw, grads, variant="nocontract")}
        # THE FLOOR (round 5): the largest gradient deviation among FIVE runs of the reference -- default build x 3, the
        # -ffp-contract=off build x 2; pairs (a2, a), (a3, a), (b, a), (b2, a), (b2, b).  Rounds 3-4 took one build pair and one run
        # pair: the maximum over a frame's few threshold flips and over the atomics' order is a heavy-tailed statistic, a single
        # sample of it put this library at up to 2.59x (C4, dL_drotations) of a GRAD_K = 3 limit.  The two
        # extra runs are compared and dropped at once (a C4 run's gradients are 1.4 GB on the host).
        pairs = {}

        def note(name, x, y):
            pairs[name] = {k: _grad_rel(x[k].numpy(), y[k].numpy()) for k in GRAD_KEYS if y[k].numel()}
        note("a2_vs_a", runs["a2"], runs["a"])
        note("b_vs_a", runs["b"], runs["a"])
        extra = ref_util.run(sc, cam, D, kw, grads)
        note("a3_vs_a", extra, runs["a"])
        extra = ref_util.run(sc, cam, D, kw, grads, variant="nocontract")
        note("b2_vs_a", extra, runs["a"])
        note("b2_vs_b", extra, runs["b"])
        del extra
        runs["floor_pairs"] = pairs
        runs["floors"] = {k: (max(p[k][0] for p in pairs.values()), max(p[k][1] for p in pairs.values())) for k in pairs["a2_vs_a"]}
        # round 4's criterion (one run pair + one build pair), asserted as well since round 6 (a maximum over more samples
        # can only grow, so the five-run floor alone is the looser test); which of the two binds is recorded in the parity report
        two = [pairs["a2_vs_a"], pairs["b_vs_a"]]
        runs["floors_two_pairs"] = {k: (max(p[k][0] for p in two), max(p[k][1] for p in two)) for k in pairs["a2_vs_a"]}
        _REF_RUNS[config] = runs
    return _REF_RUNS[config]


def _grad_rel(x, y):
    """(max, mean) of |x - y| relative to the scale of y."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64).reshape(x.shape)
    scale = max(float(np.abs(y).max()), 1e-30)
    d = np.abs(x - y)
    return float(d.max() / scale), float(d.mean() / scale)


def _noise_floor(hs, runs, W, H):
    """What two legitimate builds / two runs of the REFERENCE differ by, measured like `_compare` measures this library:
    values beyond 1e-5 between the builds (attributed with the same machinery, from this library's bit-checked records),
    integer outputs, and the gradient differences (build vs build: threshold flips + contraction; run vs run: atomics)."""
    a, a2, b = runs["a"], runs["a2"], runs["b"]
    fl = {"radii_differ": int((a["radii"] != b["radii"]).sum()), "num_rendered": [int(a["num_rendered"]), int(b["num_rendered"])],
          "median_id_differ": int((a["median"][2] != b["median"][2]).sum())}
    ia = {k: a[k].numpy() for k in ("color", "depth", "opacity", "median")}
    ib = {k: b[k].numpy() for k in ("color", "depth", "opacity", "median")}
    for k in ia:
        d = np.abs(ia[k].astype(np.float64) - ib[k].astype(np.float64))
        if k == "median":
            fl["median_depth_over_1e-5"], fl["median_weight_over_1e-5"] = int((d[0] > 1e-5).sum()), int((d[1] > 1e-5).sum())
        else:
            fl[k] = {"over_1e-5": int((d > 1e-5).sum()), "values": int(d.size), "max_abs": float(d.max())}
        # forward is deterministic: two runs of one build agree to the bit
        assert np.array_equal(ia[k], a2[k].numpy()), f"reference forward is not run-to-run deterministic ({k})"
    rep = attribution.attribute_images(hs, W, H, ia, ib, tol=1e-5, depth_scale=20.0, radii_b=b["radii"].numpy(), tol_a=1e-5,
                                       radii_a=a["radii"].numpy())
    fl["attribution"] = {k: rep[k] for k in ("flagged", "attributed", "by_kind", "max_margin")}
    fl["attribution"]["unattributed"] = len(rep["unattributed"])
    fl["attribution"]["unattributed_detail"] = rep["unattributed"][:3]
    fl["grads_build_vs_build"], fl["grads_run_vs_run"] = {}, {}
    for k in GRAD_KEYS:
        if a[k].numel() == 0:
            continue
        mx, mn = _grad_rel(b[k].numpy(), a[k].numpy())
        fl["grads_build_vs_build"][k] = {"max_rel": mx, "mean_rel": mn}
        mx, mn = _grad_rel(a2[k].numpy(), a[k].numpy())
        fl["grads_run_vs_run"][k] = {"max_rel": mx, "mean_rel": mn}
    return fl


@pytest.mark.parametrize("fast_exp", [0, 1], ids=["exact", "fast_exp"])
@pytest.mark.parametrize("P,W,H,D", [(10000, 400, 400, 0), (300000, 800, 800, 3), (1000000, 1920, 1080, 3),
                                     (5000000, 1297, 840, 3), (2500000, 3840, 2160, 3)],
                         ids=["C1", "C2", "C3", "C4", "C5"])
def test_hip_vs_reference_kernels_at_baseline_configs(request, P, W, H, D, fast_exp):
    """BASELINE configs C1, C2, the full-size headline C3 and one view of the C4 / C5 sizes (5 M Gaussians with ~3.7 k
    entries per tile: the long-list sort and row-flag regime; a 4K frame), forward and backward, BOTH compositing modes
    (bit-exact default; fast_exp = hardware exp), against the reference's own kernels.
    oracle/_ref/libgsref.so is built in the dev container (oracle/build_ref.sh) and travels with the snapshot: its
    absence on a GPU box is a FAILURE, not a skip -- this comparison is what pins the parity claim."""
    if not (ref_util.available() and ref_util.available("nocontract")):
        pytest.fail("oracle/_ref/libgsref.so / libgsref_nocontract.so missing: run oracle/build_ref.sh in the dev container "
                    "(needs /root/reference) before shipping the tree to the GPU box")
    import gaustudio_amd
    config = request.node.callspec.id.split("-")[0]
    mode = "fast_exp" if fast_exp else "exact"
    cam = scenes.make_camera(W, H)
    sc = scenes.make_scene(P, cam, seed=0)
    kw = scene_kwargs(sc, True, False)
    grads = scenes.make_output_grads(cam)
    runs = _reference_runs(config, sc, cam, D, kw, grads)
    ref = runs["a"]
    with gaustudio_amd.options(fast_exp=bool(fast_exp)):
        hs = hip_forward(sc, cam, D, kw)
    stats = _compare(hs, ref, config, W, H)
    if not fast_exp:
        floor = _noise_floor(hs, runs, W, H)
        _dump_parity(config, floor, section="reference_vs_reference")
        assert floor["attribution"]["unattributed"] == 0, f"two builds of the reference differ at a pixel that is no threshold event: {floor['attribution']['unattributed_detail']}"
    if config == "C3" and not fast_exp:
        # report: how often the reference's backward would route the median-depth gradient differently
        stats["median_gradient_census"] = attribution.median_gradient_census(hs, W, H)
    hb = hip_backward_raw(hs, sc, cam, D, kw, grads, options=dict(fast_exp=fast_exp), debug=True)   # debug: mode checked against the forward's record
    stats["grads"] = {}
    floors = runs["floors"]                                  # per tensor: (max_rel, mean_rel) over five reference runs (_reference_runs)
    floor_max = max(f[0] for f in floors.values())          # the configuration's floor for a MAXIMUM: see GRAD_K
    floors2 = runs["floors_two_pairs"]
    floor2_max = max(f[0] for f in floors2.values())
    stats["reference_floor_max_rel_any_tensor"] = floor_max
    stats["reference_floor_two_pairs_max_rel_any_tensor"] = floor2_max
    stats["binding_floor"] = "two_pairs (round 4's criterion)" if floor2_max < floor_max else "both equal"
    stats["reference_floor_pairs_max_rel_any_tensor"] = {n: max(v[0] for v in p.values()) for n, p in runs["floor_pairs"].items()}
    for k in floors:
        a = to_np(hb[k]); b = ref[k].numpy().reshape(a.shape)
        mx, mn = _grad_rel(a, b)
        mx_b = _grad_rel(a, runs["b"][k].numpy())[0]          # (report) distance to the -ffp-contract=off build of the reference
        stats["grads"][k] = {"max_rel": mx, "mean_rel": mn, "max_rel_vs_nocontract_build": mx_b, "reference_floor_max_rel": floors[k][0],
                             "reference_floor_mean_rel": floors[k][1], "max_rel_over_config_floor": mx / max(floor_max, 1e-30)}
        # both sides sum thousands of fp32 terms per Gaussian in different orders, plus rare branch flips: no further from
        # the reference than GRAD_K x what the reference is from itself
        assert mx <= GRAD_K * floor_max, (k, mode, mx, floor_max)
        assert mn <= GRAD_K * max(floors[k][1], 1e-9), (k, mode, mn, floors[k][1])
        assert mx <= GRAD_K * floor2_max, (k, mode, mx, floor2_max, "round 4's two-pair floor")
        assert mn <= GRAD_K * max(floors2[k][1], 1e-9), (k, mode, mn, floors2[k][1], "round 4's two-pair floor")
    # round 6: away from the events, without the median gradient (see _away_from_events): a bound on the ARITHMETIC difference
    event_pixels = stats.pop("_event_pixels")
    stats["grads_no_median_away_from_events"] = away = _away_from_events(hs, sc, cam, D, kw, grads, runs, event_pixels, W, fast_exp)
    for k in GRAD_KEYS:
        if k in away:
            assert away[k]["max_rel_away_from_events"] <= AWAY_TOL, (k, mode, away[k])
    _dump_parity(config, stats, section="configs" if not fast_exp else "configs_fast_exp")


@pytest.mark.parametrize("fast_exp", [0, 1], ids=["exact", "fast_exp"])
@pytest.mark.parametrize("path", REF_FILES, ids=[os.path.basename(p)[:-4] for p in REF_FILES])
def test_hip_vs_committed_reference_fixtures(path, fast_exp):
    """Same comparison against the committed fixtures (works even where libgsref.so is absent), both compositing modes."""
    import gaustudio_amd
    z = np.load(path)
    cam = scenes.Cam(int(z["width"]), int(z["height"]), float(z["tanfovx"]), float(z["tanfovy"]),
                     torch.from_numpy(z["viewmatrix"]), torch.from_numpy(z["projmatrix"]), torch.from_numpy(z["campos"]))
    P = z["means3D"].shape[0]
    dummy = torch.zeros(P, 1)
    sc = scenes.Scene(torch.from_numpy(z["means3D"]), dummy, dummy, torch.from_numpy(z["opacities"]), dummy)
    kw = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    bg = torch.from_numpy(z["bg"])
    D, mod = int(z["D"]), float(z["scale_modifier"])
    with gaustudio_amd.options(fast_exp=bool(fast_exp)):
        hs = hip_forward(sc, cam, D, kw, scale_modifier=mod, bg=bg)
    assert hs["num_rendered"] == int(z["ref_num_rendered"])
    assert np.array_equal(to_np(hs["radii"]), z["ref_radii"])
    for k in ("color", "depth", "opacity"):
        assert np.abs(to_np(hs[k]) - z["ref_" + k]).max() <= 1e-5, k
    med = to_np(hs["median"])
    assert np.array_equal(med[2], z["ref_median"][2])                       # median Gaussian id (forward.cu:372, 394)
    assert np.abs(med[0] - z["ref_median"][0]).max() <= 1e-5, "median depth"      # forward.cu:370, 392
    assert np.abs(med[1] - z["ref_median"][1]).max() <= 1e-5, "median weight"     # forward.cu:371, 393
    grads = [torch.from_numpy(z[k]) for k in ("grad_color", "grad_depth", "grad_median", "grad_opacity")]
    hb = hip_backward_raw(hs, sc, cam, D, kw, grads, scale_modifier=mod, bg=bg, options=dict(fast_exp=fast_exp), debug=True)
    for k in GRAD_KEYS:
        ref = z["ref_" + k]
        if ref.size == 0:
            continue
        a = to_np(hb[k]).reshape(ref.shape)
        assert np.abs(a - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-12, k


def test_full_size_c3_properties():
    """Size-independent properties at the headline size (1M Gaussians, 1920x1080): opacity == 1 - final_T,
    colour = 0 where nothing contributed, per-tile lists sorted by (depth, id), ranges partition [0, R)."""
    cam = scenes.make_camera(1920, 1080)
    sc = scenes.make_scene(1_000_000, cam, seed=0)
    hs = hip_forward(sc, cam, 3, scene_kwargs(sc, True, False))
    assert torch.equal(hs["opacity"][0], 1 - hs["final_T"])
    empty = hs["n_contrib"] == 0
    assert float(hs["color"][:, empty].abs().sum()) == 0.0 and float(hs["opacity"][0][empty].abs().sum()) == 0.0
    assert bool((hs["median"][0][empty] == 15.0).all())
    r = hs["ranges"].long()
    assert int(r[0, 0]) == 0 and int(r[-1, 1]) == hs["num_binned"] and bool((r[1:, 0] == r[:-1, 1]).all())
    assert hs["num_binned"] < hs["num_rendered"]                                       # tight binning vs the reference's count
    pl = hs["point_list"].long()
    depth_bits = hs["depths"].view(torch.int32).long()[pl]
    key = depth_bits * (1 << 32) + pl
    tile_of = torch.repeat_interleave(torch.arange(r.shape[0], device=pl.device), (r[:, 1] - r[:, 0]))
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool((key[1:][same_tile] > key[:-1][same_tile]).all()), "per-tile lists must be strictly (depth, id) sorted"
    assert int(hs["tiles_touched"].long().sum()) == hs["num_binned"]
