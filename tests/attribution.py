"""TEST INFRASTRUCTURE: threshold-event attribution for pixels that differ between two forward implementations.

The compositing loop (forward.cu:330-380) takes three data-dependent decisions per (pixel, Gaussian):

    power > 0            -> skip                       (forward.cu:341-342)
    alpha < 1/255        -> skip                       (forward.cu:349-350)
    T (1 - alpha) < 1e-4 -> the pixel is done          (forward.cu:351-356)

Two implementations that evaluate exp() / the FMA contraction differently agree to ~1e-6 on every smooth quantity,
but where one of these quantities lies within rounding distance of its threshold they can take opposite branches, and
the pixel then changes by up to alpha*T*c (~4e-3).  Counting such pixels is not a proof that they ARE flips.  This module
proves it constructively: for a pixel whose two values differ by more than the tolerance, the
tile's list is replayed in float64 from the (bit-identical) per-Gaussian records; every decision whose operand lies
inside a stated relative window of its threshold is a branch point, and the replay explores both branches.  The pixel
is ATTRIBUTED iff
  * one leaf of that decision tree reproduces implementation A's value (all five channels) and
  * a DIFFERENT leaf reproduces implementation B's value to the same tolerance,
i.e. the whole difference is explained by decisions taken inside the windows.  A fourth kind of event covers the
<= 1e-5 of the Gaussians whose integer `radii` differ between the implementations (a ceil() taken on the other side:
the Gaussian is binned into a different set of tiles): membership of such a Gaussian in the pixel's list is a branch
point as well.  Everything else -- a wrong list order, a wrong record, a wrong accumulation -- reproduces neither
value and is reported as unattributed.
"""
import numpy as np

ALPHA_MIN = 1.0 / 255.0
# relative windows around the thresholds inside which two implementations may legitimately decide differently.  Round 4:
# tied to MEASUREMENT -- each window is <= 3x the largest margin of any attributed event observed on the MI355X over
# C1-C5, the adversarial scenes and the fuzz seeds, in ALL of: this library (bit-exact mode, fast_exp mode) vs the
# reference's kernels, fast_exp vs the CPU oracle, and the reference's default build vs its -ffp-contract=off build
# (profiles/r04_parity.json: reference_vs_reference alpha 1.79e-6 / T 7.0e-7; this library vs the reference alpha
# 1.79e-6 / T 4.51e-6).  Rounds 2-3 used 2e-5 / 5e-5 / 1e-5, chosen by the builder at 10x the observations.
WIN_ALPHA = 5.0e-6       # |alpha * 255 - 1|            (largest observed margin 1.79e-6)
WIN_T = 1.2e-5           # |T (1 - alpha) / 1e-4 - 1|   (largest observed 4.51e-6)
WIN_POWER = 2.0e-6       # |power| / (|a dx^2| + |b dx dy| + |c dy^2|): no event of this kind has ever been observed; ~16 ulp of the terms
# round 6: the median test `T > 0.5 && T (1 - alpha) < 0.5` (forward.cu:366-374), a fourth decision.  Measured on the MI355X over C1-C5,
# both modes of this library vs the reference's kernels and the reference's two builds against each other (profiles/r06_parity.json):
# 13 events, largest margin 1.06e-7 (`median_exact_half`, C5 fast_exp) / 6.4e-8 (`median`) -- i.e. one or two fp32 ulp of 0.5.
# Window = 3x that.  (The first run used 2e-5; nothing between 1.1e-7 and 2e-5 was ever needed.)
WIN_MEDIAN = float(__import__("os").environ.get("GSR_WIN_MEDIAN", "3.0e-7"))     # |T (1 - alpha) / 0.5 - 1|
MAX_LEAVES = 256


def tile_list(st, tile):
    """ids of the tile's depth-sorted list, from the decoded state of a forward (util.hip_forward / the oracle)."""
    r = np.asarray(st["ranges"][tile].cpu() if hasattr(st["ranges"], "cpu") else st["ranges"][tile]).astype(np.int64)
    pl = st["point_list"]
    ids = pl[int(r[0]):int(r[1])]
    return np.asarray(ids.cpu() if hasattr(ids, "cpu") else ids).astype(np.int64)


def _records(st, ids):
    g = lambda k: np.asarray(st[k][ids].cpu() if hasattr(st[k], "cpu") else st[k][ids]).astype(np.float64)
    return g("means2D"), g("conic_opacity"), g("rgb"), g("depths")


def _explore(n, power, mag, alpha_raw, alpha, rgb, dep, include, ids=None):
    """Decision tree of one pixel.  A leaf = (values, events); values = [r, g, b, depth, opacity] and, when `ids` is given,
    + [median depth, median weight, median id] (forward.cu:366-374, 392-394).  The median test `T > 0.5 && T (1 - alpha) < 0.5`
    is a fourth data-dependent decision: where T (1 - alpha) lies inside WIN_MEDIAN of 0.5 one implementation records THIS
    contributor and the other the NEXT one (whose T is that same product) -- both branches are explored (`med` mode 1 = the
    next contributor fires whatever float64 says, mode 2 = it does not) -- and a third, `median_exact_half`: the product is
    exactly 0.5 in fp32 and the test fires at neither."""
    leaves = []
    with_med = ids is not None

    def leaf(c0, c1, c2, d, T, med, events):
        v = [c0, c1, c2, d, 1.0 - T]
        if with_med:
            v += [med[0], med[1], med[2]]
        leaves.append((np.array(v), tuple(events)))

    def med_step(i, T, test_T, med, events):
        """-> (med after contributor i on the default branch, [(med, events) of the alternative branch] or [])."""
        if not with_med:
            return med, []
        md, mw, mid, mode = med
        fired = (dep[i], alpha[i] * T, float(ids[i]), 0)
        if mode == 1:
            return fired, []
        if mode == 2:
            return (md, mw, mid, 0), []
        fire = T > 0.5 and test_T < 0.5
        alt = []
        margin = abs(test_T / 0.5 - 1.0)
        if margin < WIN_MEDIAN and T > 0.5 * (1.0 + WIN_MEDIAN):
            ev = list(events) + [(i, "median", margin)]
            alt = [((md, mw, mid, 1), ev)] if fire else [((fired[0], fired[1], fired[2], 2), ev)]
            # the reference's own hole: `T > 0.5f && test_T < 0.5` (forward.cu:368) fires NOWHERE when an implementation's fp32
            # product T (1 - alpha) is EXACTLY 0.5 -- not below 0.5 at this contributor, not above 0.5 at the next: the pixel keeps
            # (15, 0, 0) although it saturates.  Observed on the MI355X between the reference's own two builds (C4: 1 pixel of 1.09 M)
            # and between this library and the reference (C5: 2 of 8.3 M); this library restates the same test and has the same hole.
            alt.append(((md, mw, mid, 2), list(events) + [(i, "median_exact_half", margin)]))
        return (fired if fire else med), alt

    def rec(i, T, c0, c1, c2, d, med, events):
        while i < n:
            if len(leaves) >= MAX_LEAVES:
                return
            inc = True if include is None else include[i]
            if inc is False:
                i += 1
                continue
            if inc is None:                                            # in one implementation's list only
                rec2 = list(events) + [(i, "radius", 0.0)]
                # branch A: absent
                rec(i + 1, T, c0, c1, c2, d, med, rec2)
                # branch B (fall through): present
            # --- power > 0 ---
            p_skip = power[i] > 0.0
            p_margin = abs(power[i]) / mag[i] if mag[i] > 1e-200 else np.inf
            if p_margin < WIN_POWER:
                if p_skip:   # alternative: treat as not skipped -> needs the rest of the body; handled by flipping below
                    _body(i, T, c0, c1, c2, d, med, list(events) + [(i, "power", p_margin)], force_alpha=None)
                else:
                    rec(i + 1, T, c0, c1, c2, d, med, list(events) + [(i, "power", p_margin)])
            if p_skip:
                i += 1
                continue
            # --- alpha < 1/255 ---
            a_skip = alpha[i] < ALPHA_MIN
            a_margin = abs(alpha_raw[i] / ALPHA_MIN - 1.0)
            if a_margin < WIN_ALPHA:
                if a_skip:
                    _body(i, T, c0, c1, c2, d, med, list(events) + [(i, "alpha", a_margin)], force_alpha=True)
                else:
                    rec(i + 1, T, c0, c1, c2, d, med, list(events) + [(i, "alpha", a_margin)])
            if a_skip:
                i += 1
                continue
            # --- T (1 - alpha) < 1e-4 ---
            test_T = T * (1.0 - alpha[i])
            t_stop = test_T < 1e-4
            t_margin = abs(test_T / 1e-4 - 1.0)
            if t_margin < WIN_T:
                if t_stop:
                    # alternative: not done -> apply and go on
                    _apply_and_go(i, T, test_T, c0, c1, c2, d, med, list(events) + [(i, "T", t_margin)])
                else:
                    leaf(c0, c1, c2, d, T, med, list(events) + [(i, "T", t_margin)])
            if t_stop:
                leaf(c0, c1, c2, d, T, med, events)
                return
            w = alpha[i] * T
            c0 += rgb[i, 0] * w
            c1 += rgb[i, 1] * w
            c2 += rgb[i, 2] * w
            d += dep[i] * w
            med, alt = med_step(i, T, test_T, med, events)
            for m2, ev2 in alt:
                rec(i + 1, test_T, c0, c1, c2, d, m2, ev2)
            T = test_T
            i += 1
        leaf(c0, c1, c2, d, T, med, events)

    def _apply_and_go(i, T, test_T, c0, c1, c2, d, med, events):
        w = alpha[i] * T
        args = (i + 1, test_T, c0 + rgb[i, 0] * w, c1 + rgb[i, 1] * w, c2 + rgb[i, 2] * w, d + dep[i] * w)
        m1, alt = med_step(i, T, test_T, med, events)
        for m2, ev2 in alt:
            rec(*args, m2, ev2)
        rec(*args, m1, events)

    def _body(i, T, c0, c1, c2, d, med, events, force_alpha):
        # the instance is evaluated although the default decision skipped it: alpha test (unless forced), T test, apply
        if force_alpha is None and alpha[i] < ALPHA_MIN:
            rec(i + 1, T, c0, c1, c2, d, med, events)
            return
        test_T = T * (1.0 - alpha[i])
        if test_T < 1e-4:
            leaf(c0, c1, c2, d, T, med, events)
            return
        _apply_and_go(i, T, test_T, c0, c1, c2, d, med, events)

    rec(0, 1.0, 0.0, 0.0, 0.0, 0.0, (15.0, 0.0, 0.0, 0), [])
    return leaves


def attribute_pixel(st, tile, px, py, val_a, val_b, tol_a, tol_b, maybe_ids=(), extra_ids=()):
    """val_* = (r, g, b, depth, opacity[, median depth, median weight, median id]) of the two implementations at pixel
    (px, py) of `tile`; tol_* = per-channel
    absolute tolerances for matching a replay leaf to them (the replay is float64, the implementations accumulate in
    fp32).  maybe_ids: Gaussians of the list whose membership is undecided (their radii differ between the
    implementations); extra_ids: Gaussians NOT in this list that the other implementation may have binned here.
    Returns dict(attributed, events, kinds, leaves)."""
    ids = tile_list(st, tile)
    include = None
    if len(extra_ids):
        # insert by depth (stable: equal depths by id), membership undecided
        depths_all = st["depths"]
        dep_of = lambda i: float(depths_all[int(i)])
        cur = [(dep_of(i), int(i), True) for i in ids] + [(dep_of(i), int(i), None) for i in extra_ids]
        cur.sort(key=lambda t: (t[0], t[1]))
        ids = np.array([c[1] for c in cur], dtype=np.int64)
        include = [c[2] for c in cur]
    if len(maybe_ids):
        include = [True] * len(ids) if include is None else include
        ms = set(int(i) for i in maybe_ids)
        include = [None if (int(g) in ms) else inc for g, inc in zip(ids, include)]
    xy, co, rgb, dep = _records(st, ids)
    va, vb = np.asarray(val_a, np.float64), np.asarray(val_b, np.float64)
    leaves = _explore(len(ids), *_terms(xy, co, float(px), float(py)), rgb, dep, include, ids if len(va) == 8 else None)
    hit_a = [k for k, (v, _) in enumerate(leaves) if np.all(np.abs(v - va) <= tol_a)]
    hit_b = [k for k, (v, _) in enumerate(leaves) if np.all(np.abs(v - vb) <= tol_b)]
    best = None
    for ka in hit_a:
        for kb in hit_b:
            ea, eb = set(leaves[ka][1]), set(leaves[kb][1])
            diff = ea ^ eb                                   # the decisions the two implementations took differently
            if not diff:
                continue
            cand = sorted(diff)
            if best is None or len(cand) < len(best):
                best = cand
    return {"attributed": best is not None, "events": best or [], "leaves": len(leaves),
            "matches_a": len(hit_a), "matches_b": len(hit_b),
            "kinds": sorted(set(e[1] for e in (best or [])))}


def _terms(xy, co, px, py):
    dx = xy[:, 0] - px
    dy = xy[:, 1] - py
    ta, tb, tc = co[:, 0] * dx * dx, co[:, 1] * dx * dy, co[:, 2] * dy * dy
    power = -0.5 * (ta + tc) - tb
    mag = np.abs(ta) + np.abs(tb) + np.abs(tc) + 1e-300      # all terms 0 (pixel on the centre): power is exactly 0 everywhere
    alpha_raw = co[:, 3] * np.exp(np.minimum(power, 0.0))
    alpha = np.minimum(0.99, alpha_raw)
    return power, mag, alpha_raw, alpha


def attribute_images(st, W, H, imgs_a, imgs_b, tol=1e-5, depth_scale=1.0, radii_b=None, max_pixels=4000, tol_a=None, radii_a=None):
    """imgs_* = dict(color[3,H,W], depth[1,H,W], opacity[1,H,W][, median[3,H,W]]) as numpy arrays of implementations A (the
    one `st` was decoded from) and B.  Every pixel with a channel differing by more than `tol` is attributed.  With `median`
    in both dicts the three median channels (depth, weight, Gaussian id; forward.cu:366-374, 392-394) are compared and
    replayed like the other five: a differing median id is a flagged pixel that needs a `median` (or earlier) event.
    tol_a / radii_a: when A is NOT the implementation `st` was decoded from either (two builds of the reference compared
    with each other, replayed from this library's records), A's values are matched with tolerance tol_a like B's, and a
    Gaussian whose integer radius differs between ANY two of (st, A, B) has undecided list membership.
    Returns dict(flagged, attributed, unattributed=[...], by_kind, max_margin, events=[...])."""
    keys = ["color", "depth", "opacity"] + (["median"] if ("median" in imgs_a and "median" in imgs_b) else [])
    a = np.concatenate([np.asarray(imgs_a[k]) for k in keys], 0).astype(np.float64)   # [5 or 8,H,W]
    b = np.concatenate([np.asarray(imgs_b[k]) for k in keys], 0).astype(np.float64)
    bad = (np.abs(a - b) > tol).any(0)
    ys, xs = np.nonzero(bad)
    gx = (W + 15) // 16
    out = {"flagged": int(len(ys)), "attributed": 0, "unattributed": [], "by_kind": {}, "max_margin": {}, "events": []}
    if len(ys) > max_pixels:
        out["unattributed"] = [f"{len(ys)} pixels differ: more than max_pixels={max_pixels}, not a handful of flips"]
        return out
    # Gaussians whose radii differ between the implementations (a ceil() on the other side)
    maybe = np.zeros(0, np.int64)
    rect_b = None
    if radii_b is not None:
        ra = np.asarray(st["radii"].cpu()).astype(np.int64)
        rb = np.asarray(radii_b).astype(np.int64)
        rc = rb if radii_a is None else np.asarray(radii_a).astype(np.int64)
        maybe = np.nonzero((ra != rb) | (ra != rc))[0]
        if len(maybe):
            m2 = np.asarray(st["means2D"][maybe].cpu()).astype(np.float64)
            rmax = np.maximum(np.maximum(ra[maybe], rb[maybe]), rc[maybe]).astype(np.float64)
            gy = (H + 15) // 16
            rect_b = (np.clip(((m2[:, 0] - rmax) // 16).astype(np.int64), 0, gx), np.clip(((m2[:, 1] - rmax) // 16).astype(np.int64), 0, gy),
                      np.clip(((m2[:, 0] + rmax + 15) // 16).astype(np.int64), 0, gx), np.clip(((m2[:, 1] + rmax + 15) // 16).astype(np.int64), 0, gy))
    # fp32 accumulation noise of the implementations against the float64 replay: colour/opacity values are O(1), depth
    # is in scene units
    tol_leaf = np.array([4e-6, 4e-6, 4e-6, 4e-6 * max(depth_scale, 1.0), 4e-6])
    if a.shape[0] == 8:       # median depth: one Gaussian's depth (scene units); weight: alpha * T; id: an integer, exact
        tol_leaf = np.concatenate([tol_leaf, [4e-6 * max(depth_scale, 1.0), 4e-6, 0.5]])
    for y, x in zip(ys.tolist(), xs.tolist()):
        tile = (y // 16) * gx + (x // 16)
        ids = tile_list(st, tile)
        mb, extra = [], []
        if rect_b is not None:
            tx, ty = x // 16, y // 16
            cover = (rect_b[0] <= tx) & (tx < rect_b[2]) & (rect_b[1] <= ty) & (ty < rect_b[3])
            cand = maybe[cover]
            inl = set(ids.tolist())
            mb = [int(g) for g in cand if int(g) in inl]
            extra = [int(g) for g in cand if int(g) not in inl]
        res = attribute_pixel(st, tile, x, y, a[:, y, x], b[:, y, x], tol_leaf if tol_a is None else np.maximum(tol_leaf, tol_a),
                              np.maximum(tol_leaf, tol), mb, extra)
        if res["attributed"]:
            out["attributed"] += 1
            for (_pos, kind, margin) in res["events"]:
                out["by_kind"][kind] = out["by_kind"].get(kind, 0) + 1
                out["max_margin"][kind] = max(out["max_margin"].get(kind, 0.0), float(margin))
            out["events"].append({"pixel": [x, y], "events": [[int(p), k, float(m)] for p, k, m in res["events"]]})
        else:
            out["unattributed"].append({"pixel": [x, y], "a": a[:, y, x].tolist(), "b": b[:, y, x].tolist(),
                                        "leaves": res["leaves"], "matches_a": res["matches_a"], "matches_b": res["matches_b"]})
    return out


def median_margin(st, tile, px, py):
    """Smallest relative distance to 0.5 of a transmittance the median test (forward.cu:366-374: T > 0.5 and
    T (1 - alpha) < 0.5) looks at, along the default walk of the pixel: a median id that differs between two
    implementations although the pixel's values agree must have such a transmittance within rounding distance of 0.5."""
    ids = tile_list(st, tile)
    xy, co, _rgb, _dep = _records(st, ids)
    power, _mag, _araw, alpha = _terms(xy, co, float(px), float(py))
    T, best = 1.0, np.inf
    for i in range(len(ids)):
        if power[i] > 0.0 or alpha[i] < ALPHA_MIN:
            continue
        test_T = T * (1.0 - alpha[i])
        if test_T < 1e-4:
            break
        best = min(best, abs(T / 0.5 - 1.0), abs(test_T / 0.5 - 1.0))
        T = test_T
    return best


def threshold_margins(st, tile, px, py):
    """Smallest relative distance of an operand to its threshold along the default walk of the pixel, per kind of decision
    (alpha vs 1/255, T (1 - alpha) vs 1e-4, power vs 0 relative to its terms).  A pixel whose INTEGER state (n_contrib,
    final_T's factor count) differs between two implementations although its values agree to the tolerance -- the flipped
    Gaussian carried a weight alpha * T below it -- must have such an operand within the window of its threshold."""
    ids = tile_list(st, tile)
    xy, co, _rgb, _dep = _records(st, ids)
    power, mag, araw, alpha = _terms(xy, co, float(px), float(py))
    T = 1.0
    best = {"power": np.inf, "alpha": np.inf, "T": np.inf}
    for i in range(len(ids)):
        best["power"] = min(best["power"], abs(power[i]) / mag[i] if mag[i] > 1e-200 else np.inf)
        if power[i] > 0.0:
            continue
        best["alpha"] = min(best["alpha"], abs(araw[i] / ALPHA_MIN - 1.0))
        if alpha[i] < ALPHA_MIN:
            continue
        test_T = T * (1.0 - alpha[i])
        best["T"] = min(best["T"], abs(test_T / 1e-4 - 1.0))
        if test_T < 1e-4:
            # the walk ends here by default; had it gone on (the other branch), the next contributors face the same test
            # with a smaller T: not nearer to the threshold
            break
        T = test_T
    return best


def median_gradient_census(st, W, H, chunk_tiles=128):
    """REPORT (not a pass/fail criterion): how often does the reference's backward route dL/dmedian_depth differently
    from the forward's own median decision?  backward.cu:566-569 re-derives the crossing from a transmittance
    reconstructed by repeated DIVISION (T <- T / (1 - alpha), back to front from final_T) and fires where
    `test_T > 0.5 && T < 0.5`; this library sends the gradient to the Gaussian the FORWARD recorded (median image,
    channel 2).  Evaluated here in fp32 torch on the device for every pixel: counts of pixels where the reconstructed
    test fires exactly once at the forward's Gaussian / at another one / never although the forward has a median /
    more than once."""
    import torch
    dev = st["means2D"].device
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T_tiles = gx * gy
    ranges = st["ranges"].long()
    lens = (ranges[:, 1] - ranges[:, 0])
    pl = st["point_list"].long()
    xy, co = st["means2D"], st["conic_opacity"]
    fT, nc = st["final_T"], st["n_contrib"].long()
    med_id = st["median"][2].long()
    med_has = st["median"][1] > 0                                  # median weight > 0: the forward found a crossing
    ys, xs = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
    out = {"pixels": 0, "same": 0, "other": 0, "never": 0, "multiple": 0, "no_median": 0}
    for t0 in range(0, T_tiles, chunk_tiles):
        tiles = torch.arange(t0, min(T_tiles, t0 + chunk_tiles), device=dev)
        L = int(lens[tiles].max())
        if L == 0:
            continue
        pos = torch.arange(L, device=dev)[None, :]
        valid = pos < lens[tiles][:, None]
        idx = (ranges[tiles, 0][:, None] + pos).clamp_(max=pl.numel() - 1)
        ids = pl[idx]                                              # [t, L]
        px = ((tiles % gx) * 16)[:, None, None] + xs[None]         # [t,16,16]
        py = ((tiles // gx) * 16)[:, None, None] + ys[None]
        inside = (px < W) & (py < H)
        pxc, pyc = px.clamp(max=W - 1), py.clamp(max=H - 1)
        dx = xy[ids, 0][:, None, None, :] - pxc[..., None].float()
        dy = xy[ids, 1][:, None, None, :] - pyc[..., None].float()
        c = co[ids]
        power = -0.5 * (c[..., 0][:, None, None] * dx * dx + c[..., 2][:, None, None] * dy * dy) - c[..., 1][:, None, None] * dx * dy
        alpha = torch.clamp_max(c[..., 3][:, None, None] * torch.exp(power), 0.99)
        live = (power <= 0) & (alpha >= 1.0 / 255.0) & valid[:, None, None, :] & (pos[None, None] < nc[pyc, pxc][..., None])
        T = fT[pyc, pxc].clone()
        fires = torch.zeros_like(T, dtype=torch.long)
        fired_id = torch.full_like(fires, -1)
        for k in range(L - 1, -1, -1):
            lv = live[..., k]
            test_T = T / (1.0 - alpha[..., k])
            ev = lv & (test_T > 0.5) & (T < 0.5)
            fires += ev.long()
            fired_id = torch.where(ev, ids[:, k][:, None, None].expand_as(fired_id), fired_id)
            T = torch.where(lv, test_T, T)
        mh = med_has[pyc, pxc] & inside
        mid = med_id[pyc, pxc]
        out["pixels"] += int(inside.sum())
        out["no_median"] += int((inside & ~mh & (fires == 0)).sum())
        out["same"] += int((mh & (fires == 1) & (fired_id == mid)).sum())
        out["other"] += int((inside & (fires == 1) & ((fired_id != mid) | ~mh)).sum())
        out["never"] += int((mh & (fires == 0)).sum())
        out["multiple"] += int((inside & (fires > 1)).sum())
    return out
