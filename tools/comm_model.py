#!/usr/bin/env python
"""The alpha-beta model behind DESIGN.md s7: exposed exchange time and N-GPU speed-up of the gradient exchanges of
gaustudio_amd/parallel.py for a given compute step.  NOTHING HERE IS MEASURED ON MULTI-GPU HARDWARE: no 8-GPU node has been
available to the builder (or, rounds 1-3, to the driver); the single-GPU inputs (step times, stage times, visible
fractions) are measured, the interconnect is a model.

    python tools/comm_model.py                      # the tables of DESIGN.md s7
    python tools/comm_model.py --P 1e6 --t_c 1.06 --tail 0.05 --B 330 --alpha 25 --N 8 --V 1 --vis_view 0.85 --vis_union 0.87

Interconnect.  8 x MI355X, fully connected xGMI mesh: 7 links per GPU, ~64 GB/s per direction each on the previous
generation (448 GB/s out per GPU), more on this one (the task sheet quotes ~153 GB/s per link, i.e. ~77 per direction,
537 GB/s out).  RCCL's large-message all-reduce reaches ~70-75 % of that as bus bandwidth; B = 330 GB/s is the
conservative default, B = 400 the other column.  alpha = latency of one collective (25 us).

  all-reduce of S bytes:           alpha + 2 (N-1)/N S / B
  all-gather of S bytes per rank:  alpha + (N-1) S / B

Exchanges (bytes per Gaussian and rank; M = 16 SH coefficients, V views per rank):
  dense                one all-reduce of 236 B                                        (north star's "single all-reduce")
  factored             all-gather 12 V B (dRGB per view) + all-reduce 44 B (geometry)  (round 3)
  factored, early      the same, each view's all-gather started from inside its backward: at V = 1 the SH-direction stage
                       (`tail` ms) overlaps it; at V > 1 additionally the (V-1) later views' compute
  view                 all-gather of header (P/8 + P/64 B) + 12 B x vis_view per view; geometry all-reduce dense
  view+geometry        as view, geometry all-reduce on the union rows (44 B x vis_union) + ~30 us of host synchronisation

Collectives on one communicator run one after the other; what is not hidden behind compute is exposed.  speed-up =
N V t_c / (V t_c + exposed)."""
import argparse


def allreduce_ms(S, N, B, alpha_us):
    return alpha_us * 1e-3 + 2.0 * (N - 1) / N * S / (B * 1e9) * 1e3


def allgather_ms(S_rank, N, B, alpha_us):
    return alpha_us * 1e-3 + (N - 1) * S_rank / (B * 1e9) * 1e3


def model(P, t_c, N, V, B, alpha, M=16, vis_view=1.0, vis_union=1.0, tail=0.05, sync_ms=0.03):
    """-> {exchange: (bytes moved per rank, exposed ms, speed-up)}.  t_c: compute per view (ms); tail: the part of a backward that
    follows its geometry stage (SH-direction kernel), which an early all-gather overlaps."""
    comp = t_c * V
    out = {}

    def put(name, moved, exposed):
        out[name] = (moved, exposed, N * comp / (comp + exposed))

    dense = P * (M * 3 + 11) * 4
    put("dense", 2.0 * (N - 1) / N * dense, allreduce_ms(dense, N, B, alpha))
    geo = allreduce_ms(P * 44, N, B, alpha)
    ag_view = allgather_ms(P * 12, N, B, alpha)                       # one view's colour slots
    put("factored", 2.0 * (N - 1) / N * P * 44 + (N - 1) * P * 12 * V, geo + allgather_ms(P * 12 * V, N, B, alpha))
    # early: view v's all-gather starts `tail` before its backward ends; the views after it keep computing
    hidden = lambda ag: sum(min(ag, tail + (V - 1 - v) * t_c) for v in range(V))
    put("factored, early", out["factored"][0], geo + V * ag_view - hidden(ag_view))
    hdr = P / 8.0 + P / 64.0 + 16
    ag_pack = allgather_ms(hdr + P * 12 * vis_view, N, B, alpha)
    put("view", 2.0 * (N - 1) / N * P * 44 + (N - 1) * V * (hdr + P * 12 * vis_view), geo + V * ag_pack - hidden(ag_pack))
    geo_u = allreduce_ms(P * 44 * vis_union, N, B, alpha) + sync_ms
    put("view+geometry", 2.0 * (N - 1) / N * P * 44 * vis_union + (N - 1) * V * (hdr + P * 12 * vis_view), geo_u + V * ag_pack - hidden(ag_pack))
    return out


def table(title, P, t_c, tail, cases, B_list=(330.0, 400.0), alpha=25.0, N=8):
    print(title)
    print(f"  {'V':>2} {'vis/view':>8} {'union':>6}   " + "   ".join(f"{'B=%d' % B:^58}" for B in B_list))
    names = ("dense", "factored", "factored, early", "view", "view+geometry")
    for V, vv, vu in cases:
        cols = []
        for B in B_list:
            m = model(P, t_c, N, V, B, alpha, vis_view=vv, vis_union=vu, tail=tail)
            cols.append(" ".join(f"{m[k][2]:5.2f}x" for k in names) + f"  [{m['view'][0] / 1e6:5.0f} MB, {m['view'][1]:.2f} ms]")
        print(f"  {V:>2} {vv:>8.2f} {vu:>6.2f}   " + "   ".join(cols))
    print("     columns per B: dense | factored | factored, early | view | view+geometry   [view: MB moved per rank, exposed ms]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=float, default=None)
    ap.add_argument("--t_c", type=float, default=1.06)
    ap.add_argument("--tail", type=float, default=0.05)
    ap.add_argument("--B", type=float, default=330.0)
    ap.add_argument("--alpha", type=float, default=25.0)
    ap.add_argument("--N", type=int, default=8)
    ap.add_argument("--V", type=int, default=1)
    ap.add_argument("--vis_view", type=float, default=1.0)
    ap.add_argument("--vis_union", type=float, default=1.0)
    a = ap.parse_args()
    if a.P is not None:
        for k, (moved, exposed, sp) in model(int(a.P), a.t_c, a.N, a.V, a.B, a.alpha, vis_view=a.vis_view, vis_union=a.vis_union, tail=a.tail).items():
            print(f"{k:16s} {moved / 1e6:8.0f} MB moved per rank  exposed {exposed:.2f} ms  speed-up at {a.N} GPUs {sp:.2f}x")
        return
    print("alpha = 25 us per collective.  MODEL, not a measurement (no multi-GPU node available); single-GPU inputs are measured:")
    print("C3 step 1.06 ms (BENCH r04 builder runs, fast_exp default), SH-direction tail 0.05 ms; C4 share step 2.04 ms, tail 0.25 ms;")
    print("visible fractions: bench scenes (synthetic frustum cloud: every rank's view sees 0.85, the union of 8 views 3 degrees apart 0.87);")
    print("ring cameras OUTSIDE a ball of Gaussians: 0.81 per view / 0.94 union (camera radius 5, ball 3), 0.62 / 0.87 (radius 4);")
    print("cameras INSIDE the scene (a 360-degree capture: camera ring radius 2-3 in a ball of radius 6): 0.14-0.19 per view / 0.52-0.59 union")
    print("(oracle radii > 0, 200 k Gaussians, 8 ring cameras 1297x840: the numbers in tools/comm_model.py's docstring of DESIGN.md s7).\n")
    table("C3 (1 M Gaussians), N = 8", 1_000_000, 1.06, 0.05, [(1, 1.0, 1.0), (1, 0.85, 0.87), (2, 0.85, 0.90), (4, 0.85, 0.95)])
    print()
    table("C4 share (5 M Gaussians), N = 8", 5_000_000, 2.04, 0.25,
          [(1, 1.0, 1.0), (1, 0.81, 0.94), (1, 0.62, 0.87), (1, 0.19, 0.59), (1, 0.14, 0.52), (2, 0.19, 0.75)])
    print()
    for N in (2, 4):
        table(f"C3, N = {N}", 1_000_000, 1.06, 0.05, [(1, 0.85, 0.87)], N=N)


if __name__ == "__main__":
    main()
