#!/usr/bin/env python
"""The alpha-beta model behind DESIGN.md s7: exposed exchange time and 8-GPU speed-up of the dense and the factored
gradient exchange (gaustudio_amd/parallel.py) for a given compute step.  Nothing here is measured on multi-GPU hardware.

    python tools/comm_model.py                      # the table of DESIGN.md s7 (C3, C4)
    python tools/comm_model.py --P 1e6 --t_c 1.115 --B 330 --alpha 25 --N 8 --V 1 --visible 0.85

B: bus bandwidth RCCL reaches on large messages on an 8 x MI355X node (GB/s; xGMI: 7 links x ~64 GB/s per direction),
alpha: latency per collective (us), t_c: compute step per view (ms), V: views per rank, visible: fraction of the Gaussians
with a non-zero gradient row in some view of the step (compaction)."""
import argparse


def allreduce_ms(S_bytes, N, B, alpha_us):
    return alpha_us * 1e-3 + 2.0 * (N - 1) / N * S_bytes / (B * 1e9) * 1e3


def allgather_ms(S_rank_bytes, N, B, alpha_us):
    return alpha_us * 1e-3 + (N - 1) * S_rank_bytes / (B * 1e9) * 1e3


def model(P, t_c, N, V, B, alpha, M=16, visible=1.0, chunk_overlap_ms=0.0):
    dense_bytes = P * (M * 3 + 11) * 4
    t_dense = max(0.0, allreduce_ms(dense_bytes, N, B, alpha) - chunk_overlap_ms)
    rows = P * visible
    t_fact = allreduce_ms(rows * 44, N, B, alpha) + allgather_ms(rows * 12 * V, N, B, alpha)
    comp = t_c * V
    return {"dense": dict(bytes_moved=2.0 * (N - 1) / N * dense_bytes, exposed_ms=t_dense, speedup=N * comp / (comp + t_dense)),
            "factored": dict(bytes_moved=2.0 * (N - 1) / N * rows * 44 + (N - 1) * rows * 12 * V, exposed_ms=t_fact,
                             speedup=N * comp / (comp + t_fact))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=float, default=None)
    ap.add_argument("--t_c", type=float, default=1.115)
    ap.add_argument("--B", type=float, default=330.0)
    ap.add_argument("--alpha", type=float, default=25.0)
    ap.add_argument("--N", type=int, default=8)
    ap.add_argument("--V", type=int, default=1)
    ap.add_argument("--visible", type=float, default=1.0)
    a = ap.parse_args()
    if a.P is not None:
        for k, v in model(int(a.P), a.t_c, a.N, a.V, a.B, a.alpha, visible=a.visible).items():
            print(f"{k:9s} {v['bytes_moved'] / 1e6:8.0f} MB moved per rank  exchange {v['exposed_ms']:.2f} ms  speed-up at {a.N} GPUs {v['speedup']:.2f}x")
        return
    print(f"B = {a.B} GB/s, alpha = {a.alpha} us; nothing assumed hidden except the SH rebuild kernel")
    for name, P, t_c in (("C3 (1 M, step 1.115 ms)", 1_000_000, 1.115), ("C4 share (5 M, step 2.13 ms)", 5_000_000, 2.13)):
        print(name)
        for N in (2, 4, 8):
            for V, vis in ((1, 1.0), (1, 0.85), (2, 1.0), (4, 1.0)):
                m = model(P, t_c, N, V, a.B, a.alpha, visible=vis)
                print(f"  N={N} V={V} visible={vis:4.2f}:  dense {m['dense']['bytes_moved'] / 1e6:6.0f} MB {m['dense']['exposed_ms']:.2f} ms {m['dense']['speedup']:.2f}x"
                      f"   factored {m['factored']['bytes_moved'] / 1e6:6.0f} MB {m['factored']['exposed_ms']:.2f} ms {m['factored']['speedup']:.2f}x")


if __name__ == "__main__":
    main()
