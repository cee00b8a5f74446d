#!/bin/bash
# (one-off of round 5; GSR_SH_PREFETCH was the A/B switch of the estimate-gated SH request, since replaced by the late-only request and removed)
# Round-5 session b: the SH-request gating A/B and the world-1 (FORCE_PG) lines of the N > 1 code path that feed tools/comm_model.py.
name="${1:-r05b}"; out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "gating or staged or c4_inside or fused or baseline_configs_bit_exact" > "$out/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$out/summary.txt"; tail -4 "$out/pytest.log" | cut -c1-300 | tee -a "$out/summary.txt"
B="--steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab --no-extras"
for w in C4-inside C4 C3; do
  for g in 0 1; do
    GSR_SH_PREFETCH=$g timeout 300 python bench.py --workload $w $B > "$out/bench_${w}_gate$g.json" 2> "$out/bench_${w}_gate$g.err"
  done
done
# the N > 1 code path on one GPU (process group of one rank over RCCL): per-rank compute of the factored / dense step
for w in C3 C4 C4-inside; do
  for ex in "factored none" "factored view" "dense none"; do
    set -- $ex
    GSR_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --gpus 1 --workload $w $B --exchange $1 --compact $2 > "$out/pg1_${w}_$1_$2.json" 2> "$out/pg1_${w}_$1_$2.err"
  done
done
python - "$out" <<'PY' | tee -a "$out/summary.txt"
import json, sys, glob, os
r4 = lambda d: {k: round(v, 4) for k, v in (d or {}).items() if isinstance(v, float)}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d.get("comm") or {}
        print(os.path.basename(f), d["ms_per_step"], "ms fwd", r4(d["stage_ms"]["forward"]), "bwd", r4(d["stage_ms"]["backward"]), "vis", d["config"].get("visible"),
              "comm", {k: c.get(k) for k in ("exchange", "compute_ms", "comm_exposed_ms", "payload_bytes_per_rank", "color_rows_per_view", "exchange_fallback")} if c else None)
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
