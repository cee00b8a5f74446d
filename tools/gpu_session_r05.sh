#!/bin/bash
# Round-5 GPU session: smoke, the -m gpu tests, bench lines (C3 default incl. variants.loss_color, C4, C4-inside, C5, C2-clustered).
# usage (through gpurun): bash tools/gpu_session_r05.sh <name> [tests|bench|all] [extra pytest args]
name="${1:-s}"; what="${2:-all}"; shift 2 || true
out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke" | tee "$out/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$out/summary.txt"; tail -2 "$out/smoke.log" | tee -a "$out/summary.txt"
if [ "$what" = "tests" ] || [ "$what" = "all" ]; then
  echo "== pytest -m gpu" | tee -a "$out/summary.txt"
  GSR_DUMP_PARITY=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 "$@" > "$out/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$out/summary.txt"
  tail -25 "$out/pytest.log" | cut -c1-300 | tee -a "$out/summary.txt"
fi
if [ "$what" = "bench" ] || [ "$what" = "all" ]; then
  echo "== bench" | tee -a "$out/summary.txt"
  timeout 600 python bench.py > "$out/bench_c3.json" 2> "$out/bench_c3.err"; echo "bench C3 rc=$?" | tee -a "$out/summary.txt"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > "$out/bench_c3_driver_flags.json" 2> "$out/bench_c3_driver_flags.err"
  for w in C4 C4-inside C5 C2-clustered C3D0; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab > "$out/bench_$w.json" 2> "$out/bench_$w.err"; echo "bench $w rc=$?" | tee -a "$out/summary.txt"
  done
  timeout 300 python bench.py --workload C4-inside --loss color --steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab > "$out/bench_C4-inside_color.json" 2> "$out/bench_C4-inside_color.err"
  python - "$out" <<'PY' | tee -a "$out/summary.txt"
import json, sys, glob, os
r4 = lambda d: {k: round(v, 4) for k, v in (d or {}).items() if isinstance(v, float)}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], "Mpix/s", d["ms_per_step"], "ms fwd", r4(d["stage_ms"]["forward"]), "bwd", r4(d["stage_ms"]["backward"]),
              "R", d["config"]["num_rendered"], d["config"]["instances_binned"], "staged", d["config"].get("instances_staged_by_composite_fwd"), "vis", d["config"].get("visible"),
              "roof", (d.get("roofline") or {}).get("frac"), "ref_ms", d.get("reference_hipified_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
              "variants", {k: v.get("ms_per_step") for k, v in (d.get("variants") or {}).items()}, "rot", (d.get("rotating") or {}).get("ms_per_step"))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
fi
