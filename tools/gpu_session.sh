#!/bin/bash
# One gpurun call = one session on the MI355X box: smoke, the -m gpu tests, bench lines.  Everything is bounded by
# `timeout` (a hung kernel must not take the box down with it) and logs go to gpurun_out/<name>/.
# usage (from the repo root, through gpurun): bash tools/gpu_session.sh <name> [tests|bench|all] [extra pytest args]
name="${1:-s}"; what="${2:-all}"; shift 2 || true
out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke" | tee "$out/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$out/summary.txt"; tail -3 "$out/smoke.log" | tee -a "$out/summary.txt"
if [ "$what" = "tests" ] || [ "$what" = "all" ]; then
  echo "== pytest -m gpu" | tee -a "$out/summary.txt"
  GSR_DUMP_PARITY=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 "$@" > "$out/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$out/summary.txt"
  tail -40 "$out/pytest.log" | tee -a "$out/summary.txt"
fi
if [ "$what" = "bench" ] || [ "$what" = "all" ]; then
  echo "== bench" | tee -a "$out/summary.txt"
  timeout 600 python bench.py > "$out/bench_c3.json" 2> "$out/bench_c3.err"; echo "bench C3 rc=$?" | tee -a "$out/summary.txt"
  for w in C3D0 C2 C4 C5; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab > "$out/bench_$w.json" 2> "$out/bench_$w.err"; echo "bench $w rc=$?" | tee -a "$out/summary.txt"
  done
  timeout 300 python bench.py --fwd-only --steps 30 --warmup 10 --no-cpu-baseline > "$out/bench_c3_fwd.json" 2> "$out/bench_c3_fwd.err"
  python - "$out" <<'PY' | tee -a "$out/summary.txt"
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], "Mpix/s", d["ms_per_step"], "ms", "fwd", d["stage_ms"]["forward"], "bwd", d["stage_ms"]["backward"],
              "R", d["config"]["num_rendered"], d["config"]["instances_binned"], "roof", (d.get("roofline") or {}).get("frac"), "ref_ms", d.get("reference_hipified_ms"),
              "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
fi
