#!/bin/bash
# Quick GPU iteration: backward/forward parity tests + one C3 bench line (no CPU baseline).  usage: gpu_quick.sh <name> [pytest -k expr]
name="${1:-q}"; kexpr="${2:-backward or fuzz or baseline or caller}"
out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$kexpr" > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$out/pytest.log"
for w in C3 C4; do
timeout 300 python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-ref-ab > "$out/bench_$w.json" 2> "$out/bench_$w.err"
python - "$out/bench_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:8], d["value"], "Mpix/s", d["ms_per_step"], "ms | fwd", {k: round(v, 4) for k, v in d["stage_ms"]["forward"].items()}, "| bwd", {k: round(v, 4) for k, v in d["stage_ms"]["backward"].items()})
PY
done
