#!/bin/bash
# quick A/B session: selected tests + bench lines for the named workloads.  usage: gpu_quick2.sh <name> "<pytest -k expr|none>" "<workloads>" [bench extra args]
name="${1:-q}"; kexpr="${2:-backward}"; wls="${3:-C3 C4-inside}"; extra="${4:-}"
out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$kexpr" != "none" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$kexpr" > "$out/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$out/summary.txt"; tail -6 "$out/pytest.log" | cut -c1-300 | tee -a "$out/summary.txt"
fi
for w in $wls; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab --no-extras $extra > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
python - "$out" <<'PY' | tee -a "$out/summary.txt"
import json, sys, glob, os
r4 = lambda d: {k: round(v, 4) for k, v in (d or {}).items() if isinstance(v, float)}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), d["ms_per_step"], "ms fwd", r4(d["stage_ms"]["forward"]), "bwd", r4(d["stage_ms"]["backward"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
