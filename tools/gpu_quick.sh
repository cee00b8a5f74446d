#!/bin/bash
# Quick GPU iteration: selected parity tests + C3/C4 bench lines (no CPU baseline) under optional env variants.
# usage: gpu_quick.sh <name> [pytest -k expr] [variants: space-separated "TAG:ENV=V,ENV2=V2" entries]
name="${1:-q}"; kexpr="${2:-backward or fuzz or baseline or caller}"; variants="${3:-default:}"
out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$kexpr" != "none" ]; then
  GSR_DUMP_PARITY=1 timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$kexpr" > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$out/pytest.log"
fi
for v in $variants; do
  tag="${v%%:*}"; envs="${v#*:}"
  for w in ${GSR_QUICK_WORKLOADS:-C3 C4}; do
    ( IFS=','; for e in $envs; do [ -n "$e" ] && export "$e"; done
      timeout 300 python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-ref-ab > "$out/bench_${w}_$tag.json" 2> "$out/bench_${w}_$tag.err" )
    python - "$out/bench_${w}_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["config"]["workload"][:8], d["value"], "Mpix/s", d["ms_per_step"], "ms | fwd", {k: round(v, 4) for k, v in d["stage_ms"]["forward"].items()}, "| bwd", {k: round(v, 4) for k, v in d["stage_ms"]["backward"].items()})
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
  done
done
