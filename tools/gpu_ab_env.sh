#!/bin/bash
# A/B of an environment switch in one gpurun call.  usage: gpu_ab_env.sh <name> <ENVVAR> "<values>" "<workloads>" ["pytest -k expr"|none]
name="$1"; var="$2"; vals="$3"; wls="$4"; kexpr="${5:-none}"
out="gpurun_out/$name"; mkdir -p "$out"; export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$kexpr" != "none" ]; then timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$kexpr" 2>&1 | tail -3; fi
for rep in 1 2; do for v in $vals; do for w in $wls; do
  env $var=$v timeout 300 python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-ref-ab --no-extras > "$out/${var}${v}_${w}_$rep.json" 2> "$out/${var}${v}_${w}_$rep.err"
  python - "$out/${var}${v}_${w}_$rep.json" "$var=$v" "$w" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    f, b = d["stage_ms"]["forward"], d["stage_ms"]["backward"] or {}
    print("AB %-14s %-10s %.4f ms | pre %.4f scan %.4f scat %.4f sort %.4f comp %.4f | bwd comp %.4f pre %.4f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], f["preprocess"], f["scan"], f["scatter"], f["sort"], f["composite"], b.get("composite_bwd", 0), b.get("preprocess_bwd", 0)))
except Exception as e:
    print("AB", sys.argv[2], sys.argv[3], "unreadable", e)
PY
done; done; done
