#!/bin/bash
# A/B of kernel builds in ONE gpurun call: every gpurun_variants/libgsrast_<tag>.so (built in the dev container with
# make BWD_EXTRA=... / FWD_EXTRA=...; git-ignored, shipped with the snapshot) is copied over gaustudio_amd/libgsrast.so in the
# GPU box's scratch copy of the tree and timed with the same bench command.
# usage: gpu_variants.sh <name> ["workloads"] ["extra bench args"] [tags...]
name="${1:-var}"; wls="${2:-C3}"; extra="${3:-}"; shift; shift; shift
out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
tags="$@"; [ -z "$tags" ] && tags=$(ls gpurun_variants/libgsrast_*.so | sed 's/.*libgsrast_\(.*\)\.so/\1/')
for rep in 1 2; do
for tag in $tags; do
  cp "gpurun_variants/libgsrast_$tag.so" gaustudio_amd/libgsrast.so
  for w in $wls; do
    timeout 300 python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-ref-ab --no-extras $extra > "$out/${tag}_${w}_$rep.json" 2> "$out/${tag}_${w}_$rep.err"
    python - "$out/${tag}_${w}_$rep.json" "$tag" "$w" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f, b = d["stage_ms"]["forward"], d["stage_ms"]["backward"] or {}
    print("%-14s %-5s %8.1f Mpix/s %.4f ms | fwd pre %.4f scan %.4f scat %.4f sort %.4f comp %.4f | bwd comp %.4f pre %.4f" % (
        sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], f["preprocess"], f["scan"], f["scatter"], f["sort"], f["composite"],
        b.get("composite_bwd", 0), b.get("preprocess_bwd", 0)))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "unreadable", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
  done
done
done
