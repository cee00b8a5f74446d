out=gpurun_out/r05_pg1b; mkdir -p $out
B="--steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab --no-extras"
for w in C3 C4 C4-inside; do
  for ex in "factored view+geometry" "factored view" "factored none"; do
    set -- $ex
    GSR_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --gpus 1 --workload $w $B --exchange $1 --compact $2 > "$out/pg1_${w}_$1_$2.json" 2> "$out/pg1_${w}_$1_$2.err"
  done
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); c = d["comm"]
        print(os.path.basename(f), d["ms_per_step"], {k: c.get(k) for k in ("compute_ms", "comm_exposed_ms", "payload_bytes_per_rank", "geometry_rows", "color_rows_per_view", "geometry_fallbacks")})
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
