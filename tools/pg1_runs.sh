mkdir -p gpurun_out/r06_pg1
export GSR_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for W in C3 C4-inside C4; do
  for C in view view+geometry; do
    for B in 1 2; do
      python bench.py --gpus 1 --workload $W --steps 30 --warmup 10 --exchange factored --compact $C --bands $B --no-cpu-baseline --no-ref-ab --no-extras > gpurun_out/r06_pg1/pg1_${W}_${C}_b${B}.json 2> gpurun_out/r06_pg1/pg1_${W}_${C}_b${B}.err
      python - <<PY
import json
try:
    j=json.load(open("gpurun_out/r06_pg1/pg1_${W}_${C}_b${B}.json"))
    c=j.get("comm") or {}
    print("$W $C bands $B: ms_per_step", j["ms_per_step"], "comm_exposed", c.get("comm_exposed_ms"), "compute", c.get("compute_ms"), "payload", c.get("payload_bytes_per_rank"), "rows/view", c.get("color_rows_per_view"), "fallback", c.get("exchange_fallback"))
except Exception as e:
    print("$W $C $B FAILED", e, open("gpurun_out/r06_pg1/pg1_${W}_${C}_b${B}.err").read()[-400:])
PY
    done
  done
done
unset GSR_BENCH_FORCE_PG RANK WORLD_SIZE LOCAL_RANK
python bench.py --workload C3 --steps 30 --warmup 10 --no-cpu-baseline --no-ref-ab --no-extras > gpurun_out/r06_pg1/plain_C3.json
python -c "import json; j=json.load(open('gpurun_out/r06_pg1/plain_C3.json')); print('plain C3', j['ms_per_step'])"
