#!/bin/bash
# One gpurun call: the bench modes beyond the headline line -- C3-extract, the default line with its rotating / fast_exp
# blocks, the rotating headline, and the N = 2 plumbing of both gradient exchanges on ONE GPU (gloo; RCCL refuses two
# ranks per device -- the measured configuration on the 8-GPU node is nccl).   usage: gpu_bench_modes.sh <name>
name="${1:-modes}"; out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { tag=$1; shift; timeout 600 "$@" > "$out/$tag.json" 2> "$out/$tag.err"; echo "$tag rc=$?"; }
run default python bench.py --steps 30 --warmup 10 --no-ref-ab
run extract python bench.py --workload C3-extract --steps 16 --warmup 8
run rotate python bench.py --rotate-cameras 8 --steps 24 --warmup 8 --no-cpu-baseline --no-ref-ab
run exact python bench.py --exact --steps 30 --warmup 10 --no-cpu-baseline --no-ref-ab --no-extras
for w in C1 C2 C2-clustered C3D0 C4 C5; do run $w python bench.py --workload $w --steps 24 --warmup 8 --no-cpu-baseline --no-ref-ab --no-extras; done
# the N > 1 code path over RCCL on the one GPU (process group of one rank: every collective call of the step meets the library)
for ex in "dense none" "factored none" "factored view" "factored view+geometry"; do set -- $ex
  GSR_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 run nccl1_$1_$2 python bench.py --gpus 1 --steps 12 --warmup 4 --exchange $1 --compact $2 --no-cpu-baseline --no-ref-ab
done
for ex in dense factored; do
  GSR_BENCH_BACKEND=gloo run n2_${ex} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 6 --warmup 3 --workload C2 --exchange $ex
done
GSR_BENCH_BACKEND=gloo run n2_factored_v2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --steps 6 --warmup 3 --workload C2 --exchange factored --views-per-rank 2
GSR_BENCH_BACKEND=gloo run n2_factored_view python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 \
    bench.py --gpus 2 --steps 6 --warmup 3 --workload C2 --exchange factored --compact view
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
        print(os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], "ms", "| stage", d.get("stage_ms", {}).get("pipeline") or {k: (v and {a: round(b, 4) for a, b in v.items()}) for k, v in d["stage_ms"].items()})
        for k in ("rotating", "variants", "comm", "cpu_baseline", "epilogue_roofline"):
            if d.get(k):
                print("   ", k, json.dumps(d[k])[:600])
        if "tsdf" in d.get("config", {}): print("    tsdf", d["config"]["tsdf"], d["config"].get("valid_points_per_frame"))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e, open(f.replace(".json", ".err")).read()[-800:])
PY
