#!/bin/bash
# Round-6 GPU session (one gpurun call): smoke, the -m gpu tests (parity report dumped), the tests of the per-wave A/B kernels on an
# AB=1 build (gpurun_variants/libgsrast_ab.so, built in the dev container: tools/build_variant.sh ab "" "" AB=1), the bench lines, the
# fuzz campaigns.   usage: bash tools/gpu_session_r06.sh <name> [tests|bench|fuzz|all]
name="${1:-s}"; what="${2:-all}"
out="gpurun_out/$name"; mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke" | tee "$out/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$out/summary.txt"; tail -2 "$out/smoke.log" | tee -a "$out/summary.txt"
if [ "$what" = "tests" ] || [ "$what" = "all" ]; then
  echo "== pytest -m gpu" | tee -a "$out/summary.txt"
  rm -f gpurun_out/r06_parity.json
  GSR_DUMP_PARITY=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 -rs > "$out/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$out/summary.txt"
  tail -30 "$out/pytest.log" | cut -c1-300 | tee -a "$out/summary.txt"
  cp gpurun_out/r06_parity.json "$out/" 2>/dev/null
  if [ -f gpurun_variants/libgsrast_ab.so ]; then
    echo "== the per-wave A/B kernels (AB=1 build swapped in)" | tee -a "$out/summary.txt"
    cp gaustudio_amd/libgsrast.so /tmp/libgsrast.shipped.so; cp gpurun_variants/libgsrast_ab.so gaustudio_amd/libgsrast.so
    timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fastexp.py tests/test_gpu_forward.py -m gpu -q -p no:cacheprovider -rs -k "variants or ab_variants or cull or null_upstream or mode" > "$out/pytest_ab.log" 2>&1; echo "pytest (AB build) rc=$?" | tee -a "$out/summary.txt"
    tail -6 "$out/pytest_ab.log" | cut -c1-300 | tee -a "$out/summary.txt"
    cp /tmp/libgsrast.shipped.so gaustudio_amd/libgsrast.so
  fi
fi
if [ "$what" = "bench" ] || [ "$what" = "all" ]; then
  echo "== bench" | tee -a "$out/summary.txt"
  timeout 900 python bench.py > "$out/bench_c3.json" 2> "$out/bench_c3.err"; echo "bench C3 (default command) rc=$?" | tee -a "$out/summary.txt"
  timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench_c3_driver_flags.json" 2> "$out/bench_c3_driver_flags.err"
  for w in C4 C4-inside C5 C2-clustered C3D0 C2 C1; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab > "$out/bench_$w.json" 2> "$out/bench_$w.err"; echo "bench $w rc=$?" | tee -a "$out/summary.txt"
  done
  timeout 300 python bench.py --workload C3-extract --steps 20 --warmup 5 > "$out/bench_C3-extract.json" 2> "$out/bench_C3-extract.err"
  timeout 300 python bench.py --workload C4-inside --loss color --steps 20 --warmup 10 --no-cpu-baseline --no-ref-ab > "$out/bench_C4-inside_color.json" 2> "$out/bench_C4-inside_color.err"
  python - "$out" <<'PY' | tee -a "$out/summary.txt"
import json, sys, glob, os
r4 = lambda d: {k: round(v, 4) for k, v in (d or {}).items() if isinstance(v, float)}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], "Mpix/s", d["ms_per_step"], "ms fwd", r4((d.get("stage_ms") or {}).get("forward")), "bwd", r4((d.get("stage_ms") or {}).get("backward")),
              "roof", (d.get("roofline") or {}).get("frac"), "traffic ok", (d.get("roofline") or {}).get("traffic_counters_match_kernel_sources"), "ref_ms", d.get("reference_hipified_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
              "variants", {k: v.get("ms_per_step") for k, v in (d.get("variants") or {}).items()}, "rot", (d.get("rotating") or {}).get("ms_per_step"),
              "others", {k: v.get("ms_per_step") for k, v in (d.get("other_workloads") or {}).items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
fi
if [ "$what" = "dc" ] || [ "$what" = "all" ]; then
  # rocprofv3 kernel stats of the DEFAULT command's timed loop (settle + warm-up + timed + stage steps at warm clocks): its average
  # for composite_fwd must agree with the live HIP-event figure of the same run (`roofline.avg_ms`)
  echo "== rocprofv3 --kernel-trace --stats -- python bench.py --no-extras" | tee -a "$out/summary.txt"
  root="$(pwd)"
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$root/$out/dc" -o k -- python "$root/bench.py" --no-extras --no-cpu-baseline --no-ref-ab > "$root/$out/dc.log" 2>&1 )
  grep '^{"metric"' "$out/dc.log" | tail -1 > "$out/c3_default_command_bench_line.json"
  db=$(ls "$out"/dc/*/k_results.db "$out"/dc/k_results.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py "$db" > "$out/c3_default_command_kernel_stats.txt" 2>&1
  rm -rf "$out/dc"
  head -12 "$out/c3_default_command_kernel_stats.txt" | cut -c1-170 | tee -a "$out/summary.txt"
  python -c "
import json; j=json.load(open('$out/c3_default_command_bench_line.json')); print('live composite avg_ms', j['roofline']['avg_ms'], 'frac', j['roofline']['frac'], 'step', j['ms_per_step'])" | tee -a "$out/summary.txt"
fi
if [ "$what" = "fuzz" ] || [ "$what" = "all" ]; then
  echo "== fuzz" | tee -a "$out/summary.txt"
  timeout 1500 python tests/tools/fuzz_campaign.py --n 300 --first 2600 > "$out/fuzz_campaign.log" 2>&1; tail -3 "$out/fuzz_campaign.log" | tee -a "$out/summary.txt"
  timeout 600 python tests/tools/fuzz_sort.py --n 60 --first 560 > "$out/fuzz_sort.log" 2>&1; tail -2 "$out/fuzz_sort.log" | tee -a "$out/summary.txt"
  timeout 900 python tests/tools/fuzz_fastexp.py --n 60 --first 700 > "$out/fuzz_fastexp.log" 2>&1; tail -2 "$out/fuzz_fastexp.log" | tee -a "$out/summary.txt"
fi
