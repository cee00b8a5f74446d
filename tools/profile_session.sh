#!/bin/bash
# One gpurun call: rocprofv3 kernel-trace stats + PMC passes (SQ / FETCH / WRITE, each in its own run, never combined
# with tracing domains other than --kernel-trace) of bench.py on a workload; summaries land in gpurun_out/<name>/.
# usage: bash tools/profile_session.sh <name> [workload=C3] [passes="stats sq fetch write"]
name="${1:-prof}"; wl="${2:-C3}"; passes="${3:-stats sq fetch write}"; extra="${4:-}"   # extra: e.g. "--rotate-cameras 8"
out="gpurun_out/$name"; mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
root="$(pwd)"
bench="python $root/bench.py --workload $wl --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-ref-ab --no-extras $extra"   # (--settle 0: counters, not clocks)
cd /tmp
for p in $passes; do
  case $p in
    stats) timeout 600 rocprofv3 --kernel-trace --stats -d "$root/$out/stats" -o k -- $bench > "$root/$out/stats.log" 2>&1
           grep '^{"metric"' "$root/$out/stats.log" | tail -1 > "$root/$out/${wl}_bench_line.json"
           db=$(ls "$root/$out"/stats/*/k_results.db "$root/$out"/stats/k_results.db 2>/dev/null | head -1)
           python "$root/tools/rocpd_summary.py" "$db" > "$root/$out/${wl}_kernel_stats.txt" 2>&1 ;;
    sq)    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d "$root/$out/sq" -o k -- $bench > "$root/$out/sq.log" 2>&1
           db=$(ls "$root/$out"/sq/*/k_results.db "$root/$out"/sq/k_results.db 2>/dev/null | head -1)
           python "$root/tools/pmc_dump.py" "$db" > "$root/$out/${wl}_pmc_sq.txt" 2>&1 ;;
    sq2)   timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM -d "$root/$out/sq2" -o k -- $bench > "$root/$out/sq2.log" 2>&1
           db=$(ls "$root/$out"/sq2/*/k_results.db "$root/$out"/sq2/k_results.db 2>/dev/null | head -1)
           python "$root/tools/pmc_dump.py" "$db" > "$root/$out/${wl}_pmc_sq2.txt" 2>&1 ;;
    calib) ( cd "$root" && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$root/$out/cf" -o k -- "$root/tools/fetch_calib.bin" > "$root/$out/calib_fetch.log" 2>&1
             db=$(ls "$root/$out"/cf/*/k_results.db "$root/$out"/cf/k_results.db 2>/dev/null | head -1); python "$root/tools/pmc_dump.py" "$db" > "$root/$out/calib_pmc_fetch.txt" 2>&1
             timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$root/$out/cw" -o k -- "$root/tools/fetch_calib.bin" > "$root/$out/calib_write.log" 2>&1
             db=$(ls "$root/$out"/cw/*/k_results.db "$root/$out"/cw/k_results.db 2>/dev/null | head -1); python "$root/tools/pmc_dump.py" "$db" > "$root/$out/calib_pmc_write.txt" 2>&1
             rm -rf "$root/$out/cf" "$root/$out/cw" ) ;;
    fetch) timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$root/$out/fetch" -o k -- $bench > "$root/$out/fetch.log" 2>&1
           db=$(ls "$root/$out"/fetch/*/k_results.db "$root/$out"/fetch/k_results.db 2>/dev/null | head -1)
           python "$root/tools/pmc_dump.py" "$db" > "$root/$out/${wl}_pmc_fetch.txt" 2>&1 ;;
    write) timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$root/$out/write" -o k -- $bench > "$root/$out/write.log" 2>&1
           db=$(ls "$root/$out"/write/*/k_results.db "$root/$out"/write/k_results.db 2>/dev/null | head -1)
           python "$root/tools/pmc_dump.py" "$db" > "$root/$out/${wl}_pmc_write.txt" 2>&1 ;;
  esac
done
cd "$root"
# the raw databases are large: keep only the summaries
rm -rf "$out"/stats "$out"/sq "$out"/sq2 "$out"/fetch "$out"/write
ls -la "$out"; head -30 "$out/${wl}_kernel_stats.txt"
