# rocprofv3 kernel stats of the banded factored step WITHOUT a process group (diagnostic)
cd /tmp && export TMPDIR=/tmp
BAND_BACKEND=none timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b2 -- python $GRAFT_REPO_ROOT/tools/band_exchange_timing.py 2 2 > /tmp/prof_b2.log 2>&1
echo "rc=$?"; grep "bands=" /tmp/prof_b2.log | tail -3
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_b2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg_us", "%.1f"%(float(r["AverageNs"])/1e3), "max_us", "%.1f"%(float(r["MaxNs"])/1e3), "tot_ms", "%.2f"%(float(r["TotalDurationNs"])/1e6))
PY
