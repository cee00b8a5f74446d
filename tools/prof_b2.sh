# rocprofv3 kernel trace of the factored step WITHOUT a process group (diagnostic): BAND_COMPACT=none|view, bands from $1
cd /tmp && export TMPDIR=/tmp
BAND_BACKEND=none timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fx -o k -- python $GRAFT_REPO_ROOT/tools/band_exchange_timing.py ${1:-1} 2 > /tmp/prof_fx.log 2>&1
echo "rc=$?"; grep "bands=" /tmp/prof_fx.log | tail -3
cd $GRAFT_REPO_ROOT
db=$(ls gpurun_out/prof_fx/*/k_results.db gpurun_out/prof_fx/k_results.db 2>/dev/null | head -1)
python tools/rocpd_summary.py "$db" --sequence 40 | cut -c1-150
