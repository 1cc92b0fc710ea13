# Iteration timeline of one named instance: rocprofv3 kernel trace of scripts/mw_iter_profile.py, reduced by scripts/iter_timeline.py
#   gpurun -- bash scripts/iter_timeline.sh ns_8_15_2    -> gpurun_out/timeline_h_<name>.txt
cd /tmp && export TMPDIR=/tmp
N=${1:-polyopt40}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_h -- python3 $GRAFT_REPO_ROOT/scripts/mw_iter_profile.py $N 2 > $GRAFT_REPO_ROOT/gpurun_out/trace_h.log 2>&1; echo "rocprof rc=$?"
t=$(find $GRAFT_REPO_ROOT/gpurun_out/trace_h -name '*kernel_trace.csv' | head -1); python $GRAFT_REPO_ROOT/scripts/iter_timeline.py $t > $GRAFT_REPO_ROOT/gpurun_out/timeline_h_$N.txt; tail -48 $GRAFT_REPO_ROOT/gpurun_out/timeline_h_$N.txt
find $GRAFT_REPO_ROOT/gpurun_out/trace_h -name '*kernel_trace.csv' -delete
