cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_iter -- python3 $GRAFT_REPO_ROOT/scripts/mw_iter_profile.py ce_8_15 3 > $GRAFT_REPO_ROOT/gpurun_out/r3_iter.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/r3_iter.log
t=$(find gpurun_out/prof_iter -name '*kernel_trace.csv' | head -1); python scripts/iter_timeline.py $t > gpurun_out/r3_iter_timeline_c.txt; cat gpurun_out/r3_iter_timeline_c.txt
rm -rf gpurun_out/prof_iter
