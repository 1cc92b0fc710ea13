cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mw_parity.py -m gpu -x -q -s > gpurun_out/r3_mwparity.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_mwparity.log; grep "backward errors" gpurun_out/r3_mwparity.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_iter -- python3 $GRAFT_REPO_ROOT/scripts/mw_iter_profile.py ce_8_15 3 > $GRAFT_REPO_ROOT/gpurun_out/r3_iter.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; cat gpurun_out/r3_iter.log | tail -4
t=$(find gpurun_out/prof_iter -name '*kernel_trace.csv' | head -1); python scripts/iter_timeline.py $t > gpurun_out/r3_iter_timeline.txt; cat gpurun_out/r3_iter_timeline.txt
head -2 $t
rm -rf gpurun_out/prof_iter
