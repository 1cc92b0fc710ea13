cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_g -- python3 $GRAFT_REPO_ROOT/scripts/mw_roofline.py 5 1024 3 on > $GRAFT_REPO_ROOT/gpurun_out/prof_g.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_g -name '*kernel_stats.csv' | head -1); head -12 $f
find $GRAFT_REPO_ROOT/gpurun_out/prof_g -name '*kernel_trace.csv' -delete
