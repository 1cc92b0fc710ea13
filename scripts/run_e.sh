cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_mw_parity.py tests/test_reference_vectors.py -m gpu -x -q 2>&1 | tail -6 &&
timeout -k 10 300 python bench.py --skip-cpu > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err && python -c "
import json;d=json.loads(open('gpurun_out/bench_e.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])" &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_e -- python3 $GRAFT_REPO_ROOT/scripts/mw_iter_profile.py ce_8_15 3 > $GRAFT_REPO_ROOT/gpurun_out/trace_e.log 2>&1; echo "rocprof rc=$?"
t=$(find $GRAFT_REPO_ROOT/gpurun_out/trace_e -name '*kernel_trace.csv' | head -1); python $GRAFT_REPO_ROOT/scripts/iter_timeline.py $t > $GRAFT_REPO_ROOT/gpurun_out/timeline_e.txt; tail -45 $GRAFT_REPO_ROOT/gpurun_out/timeline_e.txt
find $GRAFT_REPO_ROOT/gpurun_out/trace_e -name '*kernel_trace.csv' -delete
