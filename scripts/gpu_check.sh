#!/bin/bash
# One GPU-box pass: smoke, GPU parity tests, bench line, rocprofv3 kernel stats.  Run through gpurun:
#   gpurun --timeout 1500 -- bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke_$TAG.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu_$TAG.log
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cat $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py --skip-cpu > $OUT/prof_$TAG.json 2> $OUT/prof_$TAG.err; echo "rocprof rc=$?"
find $OUT/prof_$TAG -name '*kernel_stats*' | head; f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -20 "$f"
t=$(find $OUT/prof_$TAG -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python $R/scripts/trace_by_grid.py "$t" > $OUT/prof_${TAG}_by_grid.csv && head -12 $OUT/prof_${TAG}_by_grid.csv
# keep only the small summaries (the raw trace can be large)
find $OUT/prof_$TAG -name '*kernel_trace.csv' -size +8M -delete
