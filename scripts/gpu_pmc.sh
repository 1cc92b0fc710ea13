#!/bin/bash
# HBM traffic of the bench kernels from PMC counters (separate passes, counters only: no tracing flags).
#   gpurun --timeout 900 -- bash scripts/gpu_pmc.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLRS_MW_STREAM_WORDS=0      # counter collection serialises kernels: the iteration's streams synchronise through events only
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${TAG}_$C -- python3 $R/bench.py --skip-cpu --steps 20 --warmup 5 > $OUT/pmc_${TAG}_$C.json 2> $OUT/pmc_${TAG}_$C.err; echo "$C rc=$?"
  f=$(find $OUT/pmc_${TAG}_$C -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $R/scripts/pmc_summary.py "$f" $C > $OUT/pmc_${TAG}_${C}_summary.csv && cat $OUT/pmc_${TAG}_${C}_summary.csv
  find $OUT/pmc_${TAG}_$C -name '*.csv' -size +4M -delete
done
