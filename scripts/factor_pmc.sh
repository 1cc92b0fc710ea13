#!/bin/bash
# SQ counters of the kernels of one multi-word interior-point solve (counter pass only, no tracing flags) + a kernel trace of the same
# command for the durations -> profiles/<round>/<tag>_pmc_iter_sq_counters.csv and profiles/mw_factor_counters.json (bench.py: roofline_timed)
#   gpurun --timeout 900 -- bash scripts/factor_pmc.sh [tag]
TAG=${1:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLRS_MW_STREAM_WORDS=0      # counter collection serialises kernels: the iteration's streams synchronise through events only
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_iter_$TAG -- python3 $R/scripts/mw_iter_profile.py ce_8_15 2 > $OUT/pmc_iter_$TAG.log 2>&1; echo "pmc rc=$?"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_iter_$TAG -- python3 $R/scripts/mw_iter_profile.py ce_8_15 2 > $OUT/trace_iter_$TAG.log 2>&1; echo "trace rc=$?"
cd $R
python3 scripts/factor_counters.py $(find $OUT/pmc_iter_$TAG -name '*counter_collection.csv' | head -1) $(find $OUT/trace_iter_$TAG -name '*kernel_trace.csv' | head -1) $OUT/${TAG}_pmc_iter_sq_counters.csv $OUT/mw_factor_counters.json "$TAG"
cat $OUT/${TAG}_pmc_iter_sq_counters.csv
rm -rf $OUT/pmc_iter_$TAG $OUT/trace_iter_$TAG
