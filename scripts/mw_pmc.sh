#!/bin/bash
# SQ counters of the multi-word assembly on the many-cluster instance (counter pass only) + timing of the same command without counters
#   gpurun --timeout 900 -- bash scripts/mw_pmc.sh [tag]    -> gpurun_out/<tag>_pmc_mw_assembly.csv
TAG=${1:-d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLRS_MW_STREAM_WORDS=0      # counter collection serialises kernels: the iteration's streams synchronise through events only
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mw_$TAG -- python3 $R/scripts/mw_roofline.py 5 1024 2 on > $OUT/pmc_mw_$TAG.log 2>&1; echo "pmc rc=$?"
cd $R
f=$(find $OUT/pmc_mw_$TAG -name '*counter_collection.csv' | head -1)
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES; do python3 scripts/pmc_summary.py $f $c | grep -v "potrf_x\|fill\|copy"; done > $OUT/${TAG}_pmc_mw_assembly.csv
cat $OUT/${TAG}_pmc_mw_assembly.csv
rm -rf $OUT/pmc_mw_$TAG
timeout -k 10 300 python3 scripts/mw_roofline.py 5 1024 5 on 2>&1 | tail -2
timeout -k 10 300 python3 scripts/mw_roofline.py 5 128 5 on 2>&1 | tail -2 | head -1
timeout -k 10 300 python3 scripts/mw_roofline.py 6 1024 3 on 2>&1 | tail -2 | head -1
timeout -k 10 300 python3 scripts/mw_roofline.py 4 1024 3 on 2>&1 | tail -2 | head -1
