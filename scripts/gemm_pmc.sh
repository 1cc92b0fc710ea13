#!/bin/bash
# SQ counter passes of the GEMM probe (scripts/gemm_probe.py):  gpurun -- bash scripts/gemm_pmc.sh [tag]
TAG=${1:-gemm}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLRS_MW_STREAM_WORDS=0      # counter collection serialises kernels: the iteration's streams synchronise through events only
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${TAG}_$i -- python3 $R/scripts/gemm_probe.py > $OUT/pmc_${TAG}_$i.log 2>&1; echo "pass $i rc=$?"
  f=$(find $OUT/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    for C in $SET; do python3 $R/scripts/pmc_summary.py "$f" $C | grep -i "gemm" ; done > $OUT/pmc_${TAG}_${i}_summary.csv
    cat $OUT/pmc_${TAG}_${i}_summary.csv
  fi
  find $OUT/pmc_${TAG}_$i -name '*.csv' -size +4M -delete
done
