#!/bin/bash
# SQ counter passes of the assembly-only loop (scripts/w3_profile.py):  gpurun -- bash scripts/w3_pmc.sh [tag] [copies]
TAG=${1:-w3}
COPIES=${2:-8192}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLRS_MW_STREAM_WORDS=0      # counter collection serialises kernels: the iteration's streams synchronise through events only
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${TAG}_$i -- python3 $R/scripts/w3_profile.py $COPIES 5 > $OUT/pmc_${TAG}_$i.log 2>&1; echo "pass $i rc=$?"
  f=$(find $OUT/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    for C in $SET; do python3 $R/scripts/pmc_summary.py "$f" $C | grep -i "assemble" ; done > $OUT/pmc_${TAG}_${i}_summary.csv
    cat $OUT/pmc_${TAG}_${i}_summary.csv
  fi
  find $OUT/pmc_${TAG}_$i -name '*.csv' -size +4M -delete
done
