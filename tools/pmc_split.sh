#!/bin/bash
# Counter passes of the split-fp16 GEMM (c_proj shape by default), for the library in $AACLIP_LIB (default: product):
#   bash tools/pmc_split.sh <tag> [shape]   -> gpurun_out/<tag>_split_<shape>_pmc.json
set -e
TAG=${1:-x}
SHAPE=${2:-c_proj}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT/pmc_$TAG
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32"
i=0
DBS=""
for P in "$P1" "$P2"; do
  i=$((i+1))
  D=$OUT/pmc_$TAG/${SHAPE}_p$i
  rm -rf $D
  rocprofv3 --pmc $P -d $D -o r -- python3 tools/profile_split.py $SHAPE > $OUT/pmc_$TAG/${SHAPE}_p$i.log 2>&1
  DBS="$DBS $(find $D -name '*_results.db' | head -1)"
done
python3 tools/rocpd_summary.py pmcjson $OUT/${TAG}_split_${SHAPE}_pmc.json gemm16_256x $DBS > $OUT/pmc_$TAG/${SHAPE}_summary.log 2>&1
rm -rf $OUT/pmc_$TAG/${SHAPE}_p[0-9]
tail -30 $OUT/pmc_$TAG/${SHAPE}_summary.log
