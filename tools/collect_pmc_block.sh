#!/bin/bash
# Counter passes over the kernels AS THE TOWER LAUNCHES THEM (tools/profile_block.py), one JSON per kernel:
#   bash tools/collect_pmc_block.sh <tag> <precision> [clip-weights]  -> gpurun_out/<tag>_<precision>_<kernel>_pmc.json
# plus the HBM traffic of the dominant kernel (c_fc) for bench.py's roofline.traffic.  Counters go in their own runs
# (no trace domains with --pmc); FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
set -e
TAG=${1:-r03}
PREC=${2:-fp16x2}
CW=${3:-fp32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT/pmc_$TAG
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
DBS=""
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  D=$OUT/pmc_$TAG/${PREC}_p$i
  rm -rf $D
  rocprofv3 --pmc $P -d $D -o r -- python3 tools/profile_block.py --precision $PREC --clip-weights $CW --blocks 3 --iters 3 > $OUT/pmc_$TAG/${PREC}_p$i.log 2>&1
  DBS="$DBS $(find $D -name '*_results.db' | head -1)"
done
if [ $PREC = fp16x2 ]; then NPS=$([ $CW = fp16 ] && echo 3 || echo 4); ATT="attn16s_kernel"; else NPS=0; ATT="attn16x2_kernel"; fi
# (the counter database holds MANGLED kernel names: gemm16_256x_kernel<_Float16, EPI, NP> = ...IDF16_Li<EPI>ELi<NP>E...)
declare -A SUB=( [qkv]="gemm16_256x_kernelIDF16_Li0ELi${NPS}E" [c_fc]="gemm16_256x_kernelIDF16_Li1ELi${NPS}E" [resid]="gemm16_256x_kernelIDF16_Li2ELi${NPS}E" [attn]="$ATT" )
# out_proj and c_proj are ONE instantiation (EPI_BIAS_RESID) launched alternately, out_proj first in every block
SUB[out_proj]="${SUB[resid]}@0/2"
SUB[c_proj]="${SUB[resid]}@1/2"
for K in qkv c_fc out_proj c_proj attn; do
  python3 tools/rocpd_summary.py pmcjson $OUT/${TAG}_${PREC}_${K}_pmc.json "${SUB[$K]}" $DBS > $OUT/pmc_$TAG/${PREC}_${K}_summary.log 2>&1 || true
  echo "== $K (${SUB[$K]})"; tail -16 $OUT/pmc_$TAG/${PREC}_${K}_summary.log
done
set -- $DBS
python3 tools/rocpd_summary.py traffic $3 $4 "${SUB[c_fc]}" 87680 4096 1024 $OUT/${TAG}_cfc_gemm_traffic_${PREC}.json $PREC $CW >> $OUT/pmc_$TAG/${PREC}_c_fc_summary.log 2>&1 || true
tail -3 $OUT/pmc_$TAG/${PREC}_c_fc_summary.log
rm -rf $OUT/pmc_$TAG/${PREC}_p[0-9]
