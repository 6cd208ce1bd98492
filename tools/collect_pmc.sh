#!/bin/bash
# Counter passes for the three kernels VERDICT r1 named (run on the GPU box from the repo root):
#   bash tools/collect_pmc.sh <tag>        -> gpurun_out/<tag>_{attn,out_proj,c_fc}_pmc.json (+ traffic json for c_fc)
# Counters go in their own runs (no trace domains with --pmc); FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT/pmc_$TAG
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VALU_CVT"
for K in ${KERNELS:-attn out_proj c_fc}; do
  i=0
  DBS=""
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    D=$OUT/pmc_$TAG/${K}_p$i
    rm -rf $D
    rocprofv3 --pmc $P -d $D -o r -- python3 tools/profile_kernels.py --kernel $K --iters 5 > $OUT/pmc_$TAG/${K}_p$i.log 2>&1
    DBS="$DBS $(find $D -name '*_results.db' | head -1)"
  done
  case $K in attn) SUB=attn16;; *) SUB=gemm16_256x;; esac
  python3 tools/rocpd_summary.py pmcjson $OUT/${TAG}_${K}_pmc.json $SUB $DBS > $OUT/pmc_$TAG/${K}_summary.log 2>&1
  if [ $K = c_fc ]; then
    set -- $DBS
    python3 tools/rocpd_summary.py traffic $3 $4 gemm16_256x 87680 4096 1024 $OUT/${TAG}_cfc_gemm_traffic.json >> $OUT/pmc_$TAG/${K}_summary.log 2>&1
  fi
  rm -rf $OUT/pmc_$TAG/${K}_p[0-9]        # the rocpd databases are tens of MB each; only the summaries travel back
  echo "== $K"; tail -25 $OUT/pmc_$TAG/${K}_summary.log
done
# plain per-kernel timing of one default bench step set (kernel trace + stats), summarised to csv
rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o $TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/prof_$TAG.json 2> $OUT/prof_$TAG.log
python3 tools/rocpd_summary.py stats $(find $OUT/prof_$TAG -name '*_results.db' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/prof_$TAG
cat $OUT/prof_$TAG.json
