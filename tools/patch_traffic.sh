#!/bin/bash
# XCD patch shape vs HBM-side traffic and time of the split c_fc product (5.74x algorithmic reads).
#   bash tools/patch_traffic.sh > gpurun_out/r04/cfc_patch_shapes.txt
# For each PM x PN (tiles of one patch = what one XCD works on together): ms per launch un-profiled, then L2 fill bytes
# (FETCH_SIZE, doubled as the guide prescribes for gfx950) and write-back bytes (WRITE_SIZE) per launch from two
# rocprofv3 --pmc passes.  Algorithmic: A 359 MB + W 16.8 MB read, 1.44 GB written.
export TMPDIR=/tmp
OUT=/tmp/patch_pmc   # (databases are large: scratch on the box, only the printed summary is kept)
mkdir -p $OUT
for P in 8,4 4,8 2,16 16,2 8,8 4,16; do
  export AACLIP_GEMM_PATCH=$P
  echo "== patch $P (PM,PN)"
  ONLY=c_fc python3 tools/bench_split_gemm.py 2>&1 | grep "c_fc"
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$OUT/${P/,/x}_$C
    rm -rf $D
    ONLY=c_fc rocprofv3 --pmc $C -d $D -o r -- python3 tools/bench_split_gemm.py > $D.log 2>&1
    DB=$(find $D -name '*_results.db' | head -1)
    python3 tools/rocpd_summary.py pmc $DB gemm16_256x_kernelIDF16_Li1ELi4E 2>&1 | tail -1
  done
done
