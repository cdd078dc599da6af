#!/bin/bash
# PMC passes for the GEMM micro-benchmark (separate --pmc runs, kernel-trace only).  Usage: tools/pmc_gemm.sh OUTDIR SHAPE...
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for P in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $P | cut -d' ' -f1)
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $R/tools/gemm_bench.py "$@" > $OUT/pmc_$tag.log 2>&1
done
