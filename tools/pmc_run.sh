#!/bin/bash
# Counter profile of one command, every PMC group in its own rocprofv3 pass (kernel trace only beside it):
#   bash tools/pmc_run.sh <tag> <command ...>     -> gpurun_out/<tag>_pmc.json (per-kernel averages, tools/pmc_sq.py)
#                                                    gpurun_out/<tag>_kernel_stats.csv (durations of the first pass)
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
finddb() { find "$1" -name '*.db' | head -1; }
i=0
DBS=""
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G -d $OUT/p$i -- "$@" > $OUT/p$i.out 2> $OUT/p$i.err )
  DB=$(finddb $OUT/p$i)
  [ -n "$DB" ] && DBS="$DBS $DB"
done
python $R/tools/pmc_sq.py $R/gpurun_out/${TAG}_pmc.json $DBS
python $R/tools/rocpd_stats.py "$(finddb $OUT/p1)" $R/gpurun_out/${TAG}_kernel_stats.csv > /dev/null
rm -rf $OUT/p*/
