#!/bin/bash
# usage: tools/pmc_generic.sh <outdir under gpurun_out> <kernel name regex> <python script + args...>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
PAT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "$PAT" > $OUT/summary.txt 2>&1 || true
echo done
