#!/bin/bash
# usage: tools/pmc_k1.sh <outdir under gpurun_out>   -- rocprofv3 counter passes of K1 at the Path-A shape of configs[2]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/run_k1.py 2 3 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "sweep_corr" > $OUT/summary.txt 2>&1 || true
echo done
