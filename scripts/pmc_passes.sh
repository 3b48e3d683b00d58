#!/bin/bash
# rocprofv3 counter passes (each --pmc set is its own run; never combined with trace domains)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python $GRAFT_REPO_ROOT/scripts/eval_loop.py > $OUT/p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
find $OUT -name "*.csv" | head -30
