#!/bin/bash
# as profile_ta.sh, but on one structure of tools/spmv_cases.py: usage profile_ta_case.sh <outdir-tag> <case-substring>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CASE=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/tools/spmv_cases.py "$CASE" > $OUT/g$i.log 2>&1
  echo "group $i ($G): rc=$?"
done
