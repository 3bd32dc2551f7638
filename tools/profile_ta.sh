#!/bin/bash
# TA / TCP (vector-memory address unit and L1) counters of the SpMV sweep, in groups small enough for the hardware's counter
# slots (a single oversized --pmc group aborts rocprofv3 with error 38, "exceeds the capabilities of the hardware").
# One group per rocprofv3 run, no tracing domain next to --pmc, the program itself right after `--`.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--skip-cpu --skip-spgemm --skip-vendor --skip-structures --steps 50 --warmup 5 $@"
i=0
for G in "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_ta$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_ta$i.log 2>&1
  echo "group $i ($G): rc=$?"
done
