#!/bin/bash
# rocprofv3 passes for the SpMV bench: kernel trace + stats, then PMC passes (separate runs, no tracing domains mixed in)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/source_hash.py > $OUT/source_hash.json
cd /tmp && export TMPDIR=/tmp
ARGS="--skip-cpu --skip-spgemm --skip-vendor --steps 100 --warmup 10 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
