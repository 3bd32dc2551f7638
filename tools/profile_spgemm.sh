#!/bin/bash
# rocprofv3 passes for the SpGEMM pipeline on one generator case (arg 2: fem | cage | dense | banded | rmat16), fp16 MFMA path only.
# kernel trace + stats first, then every PMC group in its own run (never combined with a tracing domain).
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CASE=${2:-cage}; EXTRA=$3   # arg 3: --fp32 for the fp32 V15 configuration
mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/source_hash.py > $OUT/source_hash.json
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/spgemm_stages.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $P $CASE --quick $EXTRA > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 $P $CASE --quick $EXTRA > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $P $CASE --quick $EXTRA > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $P $CASE --quick $EXTRA > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $P $CASE --quick $EXTRA > $OUT/pmc_sq2.log 2>&1
# matrix-core utilisation (north_star): busy cycles of the MFMA pipe and the fp16 MFMA op count
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $P $CASE --quick $EXTRA > $OUT/pmc_mfma.log 2>&1
find $OUT -name "*.csv" | head -30
