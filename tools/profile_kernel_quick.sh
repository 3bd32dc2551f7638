#!/bin/bash
# quick counter passes for ONE kernel of the SpGEMM stage tool: profile_kernel_quick.sh <out tag> <case> <kernel substring> [extra args of spgemm_stages.py]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CASE=$2; KERN=$3; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/spgemm_stages.py
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 $P $CASE --quick "$@" > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $P $CASE --quick "$@" > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_mfma -- python3 $P $CASE --quick "$@" > $OUT/pmc_mfma.log 2>&1
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, os, sys
out, kern = sys.argv[1], sys.argv[2]
for sub in ("pmc_sq1", "pmc_sq2", "pmc_mfma"):
    fs = sorted(glob.glob(os.path.join(out, sub, "*/*_counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        print(sub, "no csv"); continue
    acc, n = {}, {}
    for r in csv.DictReader(open(fs[-1])):
        if kern in r["Kernel_Name"]:
            c = r["Counter_Name"]; acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"]); n[c] = n.get(c, 0) + 1
    for c in sorted(acc):
        print("%-8s %-28s %16.0f  (avg of %d dispatches)" % (sub, c, acc[c] / n[c], n[c]))
PY
