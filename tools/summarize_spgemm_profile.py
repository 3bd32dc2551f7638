#!/usr/bin/env python3
"""Condenses a tools/profile_spgemm.sh output directory into profiles/<tag>_{traffic,mfma}.json for one block-MAC kernel.
usage: summarize_spgemm_profile.py <gpurun_out/dir> <tag> <kernel-substring> <surviving tasks> <workload text>"""
import csv, glob, json, os, sys
src, tag, kern, tasks, wl = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
try:
    commit = open(os.path.join(src, "commit.txt")).read().strip() or None  # `git rev-parse --short HEAD` of the clean tree that was profiled
except OSError:
    commit = None
try:
    src_hash = json.load(open(os.path.join(src, "source_hash.json")))  # tools/source_hash.py, run on the box beside the passes
except (OSError, ValueError):
    src_hash = None


def counters(sub):
    f = sorted(glob.glob(os.path.join(src, sub, "*/*_counter_collection.csv")), key=os.path.getmtime)[-1]
    acc, n = {}, {}
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            c = r["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"]); n[c] = n.get(c, 0) + 1
    return {c: acc[c] / n[c] for c in acc}


st = sorted(glob.glob(os.path.join(src, "trace/*/*_kernel_stats.csv")), key=os.path.getmtime)[-1]
avg_ns = [float(r["AverageNs"]) for r in csv.DictReader(open(st)) if kern in r["Name"]][0]
fetch, write = counters("pmc_fetch"), counters("pmc_write")
traffic = int(2 * 1024 * fetch["FETCH_SIZE"] + 1024 * write["WRITE_SIZE"])
json.dump({"kernel": kern, "profiled_at_commit": commit, "source_sha256": src_hash, "fetch_size_kib": fetch["FETCH_SIZE"], "write_size_kib": write["WRITE_SIZE"], "traffic_bytes_per_launch": traffic,
           "method": "2*FETCH_SIZE + WRITE_SIZE (KiB), separate --pmc passes; factor 2 calibrated by experiments/fetch_calib.hip"},
          open("profiles/%s_traffic.json" % tag, "w"), indent=1)
m, s1, s2 = counters("pmc_mfma"), counters("pmc_sq1"), counters("pmc_sq2")
cyc = s2["GRBM_GUI_ACTIVE"] / 8.0
json.dump({"kernel": kern, "profiled_at_commit": commit, "source_sha256": src_hash, "workload": wl, "avg_ns": avg_ns, "SQ_VALU_MFMA_BUSY_CYCLES": m["SQ_VALU_MFMA_BUSY_CYCLES"],
           "SQ_INSTS_VALU_MFMA_MOPS_F16": m["SQ_INSTS_VALU_MFMA_MOPS_F16"], "SQ_INSTS_MFMA": m["SQ_INSTS_MFMA"], "kernel_cycles_per_xcd": cyc,
           "mfma_util": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4), "valu_per_task": round(s1["SQ_INSTS_VALU"] / tasks, 2),
           "salu_per_task": round(s2["SQ_INSTS_SALU"] / tasks, 2), "lds_instr_per_task": round(s1["SQ_INSTS_LDS"] / tasks, 2),
           "lds_busy_frac": round(m["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), 4),
           "method": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; separate --pmc pass (tools/profile_spgemm.sh)"},
          open("profiles/%s_mfma.json" % tag, "w"), indent=1)
print(open("profiles/%s_traffic.json" % tag).read(), open("profiles/%s_mfma.json" % tag).read())
