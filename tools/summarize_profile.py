#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (gpurun_out/<name>) into profiles/<tag>_*.{csv,md} (tracked)."""
import csv, glob, collections, os, shutil, sys


def profiled_commit(src):
    """the commit the profiled tree was at: <src>/commit.txt (written by the person launching the gpurun call, `git rev-parse --short HEAD`
    on a clean tree), else None -- bench.py only quotes counter figures that carry this stamp and whose kernel sources have not changed since"""
    try:
        return open(os.path.join(src, "commit.txt")).read().strip() or None
    except OSError:
        return None


src, tag, kern = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "spmv_sweep")
cmd = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --skip-cpu --skip-spgemm --steps 100 --warmup 10"
os.makedirs("profiles", exist_ok=True)
stats = sorted(glob.glob(os.path.join(src, "trace/*/*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
lines = ["# rocprofv3 summary `%s` (source: %s)" % (tag, src), "",
         "command: `rocprofv3 --kernel-trace --stats --output-format csv -- %s` and, in separate runs, "
         "`rocprofv3 --pmc <counters>` with the same command." % cmd, ""]
if stats:
    shutil.copy(stats[0], "profiles/%s_kernel_stats.csv" % tag)
    lines += ["## kernel-trace --stats (top kernels)", "", "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    rows = sorted(csv.DictReader(open(stats[0])), key=lambda r: -float(r["TotalDurationNs"]))
    for i, row in enumerate(rows):
        if i >= 14: break
        nm = row["Name"]
        if nm.startswith("_Z"):  # left mangled by the tool: keep the readable middle
            import re
            m = re.search(r"\d+([a-z_0-9]+_kernel)", nm)
            nm = m.group(1) if m else nm[:60]
        nm = nm.replace("void ", "").replace("bmsp::(anonymous namespace)::", "").replace("bmsp::", "").split("(")[0][:80]
        lines.append("| %s | %s | %.0f | %s | %s | %s |" % (nm, row["Calls"], float(row["AverageNs"]), row["MinNs"], row["MaxNs"], row["Percentage"]))
lines += ["", "## PMC counters of `%s` (average per dispatch)" % kern, "", "| pass | counter | value |", "|---|---|---|"]
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d): continue
    f = sorted(glob.glob(os.path.join(d, "*/*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f[0])):
        if kern not in row["Kernel_Name"]: continue
        a = agg[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for c, (n, s) in sorted(agg.items()):
        lines.append("| %s | %s | %.1f |" % (os.path.basename(d), c, s / n))
lines += ["", "Notes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request "
          "(MI355X_MICROARCH.md, HBM section; experiments/fetch_calib.hip confirms the factor 2 for 8- and 16-byte-per-lane streams and for "
          "4-byte gathers).", ""]
# HBM traffic of the kernel per launch, corrected as MI355X_MICROARCH.md prescribes and as experiments/fetch_calib.hip confirms for
# this kernel's access shapes (8-byte-per-lane streams and 4-byte gathers both count 64 B per 128-B request): 2 x FETCH_SIZE + WRITE_SIZE
import json
tot = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    f = sorted(glob.glob(os.path.join(d, "*/*_counter_collection.csv")), key=os.path.getmtime, reverse=True) if os.path.isdir(d) else []
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f[0])):
        if kern in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for c, (n, sm) in agg.items(): tot[c] = sm / n
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    try:
        src_hash = json.load(open(os.path.join(src, "source_hash.json")))  # tools/source_hash.py, run on the box beside the passes
    except (OSError, ValueError):
        src_hash = None
    tr = {"kernel": kern, "profiled_at_commit": profiled_commit(src), "source_sha256": src_hash, "fetch_size_kib": tot["FETCH_SIZE"], "write_size_kib": tot["WRITE_SIZE"],
          "traffic_bytes_per_launch": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024),
          "method": "2*FETCH_SIZE + WRITE_SIZE (KiB), separate --pmc passes; factor 2 calibrated by experiments/fetch_calib.hip"}
    json.dump(tr, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
    lines += ["", "HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE = %.1f MB" % (tr["traffic_bytes_per_launch"] / 1e6), ""]
open("profiles/%s_summary.md" % tag, "w").write("\n".join(lines))
print("\n".join(lines))
