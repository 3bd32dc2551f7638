#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native bmSparse engine.

metric (BASELINE.json): "SpMV effective GB/s + SpGEMM GFLOP/s on SuiteSparse; % HBM/MFMA roofline".
The ONE JSON line carries the SpMV figure as `value` (configs[1]: bmSparse SpMV fp32 on webbase-1M, 1 GPU), the same sweep
on two denser structures under `spmv_structures`, the SpGEMM figures (configs[2], configs[3] and a dense-tile ceiling run)
under `spgemm`, and the CPU baselines (cusp::multiply restatement: SpMV, SpGEMM, and configs[0] -- host SpMV on a cant-sized
matrix) under `cpu_baseline`.

A step = one SpMV sweep u = A*v over one resident matrix.  SuiteSparse files cannot be downloaded here; unless
`--mtx-dir` holds webbase-1M.mtx the workload is the synthetic stand-in of SURVEY.md 8(d): R-MAT scale 20, edge factor
2, identity added (1 048 576 rows, ~3.13 M nnz -- webbase-1M has 1 000 005 rows, 3 105 536 nnz).  The matrix in bmSparse
form is ~70 MB and would sit in the 256 MiB Infinity Cache, so the timed loop rotates over enough device-resident copies
(> 512 MiB in total) that every sweep streams from HBM; the cache-warm figure is reported beside it as
`warm_ms_per_step`.  Inputs are resident in HBM and every copy's cached sweep plan is built (bmsp_matrix_prepare + one
untimed sweep per copy) before the timed region starts, whatever --warmup is.

N > 1 (one process per GPU, launched by torch.distributed.run): every rank sweeps its own copies (fixed work per GPU,
no data-path collective; weak scaling); value = bytes swept by all ranks / max-over-ranks time.  The row-panel-sharded
SpGEMM with its RCCL allgatherv (configs[4]) is timed in the same run and reported under `spgemm_sharded`.
"""
import argparse
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "bmsparse-spgemm-spmv_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a copy kernel reaches
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA spec
FP32_PEAK_TFLOPS = 157.3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mtx-dir", default=os.environ.get("BMSP_MTX_DIR", ""))
    ap.add_argument("--scale", type=int, default=20, help="R-MAT scale of the synthetic SpMV workload")
    ap.add_argument("--edge-factor", type=float, default=2.0)
    ap.add_argument("--spmv-matrix", default="", help="experiment override: banded:N:HB | rmat:SCALE:EF | cage:N | fem:SIDE")
    ap.add_argument("--batched", type=int, default=-1, help="-1 auto, 0 / 1 force the SpMV variant")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-spgemm", action="store_true")
    ap.add_argument("--skip-structures", action="store_true", help="skip the banded SpMV structures")
    ap.add_argument("--skip-vendor", action="store_true", help="skip the rocSPARSE comparison column")
    ap.add_argument("--skip-cli", action="store_true", help="skip the run of the drop-in executable on the SpGEMM workloads")
    ap.add_argument("--skip-rmat22", action="store_true", help="skip the scale-22 R-MAT product (configs[4] on one GPU: ~40 s of generation + build)")
    ap.add_argument("--only-spgemm", default="", help="experiment: run only the SpGEMM case whose tag contains this (fem | cage | ceiling)")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="bound on the CPU-baseline work, all legs together")
    ap.add_argument("--vendor-timeout", type=float, default=150.0, help="the rocSPARSE column runs in a child process; on a fresh box paging librocsparse in can take minutes")
    ap.add_argument("--vendor-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the child launcher started by --gpus N > 1 (0: a free one)")
    return ap.parse_args(argv)


def launcher_command(gpus, argv, port):
    """the command `--gpus N` (N > 1, no WORLD_SIZE in the environment) starts: the driver's own form, one rank per GPU over RCCL"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if not a.startswith("--master-port")]


def check_child_line(stdout_text, gpus):
    """(line, error): rank 0's JSON line out of a child's stdout; an error when there is none, when it reports another number of GPUs
    than asked for, or when RCCL saw fewer ranks"""
    lines = [l for l in stdout_text.splitlines() if l.startswith("{")]
    if not lines:
        return None, "the %d-rank child printed no JSON line" % gpus
    try:
        d = json.loads(lines[-1])
    except ValueError as e:
        return None, "unparsable line from the child: %s" % e
    if d.get("n_gpus") != gpus:
        return lines[-1], "asked for --gpus %d but the line reports n_gpus = %r" % (gpus, d.get("n_gpus"))
    if d.get("rccl_ranks") != gpus:
        return lines[-1], "asked for --gpus %d but the RCCL communicator had %r ranks" % (gpus, d.get("rccl_ranks"))
    return lines[-1], None


def launch_ranks(args, argv, cmd=None):
    """`python bench.py --gpus N` with N > 1 outside a launcher: start the N ranks as a CHILD process -- before this process has made
    any GPU call (an exec from a process that has touched the GPU is not allowed on this pool, and this one never touches it) -- relay
    rank 0's line, and fail loudly when the line is not an N-GPU line.  The reference's analogue is "the driver runs the executable"
    (spgemm_run_batch.sh:15)."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd or launcher_command(args.gpus, argv, port), stdout=subprocess.PIPE, text=True, env=env)
    line, err = check_child_line(r.stdout, args.gpus)
    if line:
        print(line, flush=True)
    if err or r.returncode:
        print("[bench] --gpus %d: %s" % (args.gpus, err or "child exit code %d" % r.returncode), file=sys.stderr, flush=True)
        return r.returncode or 3
    return 0


# SuiteSparse dimensions of the BASELINE configs (top): rows = cols, entries after symmetric expansion.  Asserted
# whenever --mtx-dir supplies the file, so that a stand-in or a truncated download can never pass for the named matrix.
SUITESPARSE_DIMS = {"cant.mtx": (62451, 4007383), "webbase-1M.mtx": (1000005, 3105536), "2cubes_sphere.mtx": (101492, 1647264), "cage12.mtx": (130228, 2032536)}


def assert_suitesparse_dims(fname, info):
    rows, nnz = SUITESPARSE_DIMS[os.path.basename(fname)]
    got = (info["num_rows"], info["num_cols"], info["nnz"])
    assert got == (rows, rows, nnz), "%s: expected %d x %d with %d entries (SURVEY.md section 8), the file holds %d x %d with %d" % ((fname, rows, rows, nnz) + got)


def csr_bytes(rows, nnz):
    """CUSP's bytes-per-SpMV convention for int32/fp32 CSR (cusp/performance/spmv/bytes_per_spmv.h:31-39)."""
    return 2 * 4 * rows + 4 * nnz + 2 * 4 * nnz + 2 * 4 * rows


def bmsp_spmv_bytes(info, itemsize=4):
    """algorithmic bytes of one bmSparse sweep (BASELINE.md 4 / SURVEY.md 8(d)), with the layout the kernel reads:
    key+bitmap+offset per block (24 B), values, the uint32 dense block-row pointer, x once, y once."""
    nbr = (info["num_rows"] + 7) // 8
    return 24 * info["block_num"] + itemsize * info["nnz"] + 4 * (nbr + 1) + itemsize * info["num_cols"] + itemsize * info["num_rows"]


def parse_matrix_spec(gen, spec):
    kind, *a = spec.split(":")
    return {"banded": lambda: gen.banded(int(a[0]), int(a[1])), "rmat": lambda: gen.rmat(int(a[0]), float(a[1])),
            "cage": lambda: gen.cage_like(int(a[0])), "fem": lambda: gen.fem_like(int(a[0]))}[kind]()


def load_spmv_workload(args):
    from pybmsp import gen
    path = os.path.join(args.mtx_dir, "webbase-1M.mtx") if args.mtx_dir else ""
    if path and os.path.exists(path):
        return {"name": "webbase-1M (SuiteSparse file)", "path": path}
    if args.spmv_matrix:
        return {"name": args.spmv_matrix + " (experiment)", "coo": parse_matrix_spec(gen, args.spmv_matrix)}
    n, _, r, c, v = gen.rmat(args.scale, args.edge_factor, seed=1)
    return {"name": "rmat(scale=%d, edge_factor=%g)+I, stand-in for webbase-1M" % (args.scale, args.edge_factor),
            "coo": (n, n, r, c, v)}


def profile_value(pattern, key, kernel_sources):
    """(value, stamp) from the newest committed rocprofv3 summary matching profiles/<pattern>.  A counter figure is only quoted while it
    still describes the kernel that runs: every summary carries the sha256 of the kernel sources of the tree it was profiled on
    (`source_sha256`, tools/source_hash.py run on the GPU box beside the passes) and the figure is DROPPED (None) when one of THIS kernel's
    source files no longer has that digest in the tree bench.py runs from -- a content check, so it works on the GPU box, which gets a
    snapshot without .git.  Summaries of earlier rounds carry a commit instead (`profiled_at_commit`): checked against `git log` where a
    history exists, dropped where it does not."""
    import hashlib
    import subprocess
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", pattern))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if key in d:
            best = (d[key], d, os.path.basename(f))
    if best is None:
        return None, None
    val, d, fname = best
    digests = d.get("source_sha256")
    if digests:
        for src in kernel_sources:
            try:
                now = hashlib.sha256(open(os.path.join(REPO, src), "rb").read()).hexdigest()
            except OSError:
                now = None
            if digests.get(src) != now:
                return None, "%s is stale: %s changed since it was profiled -> not quoted" % (fname, src)
        return val, "%s (source digests of %s match)" % (fname, ", ".join(os.path.basename(s) for s in kernel_sources))
    commit = d.get("profiled_at_commit")
    if not commit:
        return None, "%s: neither source digests nor a commit stamp -> not quoted" % fname
    try:
        out = subprocess.run(["git", "-C", REPO, "log", "--oneline", "%s..HEAD" % commit, "--"] + kernel_sources, capture_output=True, text=True, timeout=20)
        if out.returncode != 0:
            return None, "%s @ %s: no source digests and no history here -> not quoted" % (fname, commit)
        if out.stdout.strip():
            return None, "%s @ %s is stale: %d later commit(s) touch %s -> not quoted" % (fname, commit, len(out.stdout.strip().splitlines()), ", ".join(kernel_sources))
    except Exception:
        return None, "%s @ %s: no source digests and git not runnable here -> not quoted" % (fname, commit)
    return val, "%s @ %s" % (fname, commit)


class SpmvSet:
    """`copies` device-resident clones of one matrix (rotated so that sweeps stream from HBM), plans built, every copy swept once"""

    def __init__(self, B, np, first, x_kind="ones", min_bytes=512 * 2 ** 20):
        from pybmsp import gen
        self.B, self.L = B, B.lib()
        self.info = first.info()
        self.format_bytes = bmsp_spmv_bytes(self.info)
        self.alg_bytes = self.format_bytes  # replaced by the launched kernel's compulsory bytes once the plan exists (below)
        self.eff_bytes = csr_bytes(self.info["num_rows"], self.info["nnz"])
        self.copies = max(2, int(np.ceil(min_bytes / self.alg_bytes)) + 1)
        self.mats = [first] + [first.clone() for _ in range(self.copies - 1)]
        self.x = B.DeviceArray.from_host(gen.spmv_x(self.info["num_cols"], x_kind))  # v = 1 (SPMV.cu:279-281)
        self.ys = [B.DeviceArray(self.info["num_rows"], np.float32) for _ in range(self.copies)]
        # plan + position cache of one copy, timed: what the first product on a fresh matrix pays on top of a sweep
        B.synchronize()
        t0 = time.perf_counter()
        self.mats[0].prepare(1)
        B.synchronize()
        self.prepare_ms = (time.perf_counter() - t0) * 1e3
        # ... and what the SECOND matrix of a process pays (kernels loaded, pool warm): the per-matrix cost proper
        t0 = time.perf_counter()
        self.mats[1].prepare(1)
        B.synchronize()
        self.prepare_warm_ms = (time.perf_counter() - t0) * 1e3
        for m in self.mats[2:]:
            m.prepare(1)
        self.launch = B.spmv_launch_info(first, 0)
        self.alg_bytes = self.launch["compulsory_bytes"]
        for k in range(self.copies):  # plan builds and first-touch effects stay out of every timed region
            self.step(k, 0)
        B.synchronize()

    def step(self, i, variant):
        k = i % self.copies
        self.B.check(self.L.bmsp_spmv(self.mats[k].h, self.x.ptr, self.ys[k].ptr, variant, None))

    def timed(self, steps, variant, rotate=True):
        B = self.B
        e0, e1 = B.Event(), B.Event()
        e0.record()
        for i in range(steps):
            self.step(i if rotate else 0, variant)
        e1.record()
        B.synchronize()
        return e0.elapsed_ms(e1) / steps


_T0 = time.time()


def note(msg):
    """progress line on stderr (stdout carries exactly one JSON line)"""
    print("[bench %6.1f s] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def vendor_in_child(args):
    """the rocSPARSE column in a child process with a deadline, so that a cold librocsparse (GBs of code objects to page in on a fresh
    box) can never take the bench line down or past its time budget"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--vendor-child", "--scale", str(args.scale), "--edge-factor", str(args.edge_factor)]
    if args.skip_spgemm:
        cmd.append("--skip-spgemm")
    if args.only_spgemm:
        cmd += ["--only-spgemm", args.only_spgemm]
    if args.spmv_matrix:
        cmd += ["--spmv-matrix", args.spmv_matrix]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.vendor_timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-200:]}
    except subprocess.TimeoutExpired:
        return {"skipped": "rocSPARSE column did not finish within %.0f s (library page-in on a fresh box); rerun or raise --vendor-timeout" % args.vendor_timeout}
    except Exception as e:  # the column is optional: never let it take the bench line down
        return {"error": str(e)[:200]}


def vendor_child(args):
    import numpy as np
    from pybmsp import gen
    wl = load_spmv_workload(args)
    if "coo" in wl:
        n, _, r, c, v = wl["coo"]
        eff = csr_bytes(n, r.size)
    else:
        eff = 0
    try:
        print(json.dumps(vendor_column(np, gen, wl, eff, args)))
    except Exception as e:
        print(json.dumps({"error": str(e)[:200]}))


def main():
    args = parse()
    if args.vendor_child:
        return vendor_child(args)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("BMSP_FORCE_LAUNCH") == "1"):
        # (BMSP_FORCE_LAUNCH=1 --gpus 1 rehearses the launcher with one rank on a one-GPU box)
        os.environ["BMSP_FORCE_DIST"] = "1"
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("[bench] --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world), file=sys.stderr, flush=True)
        sys.exit(3)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    rccl_ranks = None
    use_dist = world > 1 or os.environ.get("BMSP_FORCE_DIST") == "1"  # the latter rehearses the N>1 plumbing with one rank
    if use_dist:
        import torch  # before the bmsp library: one HIP runtime for both (see pybmsp docstring)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout when the communicator comes up; stdout carries exactly one JSON line, so
        # the banner is sent to stderr (fd-level: the library writes from C)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
            ones = torch.ones(1, dtype=torch.int32, device="cuda")
            dist.all_reduce(ones)  # counted over the wire: the ranks the RCCL communicator actually holds
            rccl_ranks = int(ones.item()) if dist.get_backend() == "nccl" else 0
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    import numpy as np
    import pybmsp as B
    from pybmsp import gen
    B.set_device(local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        B.synchronize()
        if torch is not None:
            torch.cuda.synchronize()

    # ---------------- SpMV workload, resident in HBM ----------------
    wl = load_spmv_workload(args)
    note("SpMV workload generated")
    if "path" in wl:
        first = B.BmSpMatrix.from_mtx(wl["path"])
        assert_suitesparse_dims(wl["path"], first.info())
    else:
        n, _, r, c, v = wl["coo"]
        first = B.BmSpMatrix.from_coo(n, n, r, c, v)
    S = SpmvSet(B, np, first)
    note("SpMV copies resident, plans built")
    info, alg_bytes, eff_bytes, copies = S.info, S.alg_bytes, S.eff_bytes, S.copies
    variant = args.batched if args.batched >= 0 else 0

    for i in range(args.warmup):
        S.step(i, variant)
    sync(); barrier(); sync()
    e0, e1 = B.Event(), B.Event()
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        S.step(i, variant)
    e1.record()
    sync()
    wall = time.perf_counter() - t0  # this rank's K steps, from the common start (barrier + synchronise above) to its own completion ...
    barrier()                        # ... the closing barrier; the MAX over ranks below is the time at which the last rank was done
    dev_ms = e0.elapsed_ms(e1)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall * 1e3 / args.steps
    value = world * eff_bytes * args.steps / wall / 1e9  # every rank sweeps the same amount of its own data

    # cache-warm sweep (one copy only), for information
    S.timed(10, variant, rotate=False)
    warm_ms = S.timed(args.steps, variant, rotate=False)

    kern_ms = dev_ms / args.steps  # HIP events on the launch stream, over the timed region
    # what was launched and what that launch must move: asked from the library (bmsp_spmv_launch_info mirrors the launcher's decisions and
    # counts the kernel's own arrays from the plan -- SURVEY 8(d): "use that layout's compulsory bytes (never the larger figure)")
    linfo = B.spmv_launch_info(first, variant)
    compulsory = linfo["compulsory_bytes"]
    achieved = compulsory / (kern_ms * 1e-3) / 1e9
    # HBM traffic from the PMC counters (separate rocprofv3 passes, summarised under profiles/); only quoted for the workload
    # it was collected on
    traffic, traffic_src = None, None
    if "coo" in wl and not args.spmv_matrix and args.scale == 20 and args.edge_factor == 2.0:
        traffic, traffic_src = profile_value("r*_spmv_webbase_like*_traffic.json", "traffic_bytes_per_launch", ["bmsparse-spgemm-spmv_amd/csrc/spmv.hip"])
    roofline = {"bound": "hbm", "kernel": linfo["kernel"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": compulsory,
                "bytes_note": "compulsory bytes of the launched kernel's own layout, from the plan (bmsp_spmv_launch_info); format_bytes = 24 B per tile + values + row pointer + x + y",
                "format_bytes": linfo["format_bytes"], "format_frac": round(linfo["format_bytes"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "avg_launch_ms": round(kern_ms, 5)}

    out = {"metric": "bmSparse SpMV fp32 effective GB/s (CSR-convention bytes / time); SpGEMM GFLOP/s under `spgemm`",
           "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic" if "coo" in wl else "suitesparse",
           "config": {"workload": "bmSparse SpMV fp32, " + wl["name"], "rows": info["num_rows"], "nnz": info["nnz"],
                      "blocks": info["block_num"], "values_per_tile": round(info["nnz"] / max(1, info["block_num"]), 3),
                      "x": "ones", "variant": ["default (value-stream sweep over the cached plan)", "batched", "row-group"][variant],
                      "hbm_resident_copies_rotated": copies, "effective_bytes_per_spmv": eff_bytes},
           "warm_ms_per_step": round(warm_ms, 5), "warm_effective_GBs": round(eff_bytes / (warm_ms * 1e-3) / 1e9, 1),
           "prepare_ms": round(S.prepare_ms, 4), "prepare_warm_ms": round(S.prepare_warm_ms, 4),
           "prepare_note": "sweep plan + position cache of one matrix (bmsp_matrix_prepare, host wall time incl. its read-backs): outside the timed region, paid once per matrix; prepare_warm_ms = the same for the second matrix of the process (kernels loaded, pool warm)",
           "roofline": roofline}

    note("SpMV headline timed")
    # ---------------- the same sweep on denser structures (driver-verifiable roofline fractions) ----------------
    if not args.skip_structures and not use_dist and not args.spmv_matrix:
        out["spmv_structures"] = bench_spmv_structures(B, gen, np, S)
    mats0, x0, y0 = S.mats[0], S.x, S.ys[0]
    if not use_dist:
        del S.mats[1:], S.ys[1:]

    note("SpMV structures done")
    # ---------------- SpGEMM (configs[2], configs[3], dense-tile ceiling) on rank 0 / single GPU ----------------
    if not args.skip_spgemm and not use_dist:
        out["spgemm"] = bench_spgemm(B, gen, np, args)
    # the sharded operators are extras next to the headline line: a failure in them (RCCL rendezvous, memory) must not lose the line
    if use_dist and not args.skip_spgemm:
        try:
            out["spgemm_sharded"] = bench_spgemm_sharded(B, gen, np, torch, dist, rank, world)
            # top-level, where a parser of the line looks: aggregate SpGEMM rate of the sharded product, its single-GPU value measured in
            # this same run, and their ratio (the headline `value` above is a no-communication replica rate and scales by construction)
            sg = out["spgemm_sharded"]
            out["spgemm_sharded_gflops"] = sg.get("gflops")
            out["spgemm_single_gpu_gflops"] = sg.get("single_gpu_gflops")
            out["spgemm_sharded_speedup"] = sg.get("speedup_vs_single_gpu")
            out["spgemm_owner_keeps_gflops"] = sg.get("owner_keeps", {}).get("gflops")
            out["spgemm_owner_keeps_speedup"] = sg.get("owner_keeps", {}).get("speedup_vs_single_gpu")
            out["spgemm_exchange_hidden_frac"] = sg.get("exchange_hidden_frac")
        except Exception as e:  # noqa: BLE001
            out["spgemm_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if use_dist:
        try:
            out["spmv_sharded"] = bench_spmv_sharded(B, np, torch, dist, rank, world, mats0, x0, y0)
        except Exception as e:  # noqa: BLE001
            out["spmv_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}

    note("SpGEMM done")
    # ---------------- vendor comparison column (rocSPARSE CSR on the same matrices; reporting only) ----------------
    if rank == 0 and not use_dist and not args.skip_vendor:
        out["vendor"] = vendor_in_child(args)
    note("vendor column done")
    # ---------------- CPU baseline (cusp::multiply restatement) on rank 0, N = 1 only ----------------
    if rank == 0 and not use_dist and not args.skip_cpu:
        out["cpu_baseline"] = cpu_baseline(wl, eff_bytes, args)

    note("CPU baseline done")
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_spmv_structures(B, gen, np, headline):
    """the three structures of VERDICT r1 item 3 in one line: the headline R-MAT (hyper-sparse tiles), banded half-bandwidth 8
    (half-filled tiles) and banded half-bandwidth 32 (mostly full tiles).  Same kernel entry point, HBM-resident rotation,
    per-launch time from HIP events."""
    res = [{"structure": "headline (see config.workload)", "values_per_tile": round(headline.info["nnz"] / max(1, headline.info["block_num"]), 2)}]
    cases = [("banded(1000000, half_bw=8)", lambda: gen.banded(1000000, 8)), ("banded(500000, half_bw=32)", lambda: gen.banded(500000, 32)),
             ("fem_like(100^3 grid, 7pt): FEM-like, ~1 M rows", lambda: gen.fem_like(100, "7pt")), ("cage_like(1000000, 15.6/row)", lambda: gen.cage_like(1000000, 15.6))]
    for name, mk in cases:
        n, _, r, c, v = mk()
        first = B.BmSpMatrix.from_coo(n, n, r, c, v)
        del r, c, v
        S = SpmvSet(B, np, first, x_kind="cusp")
        S.timed(10, 0)
        ms = S.timed(60, 0)
        ach = S.alg_bytes / (ms * 1e-3) / 1e9
        res.append({"structure": name, "rows": S.info["num_rows"], "nnz": S.info["nnz"], "blocks": S.info["block_num"],
                    "values_per_tile": round(S.info["nnz"] / max(1, S.info["block_num"]), 2), "hbm_resident_copies_rotated": S.copies,
                    "kernel": S.launch["kernel"], "prepare_ms": round(S.prepare_ms, 4),
                    "avg_launch_ms": round(ms, 5), "algorithmic_bytes_per_launch": S.alg_bytes, "format_bytes": S.format_bytes,
                    "effective_GBs": round(S.eff_bytes / (ms * 1e-3) / 1e9, 1),
                    "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}})
        del S, first
        B.check(B.lib().bmsp_trim_pool())
    return res


SPGEMM_CASES = [
    # tag, workload name, generator, dtype name, tc_version, SuiteSparse file it stands in for, profiles/ tag
    ("fem", "2cubes_sphere-like fem_like(47^3 grid, poisson27pt, windowed random renumbering)", lambda g: g.fem_like(47, "27pt"), "F32", 5, "2cubes_sphere.mtx", "spgemm_fem_like"),
    ("cage", "cage12-like local+random(130228, 15.6/row)", lambda g: g.cage_like(130228, 15.6), "F16", 4, "cage12.mtx", "spgemm_cage_like"),
    ("ceiling", "dense-tile ceiling: banded(147456, half_bw=256), 65 full 8x8 tiles per block-row, 7.8e7 tasks, 32.8 per C tile", lambda g: g.banded(147456, 256), "F16", 4, None, "spgemm_ceiling"),
    # power-law operands (hub block-rows: C is a fifth full at tile granularity): the column-window passes (rowwindow.hip)
    ("rmat16", "rmat(scale=16, edge_factor=8)+I: 1.46e8 candidate pairs, 6.4e7 tasks, 4.5 per C tile", lambda g: g.rmat(16, 8), "F16", 4, None, "spgemm_rmat16"),
    # configs[4]'s product on ONE GPU (the N = 1 value of the sharded figure): 2.3e9 candidate pairs, run in block-row panels
    ("rmat22", "rmat(scale=22, edge_factor=1)+I (configs[4] on one GPU): 2.29e9 candidate pairs, 6.9e8 tasks, 4.8e8 C tiles, 1.44 per C tile", lambda g: g.rmat(22, 1), "F16", 4, None, "spgemm_rmat22"),
    # the reference's DEFAULT configuration (spgemm_run_batch.sh: tc_version 5 on the half-precision build): fp16 operands, V15 numerics
    ("fem_h5", "2cubes_sphere-like fem_like(47^3 grid, poisson27pt, windowed random renumbering)", lambda g: g.fem_like(47, "27pt"), "F16", 5, "2cubes_sphere.mtx", "spgemm_fem_like_h5"),
]
NO_BASELINE_TAGS = ("ceiling", "rmat16", "rmat22", "fem_h5")  # cases without a vendor / CPU leg (fem_h5: the fp32 legs of `fem` are the same product)
MAC_VARIANT = {0: "default kernel of the tc_version", 1: "block_mac_mfma32_kernel (K = 32, LDS-staged)", 2: "block_mac_direct_kernel (K = 32, lines per task)",
               3: "block_mac_strip_kernel (K = 32, two block-rows per wave, operand reuse)", 4: "block_mac_f32_mfma_kernel (v_mfma_f32_16x16x4_f32, V15 chain, lane-ordered tile copies)",
               5: "block_mac_rowsparse_kernel (V15 chain over the products of STORED values only, row-wise over CSR copies, accumulators per C value in LDS)"}


def stage_bytes(st, sort_bits):
    """compulsory bytes of the symbolic stages for THIS library's layouts (DESIGN.md section 4 states them next to SURVEY 8(d)'s figures
    for the reference's 16-byte tasks): per candidate pair / surviving task / C block.  Keyed by the stage the time is charged to."""
    cand, surv, cb = st["task_list_size"], st["surviving_tasks"], st["c_blocks"]
    if st["sort_path"] == 2 and st.get("sort_long") == 1:
        # row-merge, strip mode (rowmerge_symbolic_kernel + emit): key + bitmap of B's tile per candidate pair; C's column + bitmap to scratch
        # (12 B), read back by the emit pass, C's key + bitmap + offset written (24 B).  T_3 holds all of it (T_9 is an allocation)
        return {"T_3": 16 * cand + 48 * cb}
    if st["sort_path"] == 2:
        # row-merge, task-list mode (build + copy): 16 B per candidate pair; per surviving pair the parked {pair, product, order} written and
        # read back (2 x 18 B), its task placed (8 B), its product ORed (8 B), the task copied to the list (8 + 8 B); per C tile column +
        # first task + bitmap to scratch (16 B), read by the copy pass (16 B), key + bitmap + offset + first task written (28 B)
        return {"T_3": 16 * cand + 68 * surv + 60 * cb}
    if st["sort_path"] == 3:
        # column windows (rowwindow.hip): one 16-byte record of B's tile per candidate pair in the count pass, its 8-byte half in each half
        # of a round of the fill pass (the first steps of a wave stay in registers: counted once), the task written (8 B); per C tile
        # column + count + bitmap to scratch (16 B), read back (16 B), key + bitmap + offset + first task written (28 B)
        return {"T_3": 24 * cand + 8 * surv + 60 * cb}
    passes = -(-sort_bits // 9)
    return {
        "T_3": 16 * cand,                              # count pass: the two bitmaps of every candidate pair
        "T_4": 16 * cand + 16 * surv,                  # write pass: bitmaps again, key + task written per survivor
        "T_5": (40 * surv) if st["sort_path"] else (8 * surv * passes + 32 * surv * passes),  # segmented: keys r+w, permutation w+r, payload r+w; radix: histogram + scatter per pass
        "T_6": 16 * surv + 12 * cb,                    # run-length encode: keys read by both scan phases; C key + task_begin written
        "T_9": 32 * surv + 32 * cb,                    # key + task + two bitmap gathers per task; C bitmap written, read by the popcount scan, offset written
    }


def stage_fracs(sb, t_us, stage_idx):
    """(GB/s, fraction of the HBM peak) per stage with a byte formula.  A fraction above 1 means the formula charges bytes the stage does
    not move any more (T_9 = 2.54) -- refused here instead of being printed."""
    gbs, frac = {}, {}
    for k, i in stage_idx:
        if k not in sb or t_us[i] < 1.0:
            continue
        gbs[k] = round(sb[k] / t_us[i] / 1e3, 1)
        frac[k] = round(sb[k] / t_us[i] / 1e3 / HBM_PEAK_GBS, 4)
        assert frac[k] <= 1.0, "stage %s: %d bytes in %.1f us is above the HBM peak -- the byte formula is wrong" % (k, sb[k], t_us[i])
    return gbs, frac


def bench_spgemm(B, gen, np, args):
    """A x A on the synthetic stand-ins of 2cubes_sphere (fp32, vector-ALU block-MAC) and cage12 (fp16, MFMA block-MAC), plus a
    dense-tile input (every tile full, ~9 tasks per C block) that separates the block-MAC kernel's own ceiling from the inputs'
    sparsity."""
    res = []
    for tag, name, mk, dtn, tc, fname, ptag in SPGEMM_CASES:
        if args.only_spgemm and args.only_spgemm not in tag:
            continue
        if tag == "rmat22" and args.skip_rmat22 and not args.only_spgemm:
            continue
        dtype = getattr(B, dtn)
        cli = None
        path = os.path.join(args.mtx_dir, fname) if (args.mtx_dir and fname) else ""
        if path and os.path.exists(path):
            A = B.BmSpMatrix.from_mtx(path, False, dtype)
            At = B.BmSpMatrix.from_mtx(path, True, dtype)
            assert_suitesparse_dims(path, A.info())
            name = fname + " (SuiteSparse file)"
        else:
            n, _, r, c, v = mk(gen)
            A = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype)
            At = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=dtype)
            cli = None
            if tag in ("fem", "cage") and not args.skip_cli:
                try:
                    cli = cli_execution(np, (n, n, r, c, v), 5 if tag == "fem" else 4)
                except Exception as e:  # an extra: never lose the line over it
                    cli = {"error": "%s: %s" % (type(e).__name__, e)}
            del r, c, v
        info = A.info()
        # operand preparation (block records, dense tile copies, block-row pointers and row maxima of both operands): built once per
        # matrix and cached; the reference's `bmSparse execution` bracket starts from two bmSpMatrix operands (src/bmSparse_SPGEMM.cu:
        # 1274-1280), so this is reported beside every total instead of being hidden in a warm-up call
        B.synchronize()
        t0 = time.perf_counter()
        A.prepare(2); At.prepare(2)
        B.synchronize()
        prepare_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        Cm, st_first = B.spgemm(A, At, mode=B.SORT_AUTO, tc_version=tc)
        B.synchronize()
        first_call_ms = (time.perf_counter() - t0) * 1e3
        del Cm
        runs = []
        n_runs = 5 if tag == "ceiling" else (3 if tag == "rmat22" else 8)
        for it in range(n_runs):  # one more call warms the pool; median of the rest
            t0 = time.perf_counter()
            Cm, st = B.spgemm(A, At, mode=B.SORT_AUTO, tc_version=tc)
            st["wall_ms"] = (time.perf_counter() - t0) * 1e3
            if it >= 1:
                runs.append(st)
            del Cm
        runs.sort(key=lambda q: q["t_us"][0])
        best = runs[len(runs) // 2]
        # the numeric half alone on a C that already has the structure (bmsp_spgemm_numeric: new values on an old pattern)
        numeric_ms = None
        try:
            if tag == "rmat22":
                raise RuntimeError("not measured on this case (a paneled product keeps no task list)")
            Cs, _ = B.spgemm_symbolic(A, At, mode=B.SORT_AUTO, tc_version=tc)
            nm = []
            for it in range(4):
                B.synchronize()
                t0 = time.perf_counter()
                stn = B.spgemm_numeric(A, At, Cs, tc_version=tc)
                B.synchronize()
                if it:
                    nm.append(((time.perf_counter() - t0) * 1e3, stn["mac_variant"]))
            nm.sort()
            numeric_ms = {"wall_ms": round(nm[len(nm) // 2][0], 3), "kernel": MAC_VARIANT.get(nm[0][1], "?"),
                          "note": "bmsp_spgemm_numeric into an existing structure; the strip kernel alone where it applies, else the whole product + a value copy"}
            del Cs
        except Exception as e:  # an extra: never lose the line over it
            numeric_ms = {"error": "%s: %s" % (type(e).__name__, e)}
        P = scalar_products(np, A)
        t_total = best["t_us"][0] * 1e-6
        t_mac = best["t_us"][7] * 1e-6
        f_mac = 1024.0 * best["surviving_tasks"]
        peak = MFMA_F16_PEAK_TFLOPS if dtype == B.F16 else FP32_PEAK_TFLOPS
        # the sources that define the profiled block-MAC kernels (the dispatch in spgemm.hip and the symbolic passes do not change them)
        kernel_files = ["bmsparse-spgemm-spmv_amd/csrc/blockmac_strip.hip", "bmsparse-spgemm-spmv_amd/csrc/blockmac32.hip", "bmsparse-spgemm-spmv_amd/csrc/blockmac_f32.hip",
                        "bmsparse-spgemm-spmv_amd/csrc/mac_common.hip.h"]
        traffic, traffic_src = profile_value("r*_%s_traffic.json" % ptag, "traffic_bytes_per_launch", kernel_files)
        roof = {"bound": "mfma" if (dtype == B.F16 and tc != 5) else "fp32 matrix / vector rate", "kernel": MAC_VARIANT.get(best.get("mac_variant", 0), "?"),
                "achieved": round(f_mac / t_mac / 1e12, 3),
                "peak": peak, "unit": "TFLOP/s", "frac": round(f_mac / t_mac / 1e12 / peak, 5),
                "traffic": traffic, "traffic_source": traffic_src}
        if best.get("mac_variant") == 5:
            # the row-sparse kernel multiplies the stored values only: its arithmetic is 2 flop per scalar product, and what bounds it is the
            # bookkeeping of a product (table look-up, rank in C's bitmap, one LDS read-modify-write), not the multiply-add.  Its roofline is
            # the memory one: the bytes of its own layout -- CSR copies of A and B (8 B per value + row pointers), C's keys, bitmaps, offsets
            # and values -- against the HBM peak; the products per second are given beside it.
            rs_bytes = 16 * info["nnz"] + 8 * (info["num_rows"] + 1) + 24 * best["c_blocks"] + 4 * best["c_nnz"]
            traffic, traffic_src = profile_value("r*_%s_traffic.json" % ptag, "traffic_bytes_per_launch", ["bmsparse-spgemm-spmv_amd/csrc/blockmac_rowsparse.hip"])
            roof = {"bound": "hbm", "kernel": MAC_VARIANT[5], "achieved": round(rs_bytes / t_mac / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(rs_bytes / t_mac / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": int(rs_bytes), "scalar_products_per_s": round(P / t_mac, 0),
                    "dense_equivalent_TFLOPs": round(f_mac / t_mac / 1e12, 2),
                    "note": "dense_equivalent = the 1024 flop per task the tile-by-tile kernels perform, for comparison with the other entries"}
        mu, mu_src = profile_value("r*_%s_mfma.json" % ptag, "mfma_util", kernel_files)
        if mu is not None:
            roof["mfma_util"] = mu  # SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), from the committed PMC pass
            roof["mfma_util_source"] = mu_src
        sort_bits = max(1, int(np.ceil(np.log2(max(2, (info["num_rows"] + 7) // 8))))) + max(1, int(np.ceil(np.log2(max(2, (info["num_cols"] + 7) // 8)))))
        sb = stage_bytes(best, sort_bits)
        stage_idx = (("T_1", 1), ("T_2", 2), ("T_3", 3), ("T_4", 4), ("T_5", 5), ("T_6", 6), ("T_9", 9), ("T_7", 7))
        stage_gbs, stage_frac = stage_fracs(sb, best["t_us"], stage_idx)
        res.append({"workload": "bmSparse SpGEMM A*A %s, %s" % (("fp16 MFMA block-MAC" if tc != 5 else "fp16 operands, V15 numerics (tc_version 5: the reference's default configuration)") if dtype == B.F16 else "fp32", name),
                    "rows": info["num_rows"], "nnz": info["nnz"], "blocks": info["block_num"],
                    "values_per_tile": round(info["nnz"] / max(1, info["block_num"]), 2),
                    "tasks": best["task_list_size"], "surviving_tasks": best["surviving_tasks"], "c_blocks": best["c_blocks"],
                    "tasks_per_c_block": round(best["surviving_tasks"] / max(1, best["c_blocks"]), 2),
                    "c_nnz": best["c_nnz"], "scalar_products": int(P), "total_ms": round(t_total * 1e3, 3), "timing": "median of %d products (device time of the whole call)" % len(runs),
                    "wall_ms": round(best["wall_ms"], 3), "prepare_ms": round(prepare_ms, 3), "first_call_ms": round(first_call_ms, 3),
                    "total_with_prepare_ms": round(t_total * 1e3 + prepare_ms, 3), "numeric_only": numeric_ms, "cli": cli,
                    "prepare_note": "prepare = bmsp_matrix_prepare of both operands (host wall time); first_call = the first product after it (cold pool)",
                    "gflops": round(2.0 * P / t_total / 1e9, 2),
                    "gflops_with_prepare": round(2.0 * P / (t_total + prepare_ms * 1e-3) / 1e9, 2),
                    "stage_us": {k: round(best["t_us"][i], 1) for k, i in stage_idx},
                    "stage_GBs": stage_gbs, "stage_frac_of_hbm_peak": stage_frac,
                    "sort_path": {0: "global radix", 1: "segmented" + {0: "", 1: " (long block-rows: pieces + merge passes)",
                                                                       2: " (long block-rows: counting passes on the column bits)"}[best.get("sort_long", 0)], 2: "none (row-merge, %s: C's structure formed per block-row in LDS)" % {1: "strip mode", 2: "task-list mode"}.get(best.get("sort_long", 0), "?"),
                                  3: "none (column windows: C's structure formed per block-row and window of block columns in dense LDS tables)"}[best["sort_path"]],
                    "roofline": roof})
        del A, At
        B.check(B.lib().bmsp_trim_pool())
    return res


def cli_execution(np, coo, tc_version=5):
    """what a user of the drop-in executable sees: the workload written as MatrixMarket, `bmsparse_spgemm_float folder A A
    0 <tc_version> 0` run on it -- fp16 operands, as the reference's main builds them (src/bmSparse_SPGEMM.cu:1261-1262) -- and its own
    `bmSparse execution` bracket (ONE product in a fresh process, cold pool, :1274-1280) parsed from stdout."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(REPO, "bmsparse_spgemm_float")
    if not os.path.exists(exe):
        return {"error": "bmsparse_spgemm_float not built"}
    n, _, r, c, v = coo
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "A.mtx")
        import pandas as pd
        with open(path, "w") as f:
            f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, r.size))
        pd.DataFrame({"r": np.asarray(r, dtype=np.int64) + 1, "c": np.asarray(c, dtype=np.int64) + 1, "v": np.asarray(v, dtype=np.float64)}).to_csv(
            path, sep=" ", header=False, index=False, mode="a", float_format="%.9g")
        out = subprocess.run([exe, d, "A", "A", "0", str(tc_version), "0"], capture_output=True, text=True, timeout=120)
    if out.returncode != 0:
        return {"error": (out.stderr or out.stdout)[-200:]}
    us = {k: int(m.group(1)) for k, pat in (("load_us", r"Loading matrices from disk BMSP: (\d+)"), ("execution_us", r"bmSparse execution: (\d+)"),
                                            ("c_blocks", r"C blocks: (\d+)"), ("c_nnz", r"C nnz: (\d+)")) for m in [re.search(pat, out.stdout)] if m}
    us["command"] = "bmsparse_spgemm_float <dir> A A 0 %d 0 (fp16 operands, ONE product in a fresh process)" % tc_version
    return us


def vendor_column(np, gen, wl, eff_bytes, args):
    """rocSPARSE CSR SpMV / SpGEMM (fp32) on the bench matrices, cache-warm like `warm_ms_per_step`."""
    import ctypes as C
    import scipy.sparse as sp
    so = os.path.join(REPO, "vendor_compare", "librocsparse_ref.so")
    if not os.path.exists(so):
        return {"error": "vendor_compare/librocsparse_ref.so not built"}
    V = C.CDLL(so)
    vp, i64, dp = C.c_void_p, C.c_int64, C.POINTER(C.c_double)
    V.vendor_csr_spmv.argtypes = [C.c_int, C.c_int, i64, vp, vp, vp, vp, vp, C.c_int, C.c_int, dp, dp]
    V.vendor_csr_spgemm.argtypes = [C.c_int, C.c_int, C.c_int, i64, vp, vp, vp, i64, vp, vp, vp, C.c_int, dp, dp, C.POINTER(i64), dp]

    def csr_of(coo):
        n, _, r, c, v = coo
        m = sp.coo_matrix((v.astype(np.float32), (r, c)), shape=(n, n)).tocsr()
        m.sum_duplicates(); m.sort_indices()
        return m, m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)

    res = {"library": "rocSPARSE (ROCm 7.2), CSR fp32 int32 indices; same matrices, cache-warm timing"}
    note("vendor: library loaded")
    if "coo" in wl:
        m, ptr, col, val = csr_of(wl["coo"])
        note("vendor: CSR of the SpMV workload built")
        x = np.ones(m.shape[1], np.float32); y = np.zeros(m.shape[0], np.float32)
        best = None
        for alg, name in ((1, "adaptive"), (3, "lrb")):
            ms, pre = C.c_double(), C.c_double()
            rc = V.vendor_csr_spmv(m.shape[0], m.shape[1], m.nnz, ptr.ctypes.data, col.ctypes.data, val.ctypes.data, x.ctypes.data, y.ctypes.data,
                                   alg, 200, C.byref(ms), C.byref(pre))
            note("vendor: spmv alg %s done" % name)
            if rc == 0 and (best is None or ms.value < best[1]):
                best = (name, ms.value, pre.value)
        if best:
            res["spmv"] = {"alg": best[0], "ms": round(best[1], 5), "preprocess_ms": round(best[2], 3), "protocol": "cache-warm (one matrix, back to back)",
                           "effective_GBs": round(eff_bytes / (best[1] * 1e-3) / 1e9, 1), "y_checksum": float(y.sum())}
            # the headline protocol: rotated over device-resident copies (> 512 MiB in total), so every sweep streams from HBM
            if hasattr(V, "vendor_csr_spmv_rotated"):
                V.vendor_csr_spmv_rotated.argtypes = [C.c_int, C.c_int, i64, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, dp]
                csr_b = 4 * (m.shape[0] + 1) + 8 * m.nnz
                copies = max(2, int(np.ceil(512 * 2 ** 20 / csr_b)) + 1)
                ms = C.c_double()
                alg = {"adaptive": 1, "lrb": 3}[best[0]]
                rc = V.vendor_csr_spmv_rotated(m.shape[0], m.shape[1], m.nnz, ptr.ctypes.data, col.ctypes.data, val.ctypes.data, x.ctypes.data, alg, 200, copies, C.byref(ms))
                if rc == 0:
                    res["spmv_hbm_resident"] = {"alg": best[0], "ms": round(ms.value, 5), "copies_rotated": copies, "protocol": "rotated over HBM-resident copies (the protocol of `value`)",
                                                "effective_GBs": round(eff_bytes / (ms.value * 1e-3) / 1e9, 1)}
                note("vendor: rotated spmv done")
    if not args.skip_spgemm:
        res["spgemm"] = []
        for tag, name, mk, dtn, tc, fname, ptag in SPGEMM_CASES:
            if tag in NO_BASELINE_TAGS or (args.only_spgemm and args.only_spgemm not in tag):
                continue
            m, ptr, col, val = csr_of(mk(gen))
            note("vendor: CSR of %s built" % tag)
            ms, first, nz, sm = C.c_double(), C.c_double(), i64(), C.c_double()
            rc = V.vendor_csr_spgemm(m.shape[0], m.shape[1], m.shape[1], m.nnz, ptr.ctypes.data, col.ctypes.data, val.ctypes.data, m.nnz, ptr.ctypes.data,
                                     col.ctypes.data, val.ctypes.data, 5, C.byref(ms), C.byref(first), C.byref(nz), C.byref(sm))
            res["spgemm"].append({"workload": "CSR SpGEMM A*A fp32, " + name, "rc": rc, "ms": round(ms.value, 3), "first_call_ms": round(first.value, 3),
                                  "c_nnz": nz.value})
    return res


def scalar_products(np, A):
    """P = number of scalar products a_ik * b_kj of A*A = sum over stored a_ik of nnz(row k of A)."""
    r, c, _ = A.to_coo()
    rownnz = np.bincount(r, minlength=A.num_rows)
    return int(rownnz[c].sum())


def bench_spmv_sharded(B, np, torch, dist, rank, world, A, x, y_whole):
    """SURVEY 8(e), SpMV row: ONE matrix cut into nnz-balanced block-row panels, x replicated, y slices exchanged in place over RCCL
    through the C ABI (bmsp_spmv_sharded); strong scaling; the headline `value` above is the weak-scaling replica rate."""
    comm = B.Comm.from_torch(dist, torch)
    y, sh = B.spmv_sharded(comm, A, x)  # warm-up (builds the panel view and its plan), and the check below
    same = bool(np.array_equal(y.to_host(), y_whole.to_host()))
    reps = 20
    B.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    ex_us = 0.0
    for _ in range(reps):
        y, sh = B.spmv_sharded(comm, A, x, y)
        ex_us += sh["exchange_us"]
    B.synchronize(); dist.barrier()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    comm.free()
    return {"workload": "one matrix, row panels balanced by nnz, y slices broadcast in place (bmsp_spmv_sharded, RCCL)", "scaling": "strong", "n_gpus": world,
            "ms_per_product": round(ms, 4), "exchange_ms": round(ex_us / reps * 1e-3, 4), "exchange_bytes": sh["exchange_bytes"],
            "matches_single_gpu_sweep": same}


def bench_spgemm_sharded(B, gen, np, torch, dist, rank, world):
    """configs[4]: row-panel-sharded SpGEMM on R-MAT scale 22 with the allgatherv of the C panels over RCCL (bmsp_spgemm_sharded: panels
    broadcast straight into their final slices); fixed total work -> strong scaling.  Edge factor 1 (+ identity): 2.3 G candidate block
    pairs, 0.69 G surviving tasks, 0.48 G C blocks -- the largest scale-22 instance whose single panel (N = 1) stays inside the 32-bit
    candidate range, so that N = 1, 2, 4, 8 run the same product (edge factor 2 has 7.9 G candidates)."""
    scale = int(os.environ.get("BMSP_SHARD_SCALE", "22"))
    ef = float(os.environ.get("BMSP_SHARD_EF", "1"))
    n, _, r, c, v = gen.rmat(scale, ef)
    A = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=B.F16).prepare(2)
    Bt = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=B.F16).prepare(2)
    comm = B.Comm.from_torch(dist, torch)

    def timed(gather):
        best = None
        for it in range(3):
            B.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            Cm, st, sh = B.spgemm_sharded(comm, A, Bt, tc_version=4, gather=gather)
            B.synchronize(); dist.barrier()
            dt = time.perf_counter() - t0
            t = torch.tensor([dt, st["t_us"][0] * 1e-6], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            blocks = torch.tensor([Cm.info()["block_num"], Cm.info()["nnz"]], dtype=torch.int64, device="cuda")
            if not gather:
                dist.all_reduce(blocks)  # owner keeps: every rank holds its panel; the product's size is the sum
            if it and (best is None or float(t[0].item()) < best[0]):
                best = (float(t[0].item()), sh, {"block_num": int(blocks[0].item()), "nnz": int(blocks[1].item())}, float(t[1].item()))
            del Cm
        return best

    best = timed(True)       # every rank ends with the whole C: rounds of panels, a round's broadcasts behind the next round's products
    keeps = timed(False)     # owner keeps: no exchange ("skip the gather and report it separately")
    comm.free()
    # the same product on ONE GPU (rank 0 alone, the other ranks wait at the barrier): the N = 1 value of this strong-scaling figure
    single = None
    if rank == 0:
        for it in range(2):
            B.synchronize()
            t0 = time.perf_counter()
            Cm, _ = B.spgemm(A, Bt, tc_version=4)
            B.synchronize()
            single = time.perf_counter() - t0
            del Cm
    dist.barrier()
    P = scalar_products(np, A) if rank == 0 else 0
    sh = best[1]
    return {"workload": "row-panel-sharded SpGEMM fp16 MFMA, rmat(scale=%d, ef=%g)+I" % (scale, ef), "scaling": "strong", "n_gpus": world,
            "single_gpu_ms": round(single * 1e3, 3) if single else None, "single_gpu_gflops": round(2.0 * P / single / 1e9, 2) if single else None,
            "speedup_vs_single_gpu": round(single / best[0], 3) if single else None,
            "total_ms": round(best[0] * 1e3, 3), "slowest_rank_products_ms": round(best[3] * 1e3, 3), "gflops": round(2.0 * P / best[0] / 1e9, 2),
            "c_blocks": best[2]["block_num"], "c_nnz": best[2]["nnz"], "allgatherv_bytes": sh["exchange_bytes"], "rounds": sh["rounds"],
            "allgatherv_ms": round(sh["exchange_us"] * 1e-3, 3), "allgatherv_exposed_ms": round(sh["exchange_exposed_us"] * 1e-3, 3),
            "exchange_hidden_frac": round(sh["exchange_hidden_frac"], 3),
            "allgatherv_GBs_per_rank": round(sh["exchange_bytes"] * (world - 1) / max(world, 1) / max(sh["exchange_us"], 1e-3) / 1e3, 1),
            "panel_tasks": sh["panel_tasks"],
            "owner_keeps": {"total_ms": round(keeps[0] * 1e3, 3), "gflops": round(2.0 * P / keeps[0] / 1e9, 2), "c_blocks": keeps[2]["block_num"], "c_nnz": keeps[2]["nnz"],
                            "speedup_vs_single_gpu": round(single / keeps[0], 3) if single else None,
                            "note": "every rank keeps its panel of C (bmsp_spgemm_sharded_ex, gather = 0): no exchange"}}


def host_threads(O):
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:  # cgroup cpu quota of the box (a 1-GPU box gets a share of the host's cores)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = max(1, min(avail, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(O.max_threads(), avail, 64))


def time_cpu(fn, budget_s, max_iters=2000, min_iters=3):
    fn()
    t0 = time.perf_counter()
    fn()
    one = max(1e-6, time.perf_counter() - t0)
    iters = int(min(max_iters, max(min_iters, budget_s / one)))
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters, iters


def cpu_baseline(wl, eff_bytes, args):
    """oracle (`port` of cusp::multiply) on this box's host cores, bounded to ~cpu-seconds of work in total:
      * CSR SpMV (csr_spmv.h:56-73 + the OpenMP row-parallel variant) on the headline matrix and x  -> value
      * CSR Gustavson SpGEMM (csr_spgemm.h:39-157, omp/.../csr_spgemm.h:40-157), A*A on both SpGEMM workloads -> `spgemm`
      * configs[0]: the host CSR SpMV on a cant-sized FEM stand-in (62 451 rows, ~4.0 M nnz), no GPU involved -> `cant_like`"""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import numpy as np
    import oracle as O
    from pybmsp import gen
    th = host_threads(O)
    budget = args.cpu_seconds
    if "path" in wl:
        coo = O.mtx_read(wl["path"], strict=True)
    else:
        n, _, r, c, v = wl["coo"]
        coo = O.Coo(n, n, r, c, v)
    A = O.csr_from_coo(coo)
    x = gen.spmv_x(A.num_cols, "ones")
    res = {}
    for name, t in (("omp", th), ("seq", 1)):
        dt, iters = time_cpu(lambda: O.csr_spmv(A, x, t), budget * 0.12, min_iters=10)
        res[name] = (eff_bytes / dt / 1e9, iters, dt)
    used = th
    if res["seq"][0] > res["omp"][0]:  # oversubscribed box: the single-thread figure is the better baseline
        res["omp"], used = res["seq"], 1
    out = {"value": round(res["omp"][0], 2), "unit": "GB/s", "cores": used, "kind": "port",
           "sample": "CSR SpMV (cusp::multiply restatement, OpenMP row-parallel) on the same matrix and x, %d iterations, %.2f ms each; "
                     "single-thread: %.2f GB/s over %d iterations" % (res["omp"][1], res["omp"][2] * 1e3, res["seq"][0], res["seq"][1])}
    # configs[0]: cusp::multiply SpMV on cant (62 451 rows, 4 007 383 nnz), host CPU path
    cn, _, cr, cc, cv = gen.banded(62451, 32)
    Ac = O.csr_from_coo(O.Coo(cn, cn, cr, cc, cv))
    xc = gen.spmv_x(cn, "cusp")
    cb = csr_bytes(cn, Ac.nnz)
    legs = {}
    for name, t in (("omp", th), ("seq", 1)):
        dt, iters = time_cpu(lambda: O.csr_spmv(Ac, xc, t), budget * 0.08, min_iters=10)
        legs[name] = {"GBs": round(cb / dt / 1e9, 2), "gflops": round(2.0 * Ac.nnz / dt / 1e9, 2), "ms": round(dt * 1e3, 4), "iterations": iters,
                      "threads": t}
    # the PRODUCT's host path for the same class (CSRMatrix::multiply_host -> bmsp_csr_spmv_host; no GPU call), next to the oracle's port
    try:
        import pybmsp as B
        Pc = B.CSRMatrix.from_arrays(cn, cn, Ac.row_offsets, Ac.cols, Ac.vals)
        dt, iters = time_cpu(lambda: Pc.spmv_host(xc, th), budget * 0.06, min_iters=10)
        legs["product_host_path"] = {"GBs": round(cb / dt / 1e9, 2), "gflops": round(2.0 * Ac.nnz / dt / 1e9, 2), "ms": round(dt * 1e3, 4), "iterations": iters,
                                     "threads": th, "entry": "bmsp_csr_spmv_host"}
    except Exception as e:
        legs["product_host_path"] = {"error": str(e)[:120]}
    out["cant_like"] = {"workload": "configs[0] stand-in: cusp::multiply CSR SpMV, host only, banded(62451, half_bw=32): %d nnz (cant: 62 451 rows, "
                                    "4 007 383 nnz)" % Ac.nnz, **legs}
    if not args.skip_spgemm:
        out["spgemm"] = []
        for tag, name, mk, dtn, tc, fname, ptag in SPGEMM_CASES:
            if tag in NO_BASELINE_TAGS or (args.only_spgemm and args.only_spgemm not in tag):
                continue
            gn, _, gr, gc, gv = mk(gen)
            G = O.csr_from_coo(O.Coo(gn, gn, gr, gc, gv))
            legs = {}
            prods = 0
            for nm, t in (("omp", th), ("seq", 1)):
                box = {}

                def run():
                    box["p"] = O.csr_spgemm(G, G, t)[1]
                dt, iters = time_cpu(run, budget * 0.15, max_iters=50, min_iters=2)
                prods = box["p"]
                legs[nm] = {"gflops": round(2.0 * prods / dt / 1e9, 3), "ms": round(dt * 1e3, 2), "iterations": iters, "threads": t}
            out["spgemm"].append({"workload": "cusp::multiply(csr,csr,csr) restatement (Gustavson), A*A fp32, " + name, "scalar_products": prods, **legs})
    return out


if __name__ == "__main__":
    main()
