#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native bmSparse engine.

metric (BASELINE.json): "SpMV effective GB/s + SpGEMM GFLOP/s on SuiteSparse; % HBM/MFMA roofline".
The ONE JSON line carries the SpMV figure as `value` (configs[1]: bmSparse SpMV fp32 on webbase-1M, 1 GPU) and the
SpGEMM figures (configs[2], configs[3]) in `spgemm`.

A step = one SpMV sweep u = A*v over one resident matrix.  SuiteSparse files cannot be downloaded here; unless
`--mtx-dir` holds webbase-1M.mtx the workload is the synthetic stand-in of SURVEY.md 8(d): R-MAT scale 20, edge factor
2, identity added (1 048 576 rows, ~3.13 M nnz -- webbase-1M has 1 000 005 rows, 3 105 536 nnz).  The matrix in bmSparse
form is ~70 MB and would sit in the 256 MiB Infinity Cache, so the timed loop rotates over enough device-resident copies
(> 512 MiB in total) that every sweep streams from HBM; the cache-warm figure is reported beside it as
`warm_ms_per_step`.  Inputs are resident in HBM before the timed region starts.

N > 1 (one process per GPU, launched by torch.distributed.run): every rank sweeps its own copies (fixed work per GPU,
no data-path collective; weak scaling); value = bytes swept by all ranks / max-over-ranks time.  The row-panel-sharded
SpGEMM with its RCCL allgatherv (configs[4]) is timed in the same run and reported under `spgemm_sharded`.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "bmsparse-spgemm-spmv_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a copy kernel reaches
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA spec
FP32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mtx-dir", default=os.environ.get("BMSP_MTX_DIR", ""))
    ap.add_argument("--scale", type=int, default=20, help="R-MAT scale of the synthetic SpMV workload")
    ap.add_argument("--edge-factor", type=float, default=2.0)
    ap.add_argument("--spmv-matrix", default="", help="experiment override: banded:N:HB | rmat:SCALE:EF | cage:N")
    ap.add_argument("--batched", type=int, default=-1, help="-1 auto, 0 / 1 force the SpMV variant")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-spgemm", action="store_true")
    ap.add_argument("--skip-vendor", action="store_true", help="skip the rocSPARSE comparison column")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def csr_bytes(rows, nnz):
    """CUSP's bytes-per-SpMV convention for int32/fp32 CSR (cusp/performance/spmv/bytes_per_spmv.h:31-39)."""
    return 2 * 4 * rows + 4 * nnz + 2 * 4 * nnz + 2 * 4 * rows


def bmsp_spmv_bytes(info, itemsize=4):
    """algorithmic bytes of one bmSparse sweep (BASELINE.md 4 / SURVEY.md 8(d)), with the layout the kernel reads:
    key+bitmap+offset per block (24 B), values, the uint32 dense block-row pointer, x once, y once."""
    nbr = (info["num_rows"] + 7) // 8
    return 24 * info["block_num"] + itemsize * info["nnz"] + 4 * (nbr + 1) + itemsize * info["num_cols"] + itemsize * info["num_rows"]


def load_spmv_workload(args):
    from pybmsp import gen
    import numpy as np
    path = os.path.join(args.mtx_dir, "webbase-1M.mtx") if args.mtx_dir else ""
    if path and os.path.exists(path):
        return {"name": "webbase-1M (SuiteSparse file)", "path": path}
    if args.spmv_matrix:
        kind, *a = args.spmv_matrix.split(":")
        coo = {"banded": lambda: gen.banded(int(a[0]), int(a[1])), "rmat": lambda: gen.rmat(int(a[0]), float(a[1])),
               "cage": lambda: gen.cage_like(int(a[0]))}[kind]()
        return {"name": args.spmv_matrix + " (experiment)", "coo": coo}
    n, _, r, c, v = gen.rmat(args.scale, args.edge_factor, seed=1)
    return {"name": "rmat(scale=%d, edge_factor=%g)+I, stand-in for webbase-1M" % (args.scale, args.edge_factor),
            "coo": (n, n, r, c, v)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
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
    if "path" in wl:
        first = B.BmSpMatrix.from_mtx(wl["path"])
        mk = lambda: B.BmSpMatrix.from_mtx(wl["path"])
    else:
        n, _, r, c, v = wl["coo"]
        first = B.BmSpMatrix.from_coo(n, n, r, c, v)
        mk = lambda: B.BmSpMatrix.from_coo(n, n, r, c, v)
    info = first.info()
    alg_bytes = bmsp_spmv_bytes(info)
    eff_bytes = csr_bytes(info["num_rows"], info["nnz"])
    copies = max(2, int(np.ceil(512 * 2 ** 20 / alg_bytes)) + 1)
    mats = [first] + [mk() for _ in range(copies - 1)]
    x = B.DeviceArray.from_host(gen.spmv_x(info["num_cols"], "ones"))  # v = 1 (SPMV.cu:279-281)
    ys = [B.DeviceArray(info["num_rows"], np.float32) for _ in range(copies)]
    avg_blocks_per_row = info["block_num"] / max(1, (info["num_rows"] + 7) // 8)
    variant = args.batched if args.batched >= 0 else 0

    def step(i):
        k = i % copies
        B.check(L.bmsp_spmv(mats[k].h, x.ptr, ys[k].ptr, variant, None))

    L = B.lib()
    for i in range(args.warmup):
        step(i)
    sync(); barrier(); sync()
    e0, e1 = B.Event(), B.Event()
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        step(i)
    e1.record()
    sync(); barrier()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_ms(e1)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall * 1e3 / args.steps
    value = world * eff_bytes * args.steps / wall / 1e9  # every rank sweeps the same amount of its own data

    # cache-warm sweep (one copy only), for information
    for i in range(10):
        L.bmsp_spmv(mats[0].h, x.ptr, ys[0].ptr, variant, None)
    sync()
    w0, w1 = B.Event(), B.Event()
    w0.record()
    for i in range(args.steps):
        L.bmsp_spmv(mats[0].h, x.ptr, ys[0].ptr, variant, None)
    w1.record()
    warm_ms = w0.elapsed_ms(w1) / args.steps

    kern_ms = dev_ms / args.steps  # HIP events on the launch stream, over the timed region
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    # HBM traffic from the PMC counters (separate rocprofv3 passes, summarised under profiles/); only quoted for the workload
    # it was collected on
    traffic = None
    if "coo" in wl and not args.spmv_matrix and args.scale == 20 and args.edge_factor == 2.0:
        import glob
        for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_spmv_webbase_like*_traffic.json"))):
            traffic = json.load(open(f))["traffic_bytes_per_launch"]
    roofline = {"bound": "hbm", "kernel": "spmv_sweep_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(kern_ms, 5)}

    out = {"metric": "bmSparse SpMV fp32 effective GB/s (CSR-convention bytes / time); SpGEMM GFLOP/s under `spgemm`",
           "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic" if "coo" in wl else "suitesparse",
           "config": {"workload": "bmSparse SpMV fp32, " + wl["name"], "rows": info["num_rows"], "nnz": info["nnz"],
                      "blocks": info["block_num"], "x": "ones", "variant": ["sweep (default)", "batched", "row-group"][variant],
                      "hbm_resident_copies_rotated": copies, "effective_bytes_per_spmv": eff_bytes},
           "warm_ms_per_step": round(warm_ms, 5), "warm_effective_GBs": round(eff_bytes / (warm_ms * 1e-3) / 1e9, 1),
           "roofline": roofline}

    # ---------------- SpGEMM (configs[2], configs[3]) on rank 0 / single GPU ----------------
    if not args.skip_spgemm and not use_dist:
        out["spgemm"] = bench_spgemm(B, gen, np, args)
    if use_dist and not args.skip_spgemm:
        out["spgemm_sharded"] = bench_spgemm_sharded(B, gen, np, torch, dist, rank, world)
    if use_dist:
        out["spmv_sharded"] = bench_spmv_sharded(B, np, torch, dist, rank, world, mats[0], x, ys[0])

    # ---------------- vendor comparison column (rocSPARSE CSR on the same matrices; reporting only) ----------------
    if rank == 0 and not use_dist and not args.skip_vendor:
        try:
            out["vendor"] = vendor_column(np, gen, wl, eff_bytes, args)
        except Exception as e:  # the column is optional: never let it take the bench line down
            out["vendor"] = {"error": str(e)[:200]}

    # ---------------- CPU baseline (cusp::multiply restatement) on rank 0, N = 1 only ----------------
    if rank == 0 and not use_dist and not args.skip_cpu:
        out["cpu_baseline"] = cpu_baseline(wl, eff_bytes, args)

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_spgemm(B, gen, np, args):
    """A x A on the synthetic stand-ins of 2cubes_sphere (fp32, vector-ALU block-MAC) and cage12 (fp16, MFMA block-MAC)."""
    res = []
    cases = [("2cubes_sphere-like banded(101492, half_bw=8)", gen.banded(101492, 8), B.F32, 5, "2cubes_sphere.mtx"),
             ("cage12-like local+random(130228, 15.6/row)", gen.cage_like(130228, 15.6), B.F16, 4, "cage12.mtx")]
    for name, coo, dtype, tc, fname in cases:
        path = os.path.join(args.mtx_dir, fname) if args.mtx_dir else ""
        if path and os.path.exists(path):
            A = B.BmSpMatrix.from_mtx(path, False, dtype)
            At = B.BmSpMatrix.from_mtx(path, True, dtype)
            name = fname + " (SuiteSparse file)"
        else:
            n, _, r, c, v = coo
            A = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=dtype)
            At = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=dtype)
        runs = []
        for it in range(12):  # two calls warm the pool and the per-matrix caches; median of the next ten
            Cm, st = B.spgemm(A, At, mode=B.SORT_AUTO, tc_version=tc)
            if it >= 2:
                runs.append(st)
            del Cm
        runs.sort(key=lambda q: q["t_us"][0])
        best = runs[len(runs) // 2]
        info = A.info()
        P = scalar_products(np, A)
        t_total = best["t_us"][0] * 1e-6
        t_mac = best["t_us"][7] * 1e-6
        f_mac = 1024.0 * best["surviving_tasks"]
        peak = MFMA_F16_PEAK_TFLOPS if dtype == B.F16 else FP32_PEAK_TFLOPS
        res.append({"workload": "bmSparse SpGEMM A*A %s, %s" % ("fp16 MFMA block-MAC" if dtype == B.F16 else "fp32", name),
                    "rows": info["num_rows"], "nnz": info["nnz"], "blocks": info["block_num"],
                    "tasks": best["task_list_size"], "surviving_tasks": best["surviving_tasks"], "c_blocks": best["c_blocks"],
                    "c_nnz": best["c_nnz"], "scalar_products": int(P), "total_ms": round(t_total * 1e3, 3), "timing": "median of 10 products",
                    "gflops": round(2.0 * P / t_total / 1e9, 2),
                    "stage_us": {k: round(best["t_us"][i], 1) for k, i in (("T_1", 1), ("T_2", 2), ("T_3", 3), ("T_4", 4), ("T_5", 5), ("T_6", 6), ("T_9", 9), ("T_7", 7))},
                    "sort_path": "segmented" if best["sort_path"] else "global radix",
                    "roofline": {"bound": "mfma" if dtype == B.F16 else "valu", "kernel": "block_mac", "achieved": round(f_mac / t_mac / 1e12, 3),
                                 "peak": peak, "unit": "TFLOP/s", "frac": round(f_mac / t_mac / 1e12 / peak, 5), "traffic": None}})
    return res


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
    if "coo" in wl:
        m, ptr, col, val = csr_of(wl["coo"])
        x = np.ones(m.shape[1], np.float32); y = np.zeros(m.shape[0], np.float32)
        best = None
        for alg, name in ((0, "default"), (1, "adaptive"), (2, "rowsplit"), (3, "lrb")):
            ms, pre = C.c_double(), C.c_double()
            rc = V.vendor_csr_spmv(m.shape[0], m.shape[1], m.nnz, ptr.ctypes.data, col.ctypes.data, val.ctypes.data, x.ctypes.data, y.ctypes.data,
                                   alg, 200, C.byref(ms), C.byref(pre))
            if rc == 0 and (best is None or ms.value < best[1]):
                best = (name, ms.value, pre.value)
        if best:
            res["spmv"] = {"alg": best[0], "ms": round(best[1], 5), "preprocess_ms": round(best[2], 3),
                           "effective_GBs": round(eff_bytes / (best[1] * 1e-3) / 1e9, 1), "y_checksum": float(y.sum())}
    if not args.skip_spgemm:
        res["spgemm"] = []
        for name, coo in (("2cubes_sphere-like banded(101492, half_bw=8)", gen.banded(101492, 8)), ("cage12-like local+random(130228, 15.6/row)", gen.cage_like(130228, 15.6))):
            m, ptr, col, val = csr_of(coo)
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
    """SURVEY 8(e), SpMV row: ONE matrix cut into nnz-balanced block-row panels, x replicated, y slices all-gathered (strong
    scaling; the headline `value` above is the weak-scaling replica rate)."""
    from pybmsp import shard
    bounds = shard.spmv_row_bounds(A, world)
    view = A.row_panel(int(bounds[rank]), int(bounds[rank + 1]))  # built once: keeps its sweep plan across products
    full, st = shard.spmv_sharded(A, x, rank, world, dist, torch, bounds, view)  # warm-up, and the check below
    ref = torch.empty_like(full)
    B.check(B.lib().bmsp_memcpy_d2d(ref.data_ptr(), y_whole.ptr, full.numel() * full.element_size()))
    same = bool(torch.equal(full, ref))
    reps = 20
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    gather_ms = 0.0
    for _ in range(reps):
        full, st = shard.spmv_sharded(A, x, rank, world, dist, torch, bounds, view)
        gather_ms += st["gather_ms"]
    torch.cuda.synchronize(); dist.barrier()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    return {"workload": "one matrix, row panels balanced by nnz, y all-gathered", "scaling": "strong", "n_gpus": world,
            "ms_per_product": round(ms, 4), "allgather_ms": round(gather_ms / reps, 4), "allgather_bytes": st["gather_bytes"],
            "matches_single_gpu_sweep": same}


def bench_spgemm_sharded(B, gen, np, torch, dist, rank, world):
    """configs[4]: row-panel-sharded SpGEMM on R-MAT scale 22 with an allgatherv of the C panels over RCCL; fixed total work ->
    strong scaling.  Edge factor 1 (+ identity): 2.3 G candidate block pairs, 0.69 G surviving tasks, 0.48 G C blocks -- the largest
    scale-22 instance whose single panel (N = 1) stays inside the 32-bit candidate range, so that N = 1, 2, 4, 8 run the same product
    (edge factor 2 has 7.9 G candidates)."""
    from pybmsp import shard
    scale = int(os.environ.get("BMSP_SHARD_SCALE", "22"))
    ef = float(os.environ.get("BMSP_SHARD_EF", "1"))
    n, _, r, c, v = gen.rmat(scale, ef)
    A = B.BmSpMatrix.from_coo(n, n, r, c, v, dtype=B.F16)
    Bt = B.BmSpMatrix.from_coo(n, n, r, c, v, transposed=True, dtype=B.F16)
    best = None
    for it in range(3):
        B.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        Cm, stats = shard.spgemm_sharded(A, Bt, rank, world, dist, torch, tc_version=4)
        B.synchronize(); dist.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt, stats["panel"]["t_us"][0] * 1e-6], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if it and (best is None or float(t[0].item()) < best[0]):
            best = (float(t[0].item()), stats, Cm.info(), float(t[1].item()))
        del Cm
    P = scalar_products(np, A) if rank == 0 else 0
    return {"workload": "row-panel-sharded SpGEMM fp16 MFMA, rmat(scale=%d, ef=%g)+I" % (scale, ef), "scaling": "strong", "n_gpus": world,
            "total_ms": round(best[0] * 1e3, 3), "slowest_panel_product_ms": round(best[3] * 1e3, 3), "gflops": round(2.0 * P / best[0] / 1e9, 2),
            "c_blocks": best[2]["block_num"],
            "c_nnz": best[2]["nnz"], "allgatherv_bytes": best[1]["gather_bytes"], "allgatherv_ms": round(best[1]["gather_ms"], 3),
            "panel_tasks": best[1]["tasks"]}


def cpu_baseline(wl, eff_bytes, args):
    """oracle (`port` of cusp::multiply, csr_spmv.h:56-73 + the OpenMP row-parallel variant) on this box's host cores,
    on the same matrix and x; bounded to ~cpu-seconds of work."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import numpy as np
    import oracle as O
    from pybmsp import gen
    if "path" in wl:
        coo = O.mtx_read(wl["path"], strict=True)
    else:
        n, _, r, c, v = wl["coo"]
        coo = O.Coo(n, n, r, c, v)
    A = O.csr_from_coo(coo)
    x = gen.spmv_x(A.num_cols, "ones")
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
    th = max(1, min(O.max_threads(), avail, 64))
    res = {}
    for name, t in (("omp", th), ("seq", 1)):
        O.csr_spmv(A, x, t)
        t0 = time.perf_counter()
        O.csr_spmv(A, x, t)
        one = max(1e-6, time.perf_counter() - t0)
        iters = int(min(2000, max(10, (args.cpu_seconds / 2) / one)))
        t0 = time.perf_counter()
        for _ in range(iters):
            O.csr_spmv(A, x, t)
        dt = (time.perf_counter() - t0) / iters
        res[name] = (eff_bytes / dt / 1e9, iters, dt)
    if res["seq"][0] > res["omp"][0]:  # oversubscribed box: the single-thread figure is the better baseline
        res["omp"], th = res["seq"], 1
    return {"value": round(res["omp"][0], 2), "unit": "GB/s", "cores": th, "kind": "port",
            "sample": "CSR SpMV (cusp::multiply restatement, OpenMP row-parallel) on the same matrix and x, %d iterations, %.2f ms each; "
                      "single-thread: %.2f GB/s over %d iterations" % (res["omp"][1], res["omp"][2] * 1e3, res["seq"][0], res["seq"][1])}


if __name__ == "__main__":
    main()
