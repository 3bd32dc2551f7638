"""Row-panel sharding of the SpGEMM over one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

C's block-row i needs A's block-row i and all of B: B is replicated, A is cut into contiguous
block-row panels balanced by candidate-task count (bmsp_partition_rows), every rank runs the whole pipeline on its panel
and the four arrays of the C panels are exchanged with ONE size all-gather and TWO padded all-gathers (integer arrays
packed together, values), then concatenated with re-based offsets (bmsp_matrix_concat_panels).  RCCL has no allgatherv;
panels are balanced by work, so padding to the largest panel wastes little.

The exchange itself (`allgatherv`) only needs torch tensors, so the same code runs under gloo on CPU tensors in the
world_size-2 CPU test.
"""
import ctypes as C
import time
import numpy as np


def allgatherv(tensors, dist, torch):
    """tensors: list of 1-D tensors on this rank (same dtypes/devices on every rank, different lengths).
    returns: list over ranks of lists of tensors (views into the gathered buffers), and the bytes received."""
    world = dist.get_world_size()
    dev = tensors[0].device
    sizes = torch.tensor([t.numel() for t in tensors], dtype=torch.int64, device=dev)
    all_sizes = torch.empty(world * len(tensors), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes)
    all_sizes = all_sizes.cpu().view(world, len(tensors))
    out = [[None] * len(tensors) for _ in range(world)]
    nbytes = 0
    for j, t in enumerate(tensors):
        pad = int(all_sizes[:, j].max().item())
        send = t if t.numel() == pad else torch.cat([t, torch.zeros(pad - t.numel(), dtype=t.dtype, device=dev)])
        recv = torch.empty(world * max(pad, 1), dtype=t.dtype, device=dev)
        if pad:
            dist.all_gather_into_tensor(recv[: world * pad], send.contiguous())
        for r in range(world):
            out[r][j] = recv[r * pad: r * pad + int(all_sizes[r, j])]
        nbytes += int(all_sizes[:, j].sum().item()) * t.element_size()
    return out, nbytes


def concat_host(panels):
    """host mirror of bmsp_matrix_concat_panels for tests: panels = [(keys, bmps, offsets[nb+1], values)]."""
    keys = np.concatenate([p[0] for p in panels])
    bmps = np.concatenate([p[1] for p in panels])
    vals = np.concatenate([p[3] for p in panels])
    offs, base = [], 0
    for p in panels:
        o = np.asarray(p[2], dtype=np.uint64)
        offs.append(o[:-1] - o[0] + np.uint64(base))
        base += int(o[-1] - o[0])
    offs.append(np.array([base], dtype=np.uint64))
    return keys, bmps, np.concatenate(offs), vals


def _as_tensor(torch, darr, tdtype):
    """copies a pybmsp DeviceArray into a fresh CUDA tensor (device-to-device)."""
    import pybmsp as B
    t = torch.empty(darr.n, dtype=tdtype, device="cuda")
    if darr.n:
        B.check(B.lib().bmsp_memcpy_d2d(t.data_ptr(), darr.ptr, darr.n * darr.dtype.itemsize))
    return t


def spgemm_sharded(A, Bt, rank, world, dist, torch, tc_version=5, mode=0):
    """returns (C as a BmSpMatrix holding the whole product on every rank, stats dict)."""
    import pybmsp as B
    bounds = B.partition_rows(A, Bt, world)
    view = A.row_panel(int(bounds[rank]), int(bounds[rank + 1]))
    Cp, st = B.spgemm(view, Bt, mode=mode, tc_version=tc_version)
    k, b, o, v = Cp.device_arrays()
    info = Cp.info()
    vt = torch.float64 if info["dtype"] == B.F64 else torch.float32
    # keys | bmps | offsets packed into one int64 message
    ints = torch.cat([_as_tensor(torch, k, torch.int64), _as_tensor(torch, b, torch.int64), _as_tensor(torch, o, torch.int64)])
    vals = _as_tensor(torch, v, vt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gathered, nbytes = allgatherv([ints, vals], dist, torch)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3
    P = world
    bn = np.array([(g[0].numel() - 1) // 3 for g in gathered], dtype=np.int64)
    nz = np.array([g[1].numel() for g in gathered], dtype=np.int64)
    es = 8
    kp = (C.c_void_p * P)(*[g[0].data_ptr() for g in gathered])
    bp = (C.c_void_p * P)(*[g[0].data_ptr() + es * int(bn[r]) for r, g in enumerate(gathered)])
    op = (C.c_void_p * P)(*[g[0].data_ptr() + es * 2 * int(bn[r]) for r, g in enumerate(gathered)])
    vp = (C.c_void_p * P)(*[g[1].data_ptr() for g in gathered])
    h = C.c_void_p()
    B.check(B.lib().bmsp_matrix_concat_panels(info["num_rows"], info["num_cols"], P, bn.ctypes.data, nz.ctypes.data, kp, bp, op, vp,
                                              info["dtype"], C.byref(h)))
    stats = {"tasks": int(st["surviving_tasks"]), "gather_bytes": int(nbytes), "gather_ms": gather_ms,
             "bounds": [int(x) for x in bounds], "panel": st}
    return B.BmSpMatrix(h.value), stats


def balanced_bounds(work_per_row, parts):
    """block-row bounds [b_0 = 0, ..., b_parts = nbr]: panel p ends at the first block-row where the cumulative work reaches
    p / parts of the total (the rule of bmsp_partition_rows, csrc/shard.hip, on a host array)."""
    work = np.asarray(work_per_row, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(work)])
    total = int(cum[-1])
    bounds = [0]
    for p in range(1, parts):
        r = int(np.searchsorted(cum, total * p // parts, side="left"))
        bounds.append(max(min(r, len(work)), bounds[-1]))
    bounds.append(len(work))
    return bounds


def spmv_row_bounds(A, parts):
    """SURVEY.md 8(e), SpMV: block-row panels balanced by stored values (nnz)."""
    rowptr = A.block_row_ptr().astype(np.int64)
    offsets = A.host_arrays()[2].astype(np.int64)
    return balanced_bounds(np.diff(offsets[rowptr]), parts)


def spmv_sharded(A, x, rank, world, dist, torch, bounds=None, view=None):
    """y = A x with A cut into block-row panels (x replicated): every rank sweeps its panel, the y slices are exchanged with one
    padded all-gather, and every rank returns the whole y as a CUDA tensor.  No collective touches the sweep itself.
    Pass the rank's `view` (A.row_panel(bounds[rank], bounds[rank + 1])) to reuse its cached sweep plan across calls."""
    import pybmsp as B
    info = A.info()
    if bounds is None:
        bounds = spmv_row_bounds(A, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    if view is None:
        view = A.row_panel(lo, hi)
    y = B.spmv(view, x)  # rows outside the panel come out as zeros
    r0, r1 = min(lo * 8, info["num_rows"]), min(hi * 8, info["num_rows"])
    tdt = torch.float64 if info["dtype"] == B.F64 else torch.float32
    mine = torch.empty(r1 - r0, dtype=tdt, device="cuda")
    if r1 > r0:
        B.check(B.lib().bmsp_memcpy_d2d(mine.data_ptr(), y.ptr + r0 * y.dtype.itemsize, (r1 - r0) * y.dtype.itemsize))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gathered, nbytes = allgatherv([mine], dist, torch)
    full = torch.cat([g[0] for g in gathered])
    torch.cuda.synchronize()
    return full, {"gather_ms": (time.perf_counter() - t0) * 1e3, "gather_bytes": int(nbytes), "bounds": [int(b) for b in bounds]}
