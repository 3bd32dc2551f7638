"""CPU tests of the N > 1 path (world_size 2, gloo).  The panel products come from the oracle here (no GPU in this container); what is
under test is the exchange of csrc/comm.hip with a real second rank: one all-gather of the panel sizes, the slice layout computed by
THE LIBRARY (bmsp_shard_layout / bmsp_shard_row_slices -- libbmsp.so loads without a GPU; these are the functions the RCCL and the
loopback transports call), one broadcast per rank straight into its final slice (the shape of exchange_slices: P x broadcast(root = r)),
offsets re-based by value_start[r], terminal offset.  The device side of the same code (loopback transport, P in {2, 3, 8}) is
tests/test_gpu_parity.py::test_sharded_operators_loopback; RCCL itself with N > 1 needs N GPUs."""
import os
import socket
import sys
import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _balanced_bounds(work_per_row, parts):
    """host mirror of bmsp_partition_rows: first block-row whose cumulative work reaches p/parts of the total."""
    cum = np.concatenate([[0], np.cumsum(work_per_row)])
    total = int(cum[-1])
    bounds = [0]
    r = 0
    for p in range(1, parts):
        target = total * p // parts
        while r < len(work_per_row) and cum[r] < target:
            r += 1
        bounds.append(max(r, bounds[-1]))
    bounds.append(len(work_per_row))
    return bounds


def _worker(rank, world, port, out_dir):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "bmsparse-spgemm-spmv_amd"))
    sys.path.insert(0, os.path.join(repo, "oracle"))
    import torch
    import torch.distributed as dist
    import oracle as O
    import pybmsp as B
    from pybmsp import gen
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, _, r, c, v = gen.rmat(9, 6, seed=3)
    A = O.bmsp_from_coo(O.Coo(n, n, r, c, v), O.F32, False)
    Bt = O.bmsp_from_coo(O.Coo(n, n, r, c, v), O.F32, True)
    whole, _ = O.spgemm(A, Bt)
    # candidate-task count per block-row of A = sum over its blocks of the blocks in B's matching block-row
    nbr = (n + 7) // 8
    b_rows = (Bt.keys >> np.uint64(32)).astype(np.int64)
    b_per_row = np.bincount(b_rows, minlength=nbr)
    a_rows = (A.keys >> np.uint64(32)).astype(np.int64)
    a_cols = (A.keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    work = np.bincount(a_rows, weights=b_per_row[a_cols], minlength=nbr).astype(np.int64)
    bounds = _balanced_bounds(work, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    # this rank's panel of A (block-rows [lo, hi)) as its own matrix with global keys
    sel = (a_rows >= lo) & (a_rows < hi)
    first = int(np.argmax(sel)) if sel.any() else 0
    cnt = int(sel.sum())
    o0 = int(A.offsets[first]) if cnt else 0
    o1 = int(A.offsets[first + cnt]) if cnt else 0
    panel = O.Bmsp(n, n, O.F32, 0, A.keys[sel], A.bmps[sel], A.offsets[first:first + cnt + 1] - np.uint64(o0), A.values[o0:o1])
    Cp, st = O.spgemm(panel, Bt)
    # sizes of every panel (comm.hip: exchange_sizes), then the library's layout
    mine = torch.tensor([Cp.block_num, Cp.nnz], dtype=torch.int64)
    sizes_t = torch.empty(2 * world, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes_t, mine)
    nb, nz = sizes_t.numpy()[0::2].copy(), sizes_t.numpy()[1::2].copy()
    b0, z0 = B.shard_layout(nb, nz)
    NB, NZ = int(b0[-1]), int(z0[-1])
    keys_t, bmps_t = torch.zeros(NB, dtype=torch.int64), torch.zeros(NB, dtype=torch.int64)
    offs_t, vals_t = torch.zeros(NB + 1, dtype=torch.int64), torch.zeros(NZ, dtype=torch.float32)
    nbytes = 0
    # every panel straight into its slice: one broadcast per rank and array (comm.hip: exchange_slices)
    for src_arr, dst, start, cnt in ((Cp.keys, keys_t, b0, nb), (Cp.bmps, bmps_t, b0, nb), (Cp.offsets[:-1], offs_t, b0, nb), (Cp.values, vals_t, z0, nz)):
        for q in range(world):
            if cnt[q] == 0:
                continue
            sl = dst[int(start[q]):int(start[q]) + int(cnt[q])]
            if q == rank:
                sl.copy_(torch.from_numpy(np.ascontiguousarray(src_arr).view(np.int64) if dst.dtype == torch.int64 else np.ascontiguousarray(src_arr, dtype=np.float32)))
            dist.broadcast(sl, src=q)
            nbytes += sl.numel() * sl.element_size()
    # a panel's offsets count from its own first value: re-base by the values in front of it; terminal offset
    for q in range(world):
        if nb[q] and z0[q]:
            offs_t[int(b0[q]):int(b0[q]) + int(nb[q])] += int(z0[q])
    offs_t[NB] = NZ
    keys, bmps, offs = (t.numpy().view(np.uint64) for t in (keys_t, bmps_t, offs_t))
    ok = (np.array_equal(keys, whole.keys) and np.array_equal(bmps, whole.bmps) and np.array_equal(offs, whole.offsets)
          and np.array_equal(vals_t.numpy().astype(np.float64), whole.values.astype(np.float64)))
    sizes = [int(x) for x in nb]
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %s %d\n" % (int(ok), nbytes, ",".join(map(str, sizes)), int(work[lo:hi].sum())))
    dist.barrier()
    dist.destroy_process_group()


def _spmv_worker(rank, world, port, out_dir):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "bmsparse-spgemm-spmv_amd"))
    sys.path.insert(0, os.path.join(repo, "oracle"))
    import torch
    import torch.distributed as dist
    import oracle as O
    import pybmsp as B
    from pybmsp import gen
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, _, r, c, v = gen.rmat(10, 5, seed=4)
    n_rows = n - 3  # ragged last block-row
    keep = r < n_rows
    A = O.bmsp_from_coo(O.Coo(n_rows, n, r[keep], c[keep], v[keep]), O.F32, False)
    x = gen.spmv_x(n, "cusp")
    whole = O.spmv_f32(A, x)
    nbr = (n_rows + 7) // 8
    a_rows = (A.keys >> np.uint64(32)).astype(np.int64)
    nnz_per_block = np.diff(A.offsets.astype(np.int64))
    work = np.bincount(a_rows, weights=nnz_per_block, minlength=nbr).astype(np.int64)
    bounds = _balanced_bounds(work, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    sel = (a_rows >= lo) & (a_rows < hi)
    first = int(np.argmax(sel)) if sel.any() else 0
    cnt = int(sel.sum())
    o0 = int(A.offsets[first]) if cnt else 0
    o1 = int(A.offsets[first + cnt]) if cnt else 0
    panel = O.Bmsp(n_rows, n, O.F32, 0, A.keys[sel], A.bmps[sel], A.offsets[first:first + cnt + 1] - np.uint64(o0), A.values[o0:o1])
    y = O.spmv_f32(panel, x)
    # which rows every panel delivers: the library's arithmetic (comm.hip: shard_row_slices); slices broadcast in place
    rs, rc = B.shard_row_slices(n_rows, bounds)
    r0, r1 = int(rs[rank]), int(rs[rank] + rc[rank])
    assert r0 == min(lo * 8, n_rows) and r1 == min(hi * 8, n_rows)
    assert not y[:r0].any() and not y[r1:].any()
    full_t = torch.full((n_rows,), float("nan"), dtype=torch.float32)
    full_t[r0:r1] = torch.from_numpy(np.ascontiguousarray(y[r0:r1]))
    nbytes = 0
    for q in range(world):
        if rc[q]:
            sl = full_t[int(rs[q]):int(rs[q] + rc[q])]
            dist.broadcast(sl, src=q)
            nbytes += sl.numel() * 4
    full = full_t.numpy()
    ok = full.shape == whole.shape and np.array_equal(full, whole)
    with open(os.path.join(out_dir, "spmv_rank%d.txt" % rank), "w") as f:
        f.write("%d %d %d\n" % (int(ok), nbytes, int(work[lo:hi].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_spmv_exchange_world2(tmp_path):
    """SURVEY 8(e), SpMV row: nnz-balanced block-row panels, x replicated, the y slices broadcast in place."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_spmv_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [open(tmp_path / ("spmv_rank%d.txt" % r)).read().split() for r in range(2)]
    assert all(r[0] == "1" for r in res), res
    assert res[0][1] == res[1][1] and int(res[0][1]) > 0
    w0, w1 = int(res[0][2]), int(res[1][2])
    assert abs(w0 - w1) <= 0.35 * (w0 + w1)


def test_sharded_spgemm_exchange_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [open(tmp_path / ("rank%d.txt" % r)).read().split() for r in range(2)]
    assert all(r[0] == "1" for r in res), res            # every rank reassembled exactly the whole product
    assert res[0][1] == res[1][1] and int(res[0][1]) > 0  # same gathered byte count on both ranks
    assert res[0][2] == res[1][2]
    w0, w1 = int(res[0][3]), int(res[1][3])
    assert abs(w0 - w1) <= 0.35 * (w0 + w1)              # panels balanced by candidate-task count


def test_balanced_bounds_properties():
    rng = np.random.default_rng(0)
    work = (rng.pareto(1.2, 1000) * 10).astype(np.int64)
    for parts in (1, 2, 3, 8):
        b = _balanced_bounds(work, parts)
        assert b[0] == 0 and b[-1] == 1000 and all(x <= y for x, y in zip(b, b[1:]))
        loads = [int(work[b[i]:b[i + 1]].sum()) for i in range(parts)]
        assert max(loads) <= work.sum() / parts + work.max()


def test_shard_layout_host_functions(bmsp):
    """bmsp_shard_layout / bmsp_shard_row_slices: exclusive sums incl. empty panels, ragged last block-row, bad input refused."""
    b0, z0 = bmsp.shard_layout([3, 0, 5, 1], [10, 0, 7, 64])
    assert b0.tolist() == [0, 3, 3, 8, 9] and z0.tolist() == [0, 10, 10, 17, 81]
    rs, rc = bmsp.shard_row_slices(21, [0, 1, 1, 3])  # 21 rows = 3 block-rows, the last one ragged; an empty panel in the middle
    assert rs.tolist() == [0, 8, 8] and rc.tolist() == [8, 0, 13]
    with pytest.raises(bmsp.BmspError):
        bmsp.shard_layout([1, -1], [0, 0])
    with pytest.raises(bmsp.BmspError):
        bmsp.shard_row_slices(16, [0, 2, 1])
