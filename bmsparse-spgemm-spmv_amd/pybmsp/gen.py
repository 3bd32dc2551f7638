"""Synthetic inputs of SURVEY.md 8(d): banded (FEM-like, dense-ish tiles) and R-MAT (graph-like, hyper-sparse tiles).

Deterministic on every platform: all randomness comes from a counter-based splitmix64 evaluated with numpy
uint64 arithmetic.  Matrices are returned as host COO triples (0-based int32 rows/cols, float64 values) with
duplicates merged (summed) and entries sorted by (row, col).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """vectorised splitmix64 finaliser of the counter array x (uint64)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(counter):
    """uniform doubles in [0,1) from a uint64 counter array."""
    return (splitmix64(counter) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _merge(n_rows, n_cols, r, c, v):
    key = r.astype(np.uint64) * np.uint64(n_cols) + c.astype(np.uint64)
    uk, inv = np.unique(key, return_inverse=True)
    vals = np.bincount(inv, weights=v, minlength=uk.size)
    rows = (uk // np.uint64(n_cols)).astype(np.int32)
    cols = (uk % np.uint64(n_cols)).astype(np.int32)
    return rows, cols, vals


def banded(n, half_bw, seed=1):
    """entries at |i-j| <= half_bw, values U(-1,1).  Generated in row chunks (the 75 M-entry ceiling case of bench.py would otherwise
    hold several 64-bit temporaries of its full length); the counter of entry (i, j) is its index in the n x (2 half_bw + 1) grid."""
    w = 2 * half_bw + 1
    rows, cols, vals = [], [], []
    step = max(1, (1 << 23) // w)
    offs = np.arange(-half_bw, half_bw + 1, dtype=np.int32)
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        i = np.repeat(np.arange(r0, r1, dtype=np.int32), w)
        j = i + np.tile(offs, r1 - r0)
        ok = (j >= 0) & (j < n)
        flat = np.arange(np.uint64(r0) * np.uint64(w), np.uint64(r1) * np.uint64(w), dtype=np.uint64)
        if not ok.all():
            i, j, flat = i[ok], j[ok], flat[ok]
        rows.append(i); cols.append(j)
        vals.append(2.0 * uniform01((np.uint64(seed) << np.uint64(40)) + flat) - 1.0)
    if not rows:
        return n, n, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)
    return n, n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def rmat(scale, edge_factor, a=0.57, b=0.19, c=0.19, seed=1, add_identity=True, chunk=1 << 22):
    """R-MAT graph with 2**scale vertices and edge_factor * 2**scale generated edges; duplicates summed,
    self-loops kept, identity added (so no block-row is empty), values U(0,1)."""
    n = 1 << scale
    m = int(edge_factor * n)
    keys, vals = [], []
    for e0 in range(0, m, chunk):
        e = np.arange(e0, min(m, e0 + chunk), dtype=np.uint64)
        r = np.zeros(e.size, dtype=np.uint64)
        cc = np.zeros(e.size, dtype=np.uint64)
        for lvl in range(scale):
            u = uniform01((np.uint64(seed) << np.uint64(48)) + e * np.uint64(64) + np.uint64(lvl))
            rb = (u >= a + b).astype(np.uint64)                      # quadrants c, d -> lower half
            cb = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.uint64)  # quadrants b, d -> right half
            r = (r << np.uint64(1)) | rb
            cc = (cc << np.uint64(1)) | cb
        keys.append(r * np.uint64(n) + cc)
        vals.append(uniform01((np.uint64(seed + 7) << np.uint64(48)) + e))
    if add_identity:
        d = np.arange(n, dtype=np.uint64)
        keys.append(d * np.uint64(n) + d)
        vals.append(np.ones(n))
    key = np.concatenate(keys) if keys else np.zeros(0, np.uint64)
    val = np.concatenate(vals) if vals else np.zeros(0)
    uk, inv = np.unique(key, return_inverse=True)
    v = np.bincount(inv, weights=val, minlength=uk.size)
    return n, n, (uk // np.uint64(n)).astype(np.int32), (uk % np.uint64(n)).astype(np.int32), v


def cage_like(n, per_row=15.6, local_frac=0.75, half_bw=24, seed=1):
    """stand-in for the cage family (DNA electrophoresis): ~per_row entries per row, most of them within a band around the
    diagonal, the rest uniformly random; values U(0,1); diagonal present."""
    m = int(n * (per_row - 1))
    e = np.arange(m, dtype=np.uint64)
    s = np.uint64(seed) << np.uint64(44)
    r = (splitmix64(s + e * np.uint64(4)) % np.uint64(n)).astype(np.int64)
    u = uniform01(s + e * np.uint64(4) + np.uint64(1))
    off = (splitmix64(s + e * np.uint64(4) + np.uint64(2)) % np.uint64(2 * half_bw + 1)).astype(np.int64) - half_bw
    far = (splitmix64(s + e * np.uint64(4) + np.uint64(3)) % np.uint64(n)).astype(np.int64)
    c = np.where(u < local_frac, np.clip(r + off, 0, n - 1), far)
    v = uniform01(s + e * np.uint64(4) + np.uint64(1) + (np.uint64(1) << np.uint64(40)))
    d = np.arange(n, dtype=np.int64)
    rows, cols, vals = _merge(n, n, np.concatenate([r, d]), np.concatenate([c, d]), np.concatenate([v, np.ones(n)]))
    return n, n, rows, cols, vals


def random_coo(n_rows, n_cols, nnz, seed=1, lo=-1.0, hi=1.0, integer=False):
    """nnz random coordinates (duplicates merged), for ragged / empty-row edge cases."""
    e = np.arange(nnz, dtype=np.uint64)
    r = (splitmix64((np.uint64(seed) << np.uint64(40)) + e * np.uint64(3)) % np.uint64(max(1, n_rows))).astype(np.int64)
    c = (splitmix64((np.uint64(seed) << np.uint64(40)) + e * np.uint64(3) + np.uint64(1)) % np.uint64(max(1, n_cols))).astype(np.int64)
    v = lo + (hi - lo) * uniform01((np.uint64(seed) << np.uint64(40)) + e * np.uint64(3) + np.uint64(2))
    if integer:
        v = np.floor(v)
        v[v == 0] = 1.0
    rows, cols, vals = _merge(n_rows, n_cols, r, c, v)
    return n_rows, n_cols, rows, cols, vals


def spmv_x(n, kind="ones"):
    """x vectors of SURVEY 8(d): all-ones (reference main, SPMV.cu:279-281) or (i % 21) - 10 (CUSP benchmark)."""
    if kind == "ones":
        return np.ones(n, dtype=np.float32)
    return ((np.arange(n) % 21) - 10).astype(np.float32)


# ---- CUSP gallery generators (restated from cusp/gallery/detail/{poisson,stencil,random}.inl) ---------------------------------
_STENCILS = {
    # (offsets per grid dimension, first dimension fastest), centre value; poisson.inl:29-118
    "5pt": ([(0, -1), (-1, 0), (0, 0), (1, 0), (0, 1)], 4.0),
    "9pt": ([(i, j) for j in (-1, 0, 1) for i in (-1, 0, 1)], 8.0),
    "7pt": ([(0, 0, -1), (0, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], 6.0),
    "27pt": ([(i, j, k) for k in (-1, 0, 1) for j in (-1, 0, 1) for i in (-1, 0, 1)], 26.0),
}


def poisson(kind, *grid):
    """cusp::gallery::poisson{5,9,7,27}pt on a grid (first dimension fastest, stencil.inl:38-50): row = x + m*(y + n*z),
    neighbours outside the grid are dropped, off-centre value -1, centre value 4 / 8 / 6 / 26.  Sorted by (row, col)."""
    pts, centre = _STENCILS[kind]
    grid = [int(g) for g in grid]
    assert len(grid) == len(pts[0])
    n = int(np.prod(grid))
    idx = np.arange(n, dtype=np.int64)
    coords, rem = [], idx.copy()
    for g in grid:
        coords.append(rem % g)
        rem //= g
    strides = np.cumprod([1] + grid[:-1])
    rows, cols, vals = [], [], []
    for p in pts:
        ok = np.ones(n, dtype=bool)
        off = 0
        for d, (dx, g) in enumerate(zip(p, grid)):
            x = coords[d] + dx
            ok &= (x >= 0) & (x < g)
            off += int(strides[d]) * dx
        rows.append(idx[ok])
        cols.append(idx[ok] + off)
        vals.append(np.full(int(ok.sum()), centre if all(q == 0 for q in p) else -1.0))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    o = np.lexsort((c, r))
    return n, n, r[o].astype(np.int32), c[o].astype(np.int32), v[o]


def cusp_random(m, n, num_samples):
    """cusp::gallery::random (random.inl:30-60): srand(m ^ n ^ samples); rand() % m, rand() % n, value 1; sorted, duplicates
    dropped.  Uses the C library's rand(), as the reference does (the committed fixtures pin glibc's sequence)."""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.srand(ctypes.c_uint(m ^ n ^ num_samples))
    pairs = set()
    for _ in range(num_samples):
        r = libc.rand() % m
        c = libc.rand() % n
        pairs.add((r, c))
    pairs = sorted(pairs)
    r = np.array([p[0] for p in pairs], dtype=np.int32)
    c = np.array([p[1] for p in pairs], dtype=np.int32)
    return m, n, r, c, np.ones(len(pairs))


def fem_like(side=47, kind="27pt", window=32, seed=1, values="random"):
    """structurally faithful stand-in for a 3-D FEM matrix such as 2cubes_sphere (101 492 rows, 16.2 entries per row):
    the poisson 7pt / 27pt stencil on a side^3 grid (47^3 = 103 823 rows) under a bandwidth-limited random symmetric
    permutation -- vertex ids are shuffled inside consecutive windows of `window` ids, as a mesh generator's numbering would
    scatter neighbours -- so tiles are partly filled and block columns irregular instead of the perfect diagonals of the
    lexicographic grid.  values: "random" U(-1,1) (symmetric pattern, unsymmetric values) or "stencil" (the integer stencil)."""
    n, _, r, c, v = poisson(kind, side, side, side)
    ids = np.arange(n, dtype=np.int64)
    key = (ids // window) * np.int64(1 << 32) + (splitmix64((np.uint64(seed) << np.uint64(40)) + ids.astype(np.uint64)) >> np.uint64(40)).astype(np.int64)
    perm = np.empty(n, dtype=np.int64)
    perm[np.argsort(key, kind="stable")] = ids   # old id -> new id, windows stay in place
    r2, c2 = perm[r], perm[c]
    if values == "random":
        v = 2.0 * uniform01((np.uint64(seed + 3) << np.uint64(44)) + (r2.astype(np.uint64) * np.uint64(n) + c2.astype(np.uint64))) - 1.0
    o = np.lexsort((c2, r2))
    return n, n, r2[o].astype(np.int32), c2[o].astype(np.int32), v[o]
