// comm.hip -- the multi-GPU path behind the C ABI: one process per GPU, RCCL over xGMI.
//
// New relative to the reference (single GPU, no communication code at all: SURVEY.md 2.3).  SURVEY.md 8(e): C's block-row i
// needs A's block-row i and all of B, so B is replicated, A is cut into contiguous block-row panels balanced by candidate-task
// count (partition_rows), every rank runs the whole pipeline on its panel (spgemm on a row_panel view) and the four arrays of the
// C panels are exchanged ONCE:
//   * one ncclAllGather of the 2 x P panel sizes (blocks, values);
//   * the whole C is allocated at its final size on every rank and every rank's panel is BROADCAST straight into its final
//     slice (ncclGroupStart ... P x ncclBroadcast(root = r) ... ncclGroupEnd per array): an allgatherv without padding and
//     without a staging copy -- RCCL has no allgatherv, and panels balanced by WORK differ in SIZE by > 2x on skewed inputs.
//     On the fully connected xGMI of an MI355X node the P concurrent broadcasts use all links at once.
//   * offsets are re-based in place by an exclusive scan of the panels' value counts.
// SpMV: block-row panels balanced by stored values, x replicated, every rank sweeps its panel straight into the full-length y and
// the y slices are broadcast in place the same way.
//
// librccl is NOT a link-time dependency of libbmsp.so: it is opened on the first bmsp_comm_* call (the copy already loaded in the
// process -- e.g. PyTorch's -- if there is one, /opt/rocm/lib/librccl.so.1 otherwise), so single-GPU users never load it.
#include "matrix.h"
#include "prims.hip.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <unistd.h>
#include <vector>

namespace bmsp {
void partition_rows(bmsp_matrix_s *A, bmsp_matrix_s *B, int parts, int64_t *bounds, hipStream_t st, uint64_t *total_out);
bmsp_matrix_s *row_panel(bmsp_matrix_s *m, int64_t rb, int64_t re, hipStream_t st);
void spgemm_paneled(bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **C, int mode, int tc_version, int verbose, hipStream_t st,
                    bmsp_spgemm_stats *stats);

namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

Rccl &rccl()
{
    static Rccl R;
    if (R.h) return R;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)
        if (!R.h) R.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // a copy the process already holds (PyTorch ships its own)
    if (!R.h) {
        const char *env = getenv("BMSP_RCCL_LIB");
        R.h = dlopen(env ? env : "/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!R.h) R.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!R.h) fail(BMSP_ERR_UNSUPPORTED, "cannot open librccl: %s", dlerror());
#define BMSP_SYM(field, name)                                                   \
    R.field = (decltype(R.field))dlsym(R.h, name);                              \
    if (!R.field) fail(BMSP_ERR_UNSUPPORTED, "librccl lacks %s", name);
    BMSP_SYM(GetUniqueId, "ncclGetUniqueId")
    BMSP_SYM(CommInitRank, "ncclCommInitRank")
    BMSP_SYM(CommDestroy, "ncclCommDestroy")
    BMSP_SYM(GetErrorString, "ncclGetErrorString")
    BMSP_SYM(Broadcast, "ncclBroadcast")
    BMSP_SYM(AllGather, "ncclAllGather")
    BMSP_SYM(GroupStart, "ncclGroupStart")
    BMSP_SYM(GroupEnd, "ncclGroupEnd")
#undef BMSP_SYM
    return R;
}

#define BMSP_NCCL(call)                                                                                          \
    do {                                                                                                         \
        ncclResult_t r__ = (call);                                                                               \
        if (r__ != ncclSuccess) fail(BMSP_ERR_HIP, "%s failed at %s:%d: %s", #call, __FILE__, __LINE__, rccl().GetErrorString(r__)); \
    } while (0)

struct AddU64 {
    uint64_t *p;
    uint64_t add;
    __device__ void operator()(uint64_t i) const { p[i] += add; }
};

// every rank's `count[r]` elements of `elem` bytes land at dst + elem * start[r]; rank `me` sends `mine`
void allgatherv_inplace(bmsp_comm_s *c, const void *mine, void *dst, const std::vector<int64_t> &count, const std::vector<int64_t> &start,
                        size_t elem, hipStream_t st)
{
    Rccl &R = rccl();
    BMSP_NCCL(R.GroupStart());
    for (int r = 0; r < c->world; r++) {
        if (count[(size_t)r] == 0) continue;
        char *slice = (char *)dst + elem * (size_t)start[(size_t)r];
        BMSP_NCCL(R.Broadcast(r == c->rank ? mine : (const void *)slice, slice, elem * (size_t)count[(size_t)r], ncclChar, r, (ncclComm_t)c->comm, st));
    }
    BMSP_NCCL(R.GroupEnd());
}

}  // namespace

void comm_unique_id(void *id128)
{
    ncclUniqueId id;
    BMSP_NCCL(rccl().GetUniqueId(&id));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
}

bmsp_comm_s *comm_init(const void *id128, int world, int rank)
{
    if (world < 1 || rank < 0 || rank >= world) fail(BMSP_ERR_INVALID, "rank %d outside world %d", rank, world);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    std::unique_ptr<bmsp_comm_s> c(new bmsp_comm_s());
    c->rank = rank; c->world = world;
    BMSP_HIP(hipGetDevice(&c->device));
    ncclComm_t nc = nullptr;
    BMSP_NCCL(rccl().CommInitRank(&nc, world, id, rank));
    c->comm = nc;
    return c.release();
}

// BMSP_WORLD / BMSP_RANK / BMSP_COMM_FILE: the rendezvous of the drop-in executables (one process per GPU, started by any
// launcher): rank 0 writes the 128-byte id to the file (tmp + rename), the others wait for it
bmsp_comm_s *comm_init_from_env()
{
    const char *w = getenv("BMSP_WORLD"), *r = getenv("BMSP_RANK"), *f = getenv("BMSP_COMM_FILE");
    if (!w || !r) fail(BMSP_ERR_INVALID, "BMSP_WORLD and BMSP_RANK must be set");
    const int world = atoi(w), rank = atoi(r);
    if (world < 1 || rank < 0 || rank >= world) fail(BMSP_ERR_INVALID, "BMSP_RANK %d outside BMSP_WORLD %d", rank, world);
    char id[128];
    if (world == 1) {
        comm_unique_id(id);
        return comm_init(id, 1, 0);
    }
    if (!f) fail(BMSP_ERR_INVALID, "BMSP_COMM_FILE (a path every rank can reach) must be set when BMSP_WORLD > 1");
    const std::string path = f, tmp = path + ".tmp";
    if (rank == 0) {
        comm_unique_id(id);
        FILE *fp = fopen(tmp.c_str(), "wb");
        if (!fp || fwrite(id, 1, sizeof id, fp) != sizeof id) fail(BMSP_ERR_IO, "cannot write %s", tmp.c_str());
        fclose(fp);
        if (rename(tmp.c_str(), path.c_str()) != 0) fail(BMSP_ERR_IO, "cannot publish %s", path.c_str());
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            FILE *fp = fopen(path.c_str(), "rb");
            if (fp) {
                const size_t n = fread(id, 1, sizeof id, fp);
                fclose(fp);
                if (n == sizeof id) break;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) fail(BMSP_ERR_IO, "timed out waiting for %s", path.c_str());
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    bmsp_comm_s *c = comm_init(id, world, rank);
    if (rank == 0) (void)unlink(path.c_str());  // every rank has read it: the collective init above has completed
    return c;
}

void comm_free(bmsp_comm_s *c)
{
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy((ncclComm_t)c->comm);
    delete c;
}

void spgemm_sharded(bmsp_comm_s *c, bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **Cout, int mode, int tc_version, int verbose,
                    hipStream_t st, bmsp_spgemm_stats *stats, bmsp_shard_stats *sh)
{
    if (!c || !A || !B || !Cout) fail(BMSP_ERR_INVALID, "null argument");
    const int P = c->world;
    std::vector<int64_t> bounds((size_t)P + 1);
    partition_rows(A, B, P, bounds.data(), st, nullptr);
    // this rank's panel product
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> view(row_panel(A, bounds[(size_t)c->rank], bounds[(size_t)c->rank + 1], st), free_matrix);
    bmsp_matrix_s *cp_raw = nullptr;
    bmsp_spgemm_stats ps{};
    try {
        spgemm(view.get(), B, &cp_raw, mode, tc_version, verbose, st, &ps);
    } catch (const TaskRangeExceeded &) {  // a panel beyond one task list: run it in sub-panels (same answer)
        spgemm_paneled(view.get(), B, &cp_raw, mode, tc_version, verbose, st, &ps);
    }
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> cp(cp_raw, free_matrix);
    StageTimer tm(st, true);
    tm.mark(-1);
    // sizes of every panel
    DevBuf<int64_t> d_sizes((size_t)2 * P);
    int64_t mine[2] = {cp->block_num, cp->nnz};
    DevBuf<int64_t> d_mine(2);
    BMSP_HIP(hipMemcpyAsync(d_mine.p, mine, sizeof mine, hipMemcpyHostToDevice, st));
    BMSP_NCCL(rccl().AllGather(d_mine.p, d_sizes.p, 2, ncclInt64, (ncclComm_t)c->comm, st));
    std::vector<int64_t> sizes((size_t)2 * P);
    BMSP_HIP(hipMemcpyAsync(sizes.data(), d_sizes.p, 8 * sizes.size(), hipMemcpyDeviceToHost, st));
    BMSP_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> nb((size_t)P), nz((size_t)P), b0((size_t)P + 1, 0), z0((size_t)P + 1, 0);
    for (int r = 0; r < P; r++) {
        nb[(size_t)r] = sizes[(size_t)2 * r]; nz[(size_t)r] = sizes[(size_t)2 * r + 1];
        b0[(size_t)r + 1] = b0[(size_t)r] + nb[(size_t)r];
        z0[(size_t)r + 1] = z0[(size_t)r] + nz[(size_t)r];
    }
    const int64_t NB = b0[(size_t)P], NZ = z0[(size_t)P];
    // the whole C at its final size; every panel is broadcast into its slice
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> C(new bmsp_matrix_s(), free_matrix);
    C->num_rows = A->num_rows; C->num_cols = B->num_cols; C->dtype = cp->dtype; C->transposed = 0;
    C->block_num = NB; C->nnz = NZ;
    const size_t es = dtype_size(C->dtype);
    C->keys = (uint64_t *)pool_alloc(8 * (size_t)(NB ? NB : 1));
    C->bmps = (uint64_t *)pool_alloc(8 * (size_t)(NB ? NB : 1));
    C->offsets = (uint64_t *)pool_alloc(8 * ((size_t)NB + 1));
    C->values = pool_alloc(es * (size_t)(NZ ? NZ : 1));
    allgatherv_inplace(c, cp->keys, C->keys, nb, b0, 8, st);
    allgatherv_inplace(c, cp->bmps, C->bmps, nb, b0, 8, st);
    allgatherv_inplace(c, cp->offsets, C->offsets, nb, b0, 8, st);  // block_num entries per panel (the terminal one is rebuilt)
    allgatherv_inplace(c, cp->values, C->values, nz, z0, es, st);
    for (int r = 0; r < P; r++)
        if (nb[(size_t)r] && z0[(size_t)r]) device_for_each(AddU64{C->offsets + b0[(size_t)r], (uint64_t)z0[(size_t)r]}, (uint64_t)nb[(size_t)r], st);
    const uint64_t term = (uint64_t)NZ;
    BMSP_HIP(hipMemcpyAsync(C->offsets + NB, &term, 8, hipMemcpyHostToDevice, st));
    tm.mark(0);
    BMSP_HIP(hipStreamSynchronize(st));
    double t_us[10] = {0};
    tm.collect(t_us);
    if (stats) *stats = ps;
    if (sh) {
        sh->world = P; sh->rank = c->rank;
        sh->panel_block_row_begin = bounds[(size_t)c->rank]; sh->panel_block_row_end = bounds[(size_t)c->rank + 1];
        sh->panel_tasks = ps.surviving_tasks;
        sh->exchange_bytes = 24 * NB + (int64_t)es * NZ;
        sh->exchange_us = t_us[0];
    }
    *Cout = C.release();
}

// block-row bounds balanced by stored values
static void spmv_bounds(bmsp_matrix_s *A, int P, std::vector<int64_t> &bounds, hipStream_t st)
{
    ensure_rowptr(A, st);
    const int64_t nbr = A->num_block_rows();
    std::vector<uint32_t> rp((size_t)nbr + 1);
    BMSP_HIP(hipMemcpyAsync(rp.data(), A->rowptr, 4 * rp.size(), hipMemcpyDeviceToHost, st));
    std::vector<uint64_t> off((size_t)A->block_num + 1);
    BMSP_HIP(hipMemcpyAsync(off.data(), A->offsets, 8 * off.size(), hipMemcpyDeviceToHost, st));
    BMSP_HIP(hipStreamSynchronize(st));
    const uint64_t base = off[rp[0]], total = off[rp[(size_t)nbr]] - base;
    bounds.assign((size_t)P + 1, 0);
    int64_t r = 0;
    for (int p = 1; p < P; p++) {
        const unsigned __int128 target = (unsigned __int128)total * (unsigned)p / (unsigned)P;
        while (r < nbr && (unsigned __int128)(off[rp[(size_t)r]] - base) < target) r++;
        bounds[(size_t)p] = r;
    }
    bounds[(size_t)P] = nbr;
}

void spmv_sharded(bmsp_comm_s *c, bmsp_matrix_s *A, const void *x, void *y, int variant, hipStream_t st, bmsp_shard_stats *sh)
{
    if (!c || !A) fail(BMSP_ERR_INVALID, "null argument");
    const int P = c->world;
    // the panel view (with its cached sweep plan) is kept on the matrix for repeated products with the same communicator shape
    if (!A->shard_view || A->shard_world != P || A->shard_rank != c->rank) {
        free_matrix(A->shard_view);
        A->shard_view = nullptr;
        std::vector<int64_t> bounds;
        spmv_bounds(A, P, bounds, st);
        A->shard_bounds.assign(bounds.begin(), bounds.end());
        A->shard_view = row_panel(A, bounds[(size_t)c->rank], bounds[(size_t)c->rank + 1], st);
        A->shard_world = P; A->shard_rank = c->rank;
    }
    spmv(A->shard_view, x, y, variant, st);  // writes the whole y: zeros outside the panel
    StageTimer tm(st, sh != nullptr);
    tm.mark(-1);
    const size_t es = A->dtype == BMSP_F64 ? 8 : 4;
    std::vector<int64_t> cnt((size_t)P), start((size_t)P);
    int64_t bytes = 0;
    for (int r = 0; r < P; r++) {
        const int64_t r0 = std::min<int64_t>(A->shard_bounds[(size_t)r] * 8, A->num_rows), r1 = std::min<int64_t>(A->shard_bounds[(size_t)r + 1] * 8, A->num_rows);
        start[(size_t)r] = r0; cnt[(size_t)r] = r1 - r0;
        bytes += (r1 - r0) * (int64_t)es;
    }
    allgatherv_inplace(c, (const char *)y + es * (size_t)start[(size_t)c->rank], y, cnt, start, es, st);
    tm.mark(0);
    if (sh) {
        BMSP_HIP(hipStreamSynchronize(st));
        double t_us[10] = {0};
        tm.collect(t_us);
        sh->world = P; sh->rank = c->rank;
        sh->panel_block_row_begin = A->shard_bounds[(size_t)c->rank]; sh->panel_block_row_end = A->shard_bounds[(size_t)c->rank + 1];
        sh->panel_tasks = 0; sh->exchange_bytes = bytes; sh->exchange_us = t_us[0];
    }
}

}  // namespace bmsp
