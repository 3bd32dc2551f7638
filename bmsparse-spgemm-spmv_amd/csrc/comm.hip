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
// SpMV: block-row panels balanced by stored values, x replicated, every rank's sweep writes ITS rows of the full-length y (and no
// other row) and the y slices are broadcast in place the same way.
// The exchange is split into layout (host arithmetic, shared) and transport (RCCL, or an in-process loopback that lets a one-GPU
// box run every P > 1 branch): see "the exchange" below.
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
#include <sys/stat.h>
#include <unistd.h>
#include <array>
#include <ctime>
#include <vector>

namespace bmsp {
void partition_rows(bmsp_matrix_s *A, bmsp_matrix_s *B, int parts, int64_t *bounds, hipStream_t st, uint64_t *total_out);
bmsp_matrix_s *row_panel(bmsp_matrix_s *m, int64_t rb, int64_t re, hipStream_t st);
bmsp_matrix_s *concat_panels(int num_rows, int num_cols, int parts, const int64_t *block_nums, const int64_t *nnzs, uint64_t *const *d_keys, uint64_t *const *d_bmps,
                             uint64_t *const *d_offsets, void *const *d_values, bmsp_dtype dtype, hipStream_t st);
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

// ---- the exchange: layout (shared by every transport) and the two transports ----------------------------------------------------------
// A transport only moves bytes: `gather_sizes` makes every panel's (blocks, values) pair known on every rank, `slices` lands every
// panel's `count[r]` elements at dst + elem * start[r].  Everything else -- the size gather's result -> slice starts (shard_layout) ->
// allocation of the whole C -> offset re-basing (AddU64 for r > 0) -> terminal offset -- is the same code for both:
//   * RCCL (one process per GPU): sizes by ncclAllGather, slices by ncclGroupStart .. P x ncclBroadcast(root = r) .. ncclGroupEnd;
//   * loopback (ONE process, ONE device, any P): the P panel products are computed one after another on this device and a "broadcast"
//     is a hipMemcpyAsync of the panel into its final slice.  It exists so that every P > 1 branch of the sharded operators runs on a
//     one-GPU box (tests/test_gpu_parity.py: P in {2, 3, 8}, incl. an empty panel) -- what it does not cover is RCCL itself.
// `src[r]` is the panel of rank r where this process holds it (RCCL: only r == rank; loopback: every r).
void exchange_sizes(bmsp_comm_s *c, const std::vector<std::array<int64_t, 2>> &local, std::vector<int64_t> &sizes, hipStream_t st)
{
    const int P = c->world;
    sizes.assign((size_t)2 * P, 0);
    if (c->loopback) {
        for (int r = 0; r < P; r++) { sizes[(size_t)2 * r] = local[(size_t)r][0]; sizes[(size_t)2 * r + 1] = local[(size_t)r][1]; }
        return;
    }
    DevBuf<int64_t> d_sizes((size_t)2 * P), d_mine(2);
    BMSP_HIP(hipMemcpyAsync(d_mine.p, local[(size_t)c->rank].data(), 16, hipMemcpyHostToDevice, st));
    BMSP_NCCL(rccl().AllGather(d_mine.p, d_sizes.p, 2, ncclInt64, (ncclComm_t)c->comm, st));
    BMSP_HIP(hipMemcpyAsync(sizes.data(), d_sizes.p, 8 * sizes.size(), hipMemcpyDeviceToHost, st));
    BMSP_HIP(hipStreamSynchronize(st));
}

void exchange_slices(bmsp_comm_s *c, const std::vector<const void *> &src, void *dst, const int64_t *count, const int64_t *start, size_t elem,
                     hipStream_t st)
{
    if (c->loopback) {
        for (int r = 0; r < c->world; r++) {
            if (count[r] == 0) continue;
            char *slice = (char *)dst + elem * (size_t)start[r];
            if ((const void *)slice != src[(size_t)r]) BMSP_HIP(hipMemcpyAsync(slice, src[(size_t)r], elem * (size_t)count[r], hipMemcpyDeviceToDevice, st));
        }
        return;
    }
    Rccl &R = rccl();
    BMSP_NCCL(R.GroupStart());
    for (int r = 0; r < c->world; r++) {
        if (count[r] == 0) continue;
        char *slice = (char *)dst + elem * (size_t)start[r];
        BMSP_NCCL(R.Broadcast(r == c->rank ? src[(size_t)r] : (const void *)slice, slice, elem * (size_t)count[r], ncclChar, r, (ncclComm_t)c->comm, st));
    }
    BMSP_NCCL(R.GroupEnd());
}

// the panels this process computes: its own rank, or all of them in loopback
std::vector<int> local_ranks(const bmsp_comm_s *c)
{
    std::vector<int> v;
    if (c->loopback)
        for (int r = 0; r < c->world; r++) v.push_back(r);
    else
        v.push_back(c->rank);
    return v;
}

}  // namespace

// slice layout of the exchanged C: panel r's blocks land at block_start[r], its values at value_start[r] (exclusive sums; entry
// `parts` = the totals).  Host arithmetic only: exported as bmsp_shard_layout so that the world-size-2 CPU test drives THIS code.
void shard_layout(int parts, const int64_t *block_nums, const int64_t *nnzs, int64_t *block_start, int64_t *value_start)
{
    if (parts < 1) fail(BMSP_ERR_INVALID, "parts must be >= 1");
    block_start[0] = 0; value_start[0] = 0;
    for (int r = 0; r < parts; r++) {
        if (block_nums[r] < 0 || nnzs[r] < 0) fail(BMSP_ERR_INVALID, "negative panel size");
        block_start[r + 1] = block_start[r] + block_nums[r];
        value_start[r + 1] = value_start[r] + nnzs[r];
    }
}

// row slices of the exchanged y: panel r = block-rows [bounds[r], bounds[r+1]) -> rows [row_start[r], row_start[r] + row_count[r])
void shard_row_slices(int num_rows, int parts, const int64_t *bounds, int64_t *row_start, int64_t *row_count)
{
    if (parts < 1) fail(BMSP_ERR_INVALID, "parts must be >= 1");
    for (int r = 0; r < parts; r++) {
        if (bounds[r] < 0 || bounds[r + 1] < bounds[r]) fail(BMSP_ERR_INVALID, "panel bounds must ascend");
        const int64_t r0 = std::min<int64_t>(bounds[r] * 8, num_rows), r1 = std::min<int64_t>(bounds[r + 1] * 8, num_rows);
        row_start[r] = r0; row_count[r] = r1 - r0;
    }
}

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

bmsp_comm_s *comm_init_loopback(int world)
{
    if (world < 1) fail(BMSP_ERR_INVALID, "world must be >= 1");
    bmsp_comm_s *c = new bmsp_comm_s();
    c->world = world; c->rank = 0; c->loopback = 1;
    BMSP_HIP(hipGetDevice(&c->device));
    return c;
}

// BMSP_WORLD / BMSP_RANK / BMSP_COMM_FILE: the rendezvous of the drop-in executables (one process per GPU, started by any
// launcher).  Rank 0 removes whatever a crashed run left at the path, then publishes {magic, nonce, 128-byte id} (tmp + rename);
// the others wait for a file that carries the magic, THEIR nonce (BMSP_COMM_NONCE, a number the launcher gives every rank of one
// run; 0 when unset) and -- when no nonce is set -- is not older than 120 s before this process looked first (a file left by an
// earlier run is then refused instead of being joined: its id is dead and ncclCommInitRank would block on it).
namespace {
struct Rendezvous {
    char magic[8];
    uint64_t nonce;
    char id[128];
};
const char kRdvMagic[8] = {'B', 'M', 'S', 'P', 'c', 'o', 'm', '1'};
}  // namespace

bmsp_comm_s *comm_init_from_env()
{
    const char *w = getenv("BMSP_WORLD"), *r = getenv("BMSP_RANK"), *f = getenv("BMSP_COMM_FILE"), *ne = getenv("BMSP_COMM_NONCE");
    if (!w || !r) fail(BMSP_ERR_INVALID, "BMSP_WORLD and BMSP_RANK must be set");
    const int world = atoi(w), rank = atoi(r);
    if (world < 1 || rank < 0 || rank >= world) fail(BMSP_ERR_INVALID, "BMSP_RANK %d outside BMSP_WORLD %d", rank, world);
    Rendezvous rv{};
    if (world == 1) {
        comm_unique_id(rv.id);
        return comm_init(rv.id, 1, 0);
    }
    if (!f) fail(BMSP_ERR_INVALID, "BMSP_COMM_FILE (a path every rank can reach) must be set when BMSP_WORLD > 1");
    const uint64_t nonce = ne ? strtoull(ne, nullptr, 0) : 0ull;
    const std::string path = f, tmp = path + ".tmp";
    if (rank == 0) {
        (void)unlink(path.c_str());  // a leftover of a crashed run must never be read as this run's id
        memcpy(rv.magic, kRdvMagic, 8);
        rv.nonce = nonce;
        comm_unique_id(rv.id);
        FILE *fp = fopen(tmp.c_str(), "wb");
        if (!fp || fwrite(&rv, 1, sizeof rv, fp) != sizeof rv) fail(BMSP_ERR_IO, "cannot write %s", tmp.c_str());
        fclose(fp);
        if (rename(tmp.c_str(), path.c_str()) != 0) fail(BMSP_ERR_IO, "cannot publish %s", path.c_str());
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        const time_t first_look = time(nullptr);
        for (;;) {
            FILE *fp = fopen(path.c_str(), "rb");
            if (fp) {
                const size_t n = fread(&rv, 1, sizeof rv, fp);
                struct stat sb{};
                const bool have_stat = fstat(fileno(fp), &sb) == 0;
                fclose(fp);
                const bool fresh = ne != nullptr || !have_stat || sb.st_mtime + 120 >= first_look;
                if (n == sizeof rv && memcmp(rv.magic, kRdvMagic, 8) == 0 && rv.nonce == nonce && fresh) break;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                fail(BMSP_ERR_IO, "timed out waiting for a rendezvous file of this run at %s (stale or foreign files are refused)", path.c_str());
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    bmsp_comm_s *c = comm_init(rv.id, world, rank);
    if (rank == 0) (void)unlink(path.c_str());  // every rank has read it: the collective init above has completed
    return c;
}

void comm_free(bmsp_comm_s *c)
{
    if (!c) return;
    if (c->xstream) (void)hipStreamDestroy((hipStream_t)c->xstream);
    if (c->comm) (void)rccl().CommDestroy((ncclComm_t)c->comm);
    delete c;
}

namespace {
struct CopyU64 {
    const uint64_t *src;
    uint64_t *dst;
    __device__ void operator()(uint64_t i) const { dst[i] = src[i]; }
};
// one panel product (a sub-panel when the panel exceeds one task list), with the pair's row-merge hint carried over the views
bmsp_matrix_s *panel_product(bmsp_matrix_s *A, bmsp_matrix_s *B, int64_t rb, int64_t re, int mode, int tc_version, int verbose, hipStream_t st, bmsp_spgemm_stats *one)
{
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> view(row_panel(A, rb, re, st), free_matrix);
    rm_hint_inherit(view.get(), A);
    bmsp_matrix_s *cp_raw = nullptr;
    try {
        spgemm(view.get(), B, &cp_raw, mode, tc_version, verbose, st, one);
    } catch (const TaskRangeExceeded &) {  // a panel beyond one task list: run it in sub-panels (same answer)
        spgemm_paneled(view.get(), B, &cp_raw, mode, tc_version, verbose, st, one);
    }
    rm_hint_merge(A, view.get());
    return cp_raw;
}
void add_stats(bmsp_spgemm_stats &ps, const bmsp_spgemm_stats &one)
{
    ps.task_list_size += one.task_list_size; ps.bmp_reduction += one.bmp_reduction; ps.surviving_tasks += one.surviving_tasks;
    ps.c_blocks += one.c_blocks; ps.c_nnz += one.c_nnz;
    for (int i = 0; i < 10; i++) ps.t_us[i] += one.t_us[i];
    ps.sort_path = one.sort_path; ps.mac_kernel = one.mac_kernel; ps.mac_variant = one.mac_variant; ps.sort_long = std::max(ps.sort_long, one.sort_long);
}
}  // namespace

// The sharded product.  gather = 1: every rank returns the whole C.  The block-rows of A are cut into ROUNDS x P panels balanced by
// candidate-task count; rank r multiplies panels r, P + r, 2 P + r, ...  (round k = panels k P ... k P + P - 1, a contiguous stretch of
// C's rows that precedes every later round's).  After round k the P panel sizes are gathered (one tiny ncclAllGather), which fixes where
// the round's panels land in the whole C -- rounds are laid out in order, so nothing later can move them -- and the round's four arrays
// are broadcast straight into their final slices ON A SECOND STREAM while round k + 1 multiplies on the caller's: the exchange of all but
// the last round hides behind compute (xGMI copies against kernels; loopback: the copy engine against kernels, measurable on one GPU).
// The whole C is allocated after round 0 from its sizes (x rounds x 1.25) and grown by copy if a later round does not fit.
// gather = 0 ("owner keeps", SURVEY 8(e): "If only the owner needs C, skip the gather"): P panels, no exchange, *Cout = this rank's panel
// with global keys (loopback: all P panels concatenated -- this process owns every one).
void spgemm_sharded(bmsp_comm_s *c, bmsp_matrix_s *A, bmsp_matrix_s *B, bmsp_matrix_s **Cout, int mode, int tc_version, int verbose,
                    hipStream_t st, bmsp_spgemm_stats *stats, bmsp_shard_stats *sh, int gather, int rounds)
{
    if (!c || !A || !B || !Cout) fail(BMSP_ERR_INVALID, "null argument");
    if (rounds < 0 || rounds > 64) fail(BMSP_ERR_INVALID, "rounds must be 0 (library's choice) .. 64");
    const int P = c->world;
    typedef std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> MatPtr;
    const bmsp_dtype cdt = A->dtype == BMSP_F64 ? BMSP_F64 : BMSP_F32;
    const size_t es = dtype_size(cdt);
    bmsp_spgemm_stats ps{};  // this rank's panels (loopback: summed over all panels, like the paneled single-GPU product)
    if (sh) *sh = bmsp_shard_stats{};

    if (!gather) {
        std::vector<int64_t> bounds((size_t)P + 1);
        partition_rows(A, B, P, bounds.data(), st, nullptr);
        std::vector<bmsp_matrix_s *> mine;
        auto cleanup = [&]() { for (bmsp_matrix_s *m : mine) free_matrix(m); mine.clear(); };
        try {
            for (int r : local_ranks(c)) {
                bmsp_spgemm_stats one{};
                mine.push_back(panel_product(A, B, bounds[(size_t)r], bounds[(size_t)r + 1], mode, tc_version, verbose, st, &one));
                add_stats(ps, one);
            }
            if (mine.size() == 1) {
                *Cout = mine[0];
                mine.clear();
            } else {
                std::vector<int64_t> bn, nz;
                std::vector<uint64_t *> k, b, o;
                std::vector<void *> v;
                for (bmsp_matrix_s *m : mine) { bn.push_back(m->block_num); nz.push_back(m->nnz); k.push_back(m->keys); b.push_back(m->bmps); o.push_back(m->offsets); v.push_back(m->values); }
                *Cout = concat_panels(A->num_rows, B->num_cols, (int)mine.size(), bn.data(), nz.data(), k.data(), b.data(), o.data(), v.data(), cdt, st);
            }
        } catch (...) {
            cleanup();
            throw;
        }
        cleanup();
        if (stats) *stats = ps;
        if (sh) {
            sh->world = P; sh->rank = c->rank;
            sh->panel_block_row_begin = bounds[(size_t)c->rank]; sh->panel_block_row_end = bounds[(size_t)c->rank + 1];
            sh->panel_tasks = ps.surviving_tasks; sh->rounds = 1; sh->gathered = 0;
        }
        return;
    }

    const char *re = getenv("BMSP_SHARD_ROUNDS");
    const int K = rounds ? rounds : (re ? std::max(1, std::min(64, atoi(re))) : (P > 1 ? 4 : 1));
    const int Q = K * P;
    std::vector<int64_t> bounds((size_t)Q + 1);
    partition_rows(A, B, Q, bounds.data(), st, nullptr);
    if (!c->xstream) BMSP_HIP(hipStreamCreateWithFlags((hipStream_t *)&c->xstream, hipStreamNonBlocking));
    const hipStream_t xs = (hipStream_t)c->xstream;
    std::vector<MatPtr> cp;
    for (int q = 0; q < Q; q++) cp.emplace_back(nullptr, free_matrix);
    MatPtr C(new bmsp_matrix_s(), free_matrix);
    C->num_rows = A->num_rows; C->num_cols = B->num_cols; C->dtype = cdt; C->transposed = 0;
    int64_t cap_b = 0, cap_z = 0, base_b = 0, base_z = 0;
    std::vector<hipEvent_t> evs;
    auto new_event = [&](hipStream_t s) { hipEvent_t e; BMSP_HIP(hipEventCreate(&e)); evs.push_back(e); BMSP_HIP(hipEventRecord(e, s)); return e; };
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ex_spans;
    hipEvent_t ev_compute_done = nullptr;
    try {
        for (int k = 0; k < K; k++) {
            std::vector<std::array<int64_t, 2>> local((size_t)P, std::array<int64_t, 2>{0, 0});
            for (int r : local_ranks(c)) {
                const int q = k * P + r;
                bmsp_spgemm_stats one{};
                cp[(size_t)q].reset(panel_product(A, B, bounds[(size_t)q], bounds[(size_t)q + 1], mode, tc_version, verbose, st, &one));
                local[(size_t)r] = {cp[(size_t)q]->block_num, cp[(size_t)q]->nnz};
                add_stats(ps, one);
            }
            if (k == K - 1) ev_compute_done = new_event(st);
            // sizes of the round's panels -> where they land (rounds are laid out one after the other)
            std::vector<int64_t> sizes;
            exchange_sizes(c, local, sizes, xs);  // (every RCCL call of the product on the exchange stream: one order on every rank)
            std::vector<int64_t> nb((size_t)P), nz((size_t)P), b0((size_t)P + 1, 0), z0((size_t)P + 1, 0);
            for (int r = 0; r < P; r++) { nb[(size_t)r] = sizes[(size_t)2 * r]; nz[(size_t)r] = sizes[(size_t)2 * r + 1]; }
            shard_layout(P, nb.data(), nz.data(), b0.data(), z0.data());
            const int64_t need_b = base_b + b0[(size_t)P], need_z = base_z + z0[(size_t)P];
            if (need_b > cap_b || need_z > cap_z || !C->keys) {
                // first round: the whole C from this round's sizes; later: a round that does not fit (the earlier rounds' slices move)
                const int left = K - k;
                const int64_t nb_cap = std::max<int64_t>(need_b, base_b + (int64_t)((double)b0[(size_t)P] * left * 1.25) + 1024);
                const int64_t nz_cap = std::max<int64_t>(need_z, base_z + (int64_t)((double)z0[(size_t)P] * left * 1.25) + 1024);
                BMSP_HIP(hipStreamSynchronize(xs));  // the broadcasts in flight write the arrays that move
                uint64_t *nk = (uint64_t *)pool_alloc(8 * (size_t)nb_cap), *nbm = (uint64_t *)pool_alloc(8 * (size_t)nb_cap), *no = (uint64_t *)pool_alloc(8 * ((size_t)nb_cap + 1));
                void *nv = pool_alloc(es * (size_t)(nz_cap ? nz_cap : 1));
                if (base_b) {
                    BMSP_HIP(hipMemcpyAsync(nk, C->keys, 8 * (size_t)base_b, hipMemcpyDeviceToDevice, st));
                    BMSP_HIP(hipMemcpyAsync(nbm, C->bmps, 8 * (size_t)base_b, hipMemcpyDeviceToDevice, st));
                    BMSP_HIP(hipMemcpyAsync(no, C->offsets, 8 * (size_t)base_b, hipMemcpyDeviceToDevice, st));
                }
                if (base_z) BMSP_HIP(hipMemcpyAsync(nv, C->values, es * (size_t)base_z, hipMemcpyDeviceToDevice, st));
                BMSP_HIP(hipStreamSynchronize(st));
                pool_free(C->keys); pool_free(C->bmps); pool_free(C->offsets); pool_free(C->values);
                C->keys = nk; C->bmps = nbm; C->offsets = no; C->values = nv;
                cap_b = nb_cap; cap_z = nz_cap;
            }
            // the round's panels straight into their final slices, on the exchange stream (the panel products are complete: spgemm returns
            // synchronised); the next round's products run meanwhile
            hipEvent_t e0 = new_event(xs);
            std::vector<const void *> sk((size_t)P, nullptr), sb((size_t)P, nullptr), so((size_t)P, nullptr), sv((size_t)P, nullptr);
            std::vector<int64_t> bs((size_t)P), zs((size_t)P);
            for (int r = 0; r < P; r++) {
                const bmsp_matrix_s *m = cp[(size_t)(k * P + r)].get();
                if (m) { sk[(size_t)r] = m->keys; sb[(size_t)r] = m->bmps; so[(size_t)r] = m->offsets; sv[(size_t)r] = m->values; }
                bs[(size_t)r] = base_b + b0[(size_t)r]; zs[(size_t)r] = base_z + z0[(size_t)r];
            }
            exchange_slices(c, sk, C->keys, nb.data(), bs.data(), 8, xs);
            exchange_slices(c, sb, C->bmps, nb.data(), bs.data(), 8, xs);
            exchange_slices(c, so, C->offsets, nb.data(), bs.data(), 8, xs);  // block_num entries per panel (the terminal one is rebuilt)
            exchange_slices(c, sv, C->values, nz.data(), zs.data(), es, xs);
            // a panel's offsets count from its own first value: re-base by the values in front of it
            for (int r = 0; r < P; r++)
                if (nb[(size_t)r] && zs[(size_t)r]) device_for_each(AddU64{C->offsets + bs[(size_t)r], (uint64_t)zs[(size_t)r]}, (uint64_t)nb[(size_t)r], xs);
            ex_spans.emplace_back(e0, new_event(xs));
            base_b = need_b; base_z = need_z;
        }
        const uint64_t term = (uint64_t)base_z;
        BMSP_HIP(hipStreamSynchronize(xs));
        hipEvent_t ev_all_done = new_event(xs);
        BMSP_HIP(hipMemcpyAsync(C->offsets + base_b, &term, 8, hipMemcpyHostToDevice, st));
        BMSP_HIP(hipStreamSynchronize(st));
        BMSP_HIP(hipEventSynchronize(ev_all_done));
        C->block_num = base_b; C->nnz = base_z;
        double ex_us = 0.0, exposed_us = 0.0;
        for (auto &sp : ex_spans) {
            float ms = 0.f;
            BMSP_HIP(hipEventElapsedTime(&ms, sp.first, sp.second));
            ex_us += (double)ms * 1e3;
        }
        if (!ex_spans.empty() && ev_compute_done) {
            float ms = 0.f;
            BMSP_HIP(hipEventElapsedTime(&ms, ev_compute_done, ex_spans.back().second));
            exposed_us = std::max(0.0, (double)ms * 1e3);
        }
        if (stats) *stats = ps;
        if (sh) {
            sh->world = P; sh->rank = c->rank;
            sh->panel_block_row_begin = bounds[(size_t)c->rank]; sh->panel_block_row_end = bounds[(size_t)c->rank + 1];
            sh->panel_tasks = ps.surviving_tasks;
            sh->exchange_bytes = 24 * base_b + (int64_t)es * base_z;
            sh->exchange_us = ex_us;
            sh->exchange_exposed_us = std::min(exposed_us, ex_us);
            sh->exchange_hidden_frac = ex_us > 0.0 ? 1.0 - std::min(exposed_us, ex_us) / ex_us : 0.0;
            sh->rounds = K; sh->gathered = 1;
        }
    } catch (...) {
        (void)hipStreamSynchronize(xs);
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        throw;
    }
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    *Cout = C.release();
}

// block-row bounds balanced by stored values
static void spmv_bounds(bmsp_matrix_s *A, int P, std::vector<int64_t> &bounds, hipStream_t st)
{
    ensure_rowptr(A, st);
    const int64_t nbr = A->num_block_rows();
    std::vector<uint32_t> rp((size_t)nbr + 1);
    std::vector<uint64_t> off((size_t)A->block_num + 1);
    BMSP_HIP(hipStreamSynchronize(st));
    copy_d2h_staged(rp.data(), A->rowptr, 4 * rp.size());
    copy_d2h_staged(off.data(), A->offsets, 8 * off.size());
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
    const size_t es = A->dtype == BMSP_F64 ? 8 : 4;
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
    std::vector<int64_t> cnt((size_t)P), start((size_t)P);
    shard_row_slices(A->num_rows, P, A->shard_bounds.data(), start.data(), cnt.data());
    int64_t bytes = 0;
    for (int r = 0; r < P; r++) bytes += cnt[(size_t)r] * (int64_t)es;
    // every panel's sweep writes ITS rows of u and nothing else (rows outside the panel are another rank's to deliver)
    std::vector<const void *> src((size_t)P, nullptr);
    std::vector<DevBuf<char>> scratch;  // loopback: the other "ranks'" result vectors (poisoned, so a row nobody delivers shows)
    if (c->loopback) {
        scratch.resize((size_t)P);
        for (int r = 0; r < P; r++) {
            std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> view(row_panel(A, A->shard_bounds[(size_t)r], A->shard_bounds[(size_t)r + 1], st), free_matrix);
            scratch[(size_t)r].alloc(es * (size_t)std::max(1, A->num_rows));
            BMSP_HIP(hipMemsetAsync(scratch[(size_t)r].p, 0xff, es * (size_t)A->num_rows, st));
            spmv(view.get(), x, scratch[(size_t)r].p, variant, st, start[(size_t)r], start[(size_t)r] + cnt[(size_t)r]);
            BMSP_HIP(hipStreamSynchronize(st));  // the view (and its plan) goes away below
            src[(size_t)r] = scratch[(size_t)r].p + es * (size_t)start[(size_t)r];
        }
    } else {
        const int r = c->rank;
        spmv(A->shard_view, x, y, variant, st, start[(size_t)r], start[(size_t)r] + cnt[(size_t)r]);
        src[(size_t)r] = (const char *)y + es * (size_t)start[(size_t)r];
    }
    StageTimer tm(st, sh != nullptr);
    tm.mark(-1);
    exchange_slices(c, src, y, cnt.data(), start.data(), es, st);
    tm.mark(0);
    if (c->loopback) BMSP_HIP(hipStreamSynchronize(st));  // scratch vectors are released on return
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

BMSP_DEFINE_WARM(comm)
