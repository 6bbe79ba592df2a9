// fmd_ovlp_dist.hip -- overlap discovery on N GPUs: the two exchange steps of the path and the transport behind the C ABI.
//
// The reference joins its N workers over ONE shared index for nothing (pthreads, unitig.c:394-404) and emits results as they are
// found (unitig.c:353-354).  N GPUs hold N replicas of the index; two things move:
//   (a) the finished rows, to the rank that runs the walk -- in PIECES: pass 2 runs batch by batch in the sorted order, and while
//       batch p + 1 is computed the rows of batch p are packed on a second stream (fmd_ovlp_pack_rows_dev: ~137 bytes per strand
//       with its id) and sent to the root, every peer over its own xGMI link (direct sends, no ring).  Only the LAST piece's pack +
//       transfer cannot hide under compute;
//   (b) optionally the parked strands between the two passes (64 bytes each), all-to-all by minimizer key range: pass 1 is sharded
//       by id (the reference's worker interleave, unitig.c:333, 398-399), pass 2 by KEY, so that all strands of a genomic window
//       meet on one GPU and the rank blocks they share are fetched once -- a rank then sees the coverage of the whole read set on
//       1/N of the genome instead of 1/N of the coverage everywhere.
// The transport is fmd_comm_t (include/fmd_hip.h): RCCL here (librccl is loaded on first use, so that a single-GPU host never needs
// it), anything else through the two callbacks.
#include <dlfcn.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>
#include <vector>
#include "fmd_kernel_common.h"
#include "fmd_internal.h"

// (fmd_ovlp_sort.hip)
size_t fmd_park_sort_temp_bytes(size_t n);
int fmd_park_sort(hipStream_t st, size_t n, const FmdWalkPark *park, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *tmp, size_t tmp_bytes);

static inline double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------------ RCCL transport
namespace {
typedef int ncclResult_t_;
typedef struct { char internal[128]; } ncclUniqueId_;
struct Rccl {
    void *so = nullptr;
    ncclResult_t_ (*GetVersion)(int *) = nullptr;
    ncclResult_t_ (*GetUniqueId)(ncclUniqueId_ *) = nullptr;
    ncclResult_t_ (*CommInitRank)(void **, int, ncclUniqueId_, int) = nullptr;
    ncclResult_t_ (*CommDestroy)(void *) = nullptr;
    ncclResult_t_ (*CommCount)(void *, int *) = nullptr;
    ncclResult_t_ (*GroupStart)() = nullptr;
    ncclResult_t_ (*GroupEnd)() = nullptr;
    ncclResult_t_ (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    ncclResult_t_ (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    ncclResult_t_ (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t_) = nullptr;
    bool ok = false;
};
const int kNcclUint8 = 1;   // ncclUint8 (rccl.h ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1)
Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char *names[] = {getenv("FMD_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) { if (nm && *nm && (r.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break; }
    if (!r.so) return r;
#define RSYM(f) *(void **)(&r.f) = dlsym(r.so, "nccl" #f)
    RSYM(GetVersion); RSYM(GetUniqueId); RSYM(CommInitRank); RSYM(CommDestroy); RSYM(CommCount); RSYM(GroupStart); RSYM(GroupEnd); RSYM(Send); RSYM(Recv); RSYM(AllGather); RSYM(GetErrorString);
#undef RSYM
    r.ok = r.GetVersion && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllGather;
    return r;
}
struct RcclCtx { void *comm; int device; };
int rccl_fail(ncclResult_t_ e, const char *what)
{
    Rccl &r = rccl();
    fprintf(stderr, "[E::fmd_comm_rccl] %s: %s\n", what, r.GetErrorString ? r.GetErrorString(e) : "error");
    return FMD_E_HIP;
}
int rccl_allgather(void *ctx, void *stream, const void *d_send, void *d_recv, size_t bytes)
{
    RcclCtx *c = (RcclCtx *)ctx;
    const ncclResult_t_ e = rccl().AllGather(d_send, d_recv, bytes, kNcclUint8, c->comm, (hipStream_t)stream);
    return e ? rccl_fail(e, "ncclAllGather") : FMD_OK;
}
int rccl_exchange(void *ctx, void *stream, int n_ops, const fmd_comm_op_t *ops)
{
    RcclCtx *c = (RcclCtx *)ctx;
    Rccl &r = rccl();
    if (n_ops <= 0) return FMD_OK;
    ncclResult_t_ e = r.GroupStart();
    if (e) return rccl_fail(e, "ncclGroupStart");
    for (int i = 0; i < n_ops && !e; ++i) {
        if (ops[i].bytes == 0) continue;
        e = ops[i].is_recv ? r.Recv(ops[i].d_ptr, ops[i].bytes, kNcclUint8, ops[i].peer, c->comm, (hipStream_t)stream)
                           : r.Send(ops[i].d_ptr, ops[i].bytes, kNcclUint8, ops[i].peer, c->comm, (hipStream_t)stream);
    }
    const ncclResult_t_ e2 = r.GroupEnd();
    if (e) return rccl_fail(e, "ncclSend / ncclRecv");
    return e2 ? rccl_fail(e2, "ncclGroupEnd") : FMD_OK;
}
void rccl_destroy(void *ctx)
{
    RcclCtx *c = (RcclCtx *)ctx;
    if (c) { if (c->comm) rccl().CommDestroy(c->comm); delete c; }
}
}   // namespace

extern "C" int fmd_comm_rccl_version(void)
{
    Rccl &r = rccl();
    int v = 0;
    if (!r.ok || r.GetVersion(&v)) return 0;
    return v;
}
extern "C" int fmd_comm_rccl_unique_id(uint8_t id[FMD_COMM_ID_BYTES])
{
    Rccl &r = rccl();
    if (!id) return FMD_E_ARG;
    if (!r.ok) return FMD_E_NODEV;
    ncclUniqueId_ u;
    const ncclResult_t_ e = r.GetUniqueId(&u);
    if (e) return rccl_fail(e, "ncclGetUniqueId");
    memcpy(id, u.internal, FMD_COMM_ID_BYTES);
    return FMD_OK;
}
extern "C" int fmd_comm_rccl_init(int device, int rank, int world, const uint8_t id[FMD_COMM_ID_BYTES], fmd_comm_t **out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return FMD_E_ARG;
    *out = nullptr;
    Rccl &r = rccl();
    if (!r.ok) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    ncclUniqueId_ u;
    memcpy(u.internal, id, FMD_COMM_ID_BYTES);
    void *comm = nullptr;
    const ncclResult_t_ e = r.CommInitRank(&comm, world, u, rank);
    if (e) return rccl_fail(e, "ncclCommInitRank");
    fmd_comm_t *c = (fmd_comm_t *)calloc(1, sizeof(fmd_comm_t));
    RcclCtx *x = new RcclCtx{comm, device};
    c->rank = rank; c->world = world; c->ctx = x;
    c->allgather = rccl_allgather; c->exchange = rccl_exchange; c->destroy = rccl_destroy;
    *out = c;
    return FMD_OK;
}
// ncclCommCount of a communicator fmd_comm_rccl_init made: the number of ranks RCCL itself says it joined (-1: not such a communicator / no such symbol)
extern "C" int fmd_comm_rccl_count(const fmd_comm_t *c)
{
    if (!c || c->destroy != rccl_destroy || !c->ctx || !rccl().CommCount) return -1;
    int n = -1;
    return rccl().CommCount(((RcclCtx *)c->ctx)->comm, &n) ? -1 : n;
}
extern "C" void fmd_comm_free(fmd_comm_t *c)
{
    if (!c) return;
    if (c->destroy) c->destroy(c->ctx);
    if (c->destroy == rccl_destroy) free(c);
}

// ------------------------------------------------------------------------------------------------ key shard: kernels
// Destination of a strand by its minimizer key: W contiguous key ranges; the two special keys (0xffffffff: the strand ended inside
// the head; 0xfffffffe: no k-mer without an ambiguous base) stay where they are.  A minimizer is the SMALLEST of 17 hashes, so the
// keys crowd towards 0 (and a repeat-rich read set skews them further): the range boundaries are quantiles of the data -- every
// rank's id shard is a uniform sample of the strands, so each takes the p/W quantiles of its own sorted keys, the ranks all-gather
// them and everybody uses the median over the ranks.
__device__ static inline size_t ks_lower(const uint32_t *keys, size_t n, uint32_t v)
{
    size_t lo = 0, hi = n;
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (keys[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ void k_ks_quantiles(size_t n, const uint32_t *__restrict__ keys, int world, uint32_t *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= world) return;
    const size_t n_reg = ks_lower(keys, n, 0xfffffffeu);
    out[p] = p == 0 || n_reg == 0 ? 0u : keys[(size_t)((unsigned __int128)n_reg * (unsigned)p / (unsigned)world)];
}
// counts[p], p < world: strands for rank p = keys in [split[p], split[p + 1]); counts[world]: strands that stay (special keys).
// keys ascending; split[0] = 0, split[world] = 0xfffffffe.
__global__ void k_ks_counts(size_t n, const uint32_t *__restrict__ keys, int world, const uint32_t *__restrict__ split, unsigned long long *__restrict__ counts)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > world) return;
    const size_t a = ks_lower(keys, n, split[p]);
    const size_t b = p < world ? ks_lower(keys, n, split[p + 1]) : n;
    counts[p] = (unsigned long long)(b - a);
}
// the parked strands in key order, each stamped with its sequence id (pad.x/y); a strand that ended inside the head carries what
// its record needs (rank in x0, length in x1: k_ks_unpack rebuilds the record wherever the row ends up).  4 lanes per row.
__global__ void k_ks_gather(size_t n, const uint32_t *__restrict__ order, const FmdWalkPark *__restrict__ park, const uint64_t *__restrict__ ids,
                            const fmd_ovlp_rec_t *__restrict__ rec, FmdWalkPark *__restrict__ send)
{
    const size_t step = (size_t)gridDim.x * (blockDim.x >> 2);
    const int l4 = threadIdx.x & 3;
    for (size_t t = (size_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2); t < n; t += step) {
        const size_t row = order[t];
        uint4 v = ((const uint4 *)(park + row))[l4];
        const uint4 a = l4 == 0 ? v : ((const uint4 *)(park + row))[0];
        const bool ended = a.x == 0xffffffffu && a.y == 0xffffffffu;
        if (l4 == 3) { const uint64_t id = ids[row]; v = make_uint4((uint32_t)id, (uint32_t)(id >> 32), v.z, v.w); }   // (pad.z / pad.w: the bases and the depth of a strand parked beyond 32 bases, fmd_ovlp.hip)
        if (ended && l4 == 0) { const uint64_t rk = rec[row].rank; v.z = (uint32_t)rk; v.w = (uint32_t)(rk >> 32); }
        if (ended && l4 == 1) { v.x = (uint32_t)rec[row].len; v.y = 0; }
        ((uint4 *)(send + t))[l4] = v;
    }
}
// after the exchange: ids of the rows this rank now owns; final records of the ones that ended inside the head
__global__ void k_ks_unpack(size_t m, const FmdWalkPark *__restrict__ park, uint64_t *__restrict__ ids, fmd_ovlp_rec_t *__restrict__ rec)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += step) {
        const FmdWalkPark *p = park + j;
        ids[j] = (uint64_t)p->pad.y << 32 | p->pad.x;
        if (p->k == ~0ull) {   // what k_ovl_walk<WALK_HEAD> wrote for it on the rank that took it in (fmd_ovlp.hip)
            fmd_ovlp_rec_t o;
            o.rank = p->x0; o.k[0] = o.k[1] = o.k[2] = 0; o.len = (int32_t)p->x1; o.status = FMD_OVLP_SHORT; o.n_ovlp = 0; o.rbeg = -1;
            o.ext_len = 0; o.n_nei = 0; o.flags = 0; o.reserved = 2; o.lfork = 0;
            rec[j] = o;
        }
    }
}
// the parked strands into the order of pass 2 (row t = the t-th strand in key order): everything pass 2 and the pack touch per strand --
// parked state, record, neighbours, sequence row -- is then addressed by position, sequentially, and a piece is a contiguous range of rows
__global__ void k_park_permute(size_t m, const uint32_t *__restrict__ order, const FmdWalkPark *__restrict__ src, FmdWalkPark *__restrict__ dst)
{
    const size_t step = (size_t)gridDim.x * (blockDim.x >> 2);
    const int l4 = threadIdx.x & 3;
    for (size_t t = (size_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2); t < m; t += step) ((uint4 *)(dst + t))[l4] = ((const uint4 *)(src + order[t]))[l4];
}
__global__ void k_iota32(size_t n, uint32_t *v)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) v[i] = (uint32_t)i;
}
__global__ void k_fill_ids64(size_t n, uint64_t first, uint64_t step_, uint64_t *ids)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) ids[i] = first + step_ * i;
}
// root: a piece has arrived at rows [row0, row0 + np) of the table: where each row's variable part is, which row holds which id
__global__ void k_place(size_t np, const uint32_t *__restrict__ pid, const uint64_t *__restrict__ off, uint64_t var_base, uint64_t row0,
                        uint64_t *__restrict__ vaddr, uint32_t *__restrict__ row_of_id, uint64_t n_ids)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < np; t += step) {
        vaddr[t] = var_base + off[t];
        const uint32_t id = pid[t];
        if (id < n_ids) row_of_id[id] = (uint32_t)(row0 + t);
    }
}
static inline unsigned nblk(size_t n, unsigned per) { size_t b = (n + per - 1) / per; if (b > (1u << 20)) b = 1u << 20; return (unsigned)(b ? b : 1); }

// ------------------------------------------------------------------------------------------------ the job object
namespace {
struct DBuf {   // device memory owned by the job
    void *p = nullptr; size_t bytes = 0;
    int need(size_t b) { if (b <= bytes) return FMD_OK; if (p) { hipFree(p); p = nullptr; bytes = 0; } if (hipMalloc(&p, b ? b : 16) != hipSuccess) { (void)hipGetLastError(); return FMD_E_NOMEM; } bytes = b; return FMD_OK; }
    void drop() { if (p) hipFree(p); p = nullptr; bytes = 0; }
};
// FMD_DIST_DRY=1: fmd_ovlp_dist_new says what it allocates (every buffer of a step, the root's table and -- what a step would only allocate as rows arrive -- the
// arena of the variable parts at its usual size), so that the sizes of a configuration can be proven on ONE rank with a stand-in communicator of the
// intended world size before N ranks try (tools/scale_check.py `dry`)
static bool dist_dry() { static const bool v = getenv("FMD_DIST_DRY") && atoi(getenv("FMD_DIST_DRY")) != 0; return v; }
struct HBuf {   // pinned host memory owned by the job
    void *p = nullptr; size_t bytes = 0;
    int need(size_t b) { if (b <= bytes) return FMD_OK; if (p) { hipHostFree(p); p = nullptr; bytes = 0; } if (hipHostMalloc(&p, b ? b : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return FMD_E_NOMEM; } bytes = b; return FMD_OK; }
    void drop() { if (p) hipHostFree(p); p = nullptr; bytes = 0; }
};
struct Arena {  // variable parts at the root: large chunks, allocated when first needed, kept between steps
    bool host = false;
    size_t chunk_bytes = 0;
    std::vector<void *> chunks;
    size_t cur = 0, used = 0;   // chunk in use, bytes used in it
    void reset() { cur = 0; used = 0; }
    void *take(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > chunk_bytes) return nullptr;
        if (cur < chunks.size() && used + bytes > chunk_bytes) { ++cur; used = 0; }
        if (cur >= chunks.size()) {
            void *p = nullptr;
            if ((host ? hipHostMalloc(&p, chunk_bytes, hipHostMallocDefault) : hipMalloc(&p, chunk_bytes)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            chunks.push_back(p);
            cur = chunks.size() - 1; used = 0;
        }
        void *r = (uint8_t *)chunks[cur] + used;
        used += bytes;
        return r;
    }
    // Make sure `n_takes` takes of at most `each` bytes can follow without a new allocation failing: chunks are allocated NOW.  (Every take fits one
    // chunk; a take that does not fit what is left of the current chunk moves on to the next one, so n_takes * each bytes need at most
    // ceil(n_takes / floor(chunk_bytes / each)) chunks beyond the current one's rest.)
    bool ensure(size_t n_takes, size_t each)
    {
        each = (each + 255) & ~(size_t)255;
        if (each == 0 || n_takes == 0) return true;
        if (each > chunk_bytes) return false;
        const size_t per_chunk = chunk_bytes / each;
        const size_t in_cur = cur < chunks.size() ? (chunk_bytes - used) / each : 0;
        const size_t more = n_takes > in_cur ? (n_takes - in_cur + per_chunk - 1) / per_chunk : 0;
        const size_t want = (cur < chunks.size() ? cur + 1 : chunks.size()) + more;
        while (chunks.size() < want) {
            void *p = nullptr;
            if ((host ? hipHostMalloc(&p, chunk_bytes, hipHostMallocDefault) : hipMalloc(&p, chunk_bytes)) != hipSuccess) { (void)hipGetLastError(); return false; }
            chunks.push_back(p);
        }
        return true;
    }
    void shrink_last(size_t taken, size_t kept) { taken = (taken + 255) & ~(size_t)255; kept = (kept + 255) & ~(size_t)255; if (kept < taken && used >= taken - kept) used -= taken - kept; }
    void drop() { for (void *p : chunks) { if (host) hipHostFree(p); else hipFree(p); } chunks.clear(); reset(); }
};
struct Stage { DBuf pid, prec, off, var, vaddr; };   // one piece of one peer on its way through HBM (host table), or this rank's piece on its way out
struct HStage { HBuf pid, prec, off, var; uint64_t rows = 0; };   // root with a row sink: where such a piece lands in pinned host memory, until the sink has had it
}   // namespace

struct fmd_ovlp_dist {
    fmd_dev *h; fmd_comm_t *comm; fmd_ovlp_dist_cfg_t cfg;
    int rank, world, two_pass, pieces, on_host;
    uint32_t stride;
    uint64_t n_home, cap_rows, n_rows;          // strands of pass 1; capacity / number of the rows this rank computes
    size_t piece_max, var_cap_piece;
    DBuf ids_home, ids_loc, park_home, park_send, park_loc, keys, order, iota, rec, nei, seq, work, pack_work, cnt_dev, sizes_dev, split_dev;
    HBuf cnt_host, sizes_host, split_host;
    Stage out[2];                               // this rank's piece on its way out (non-root: set 0; root with a host table: set p & 1)
    std::vector<Stage> in[2];                   // root, host mode: staging per peer, two sets
    int sink = 0;                               // root, cfg.host_table == 2: no table -- every piece goes to cfg.row_sink from ...
    std::vector<HStage> hin[2];                 // ... these, set p & 1 (rows != 0: the sink has not had them yet)
    // the table at the root
    DBuf t_prec, t_ids, t_vaddr, t_row_of, t_off;   // (t_off: per peer (piece_max + 1) offsets of the piece being placed)
    HBuf th_prec, th_ids, th_vaddr, th_row_of;
    Arena var;
    hipStream_t sm = nullptr, s3 = nullptr;
    std::vector<hipEvent_t> done;               // piece p computed (compute stream)
    hipEvent_t ev[8] = {};                      // timing + joins
    hipEvent_t staged[2] = {}, drained[2] = {};
    fmd_ovlp_dist_stats_t last;
    const uint64_t *loc_ids;                    // ids of the rows this rank computed in the last step
};

static uint64_t shard_size(uint64_t n_ids, int r, int world) { return n_ids > (uint64_t)r ? (n_ids - (uint64_t)r + (uint64_t)world - 1) / (uint64_t)world : 0; }
// pieces shrink towards the end (P : P-1 : ... : 1 -- 40 / 30 / 20 / 10 % for four): what cannot hide under compute is the pack + transfer
// of the LAST piece, so that one is the smallest
static uint64_t piece_begin(uint64_t rows, int p, int pieces)
{
    const unsigned __int128 num = (unsigned __int128)(unsigned)p * (unsigned)(2 * pieces - p + 1), den = (unsigned __int128)(unsigned)pieces * (unsigned)(pieces + 1);
    return (uint64_t)((unsigned __int128)rows * num / den);
}

extern "C" void fmd_ovlp_dist_free(fmd_ovlp_dist_t *d)
{
    if (!d) return;
    hipSetDevice(d->h->device);
    hipDeviceSynchronize();
    DBuf *bs[] = {&d->ids_home, &d->ids_loc, &d->park_home, &d->park_send, &d->park_loc, &d->keys, &d->order, &d->iota, &d->rec, &d->nei, &d->seq, &d->work, &d->pack_work, &d->cnt_dev, &d->split_dev,
                  &d->sizes_dev, &d->t_prec, &d->t_ids, &d->t_vaddr, &d->t_row_of, &d->t_off, &d->out[0].pid, &d->out[0].prec, &d->out[0].off, &d->out[0].var, &d->out[0].vaddr, &d->out[1].pid, &d->out[1].prec, &d->out[1].off, &d->out[1].var, &d->out[1].vaddr};
    for (DBuf *b : bs) b->drop();
    for (int k = 0; k < 2; ++k) for (Stage &s : d->in[k]) { s.pid.drop(); s.prec.drop(); s.off.drop(); s.var.drop(); s.vaddr.drop(); }
    for (int k = 0; k < 2; ++k) for (HStage &s : d->hin[k]) { s.pid.drop(); s.prec.drop(); s.off.drop(); s.var.drop(); }
    HBuf *hs[] = {&d->cnt_host, &d->sizes_host, &d->split_host, &d->th_prec, &d->th_ids, &d->th_vaddr, &d->th_row_of};
    for (HBuf *b : hs) b->drop();
    d->var.drop();
    for (hipEvent_t e : d->done) if (e) hipEventDestroy(e);
    for (hipEvent_t e : d->ev) if (e) hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) { if (d->staged[k]) hipEventDestroy(d->staged[k]); if (d->drained[k]) hipEventDestroy(d->drained[k]); }
    if (d->sm) hipStreamDestroy(d->sm);
    if (d->s3) hipStreamDestroy(d->s3);
    (void)hipGetLastError();
    delete d;
}

extern "C" int fmd_ovlp_dist_new(fmd_dev_t *h, fmd_comm_t *comm, const fmd_ovlp_dist_cfg_t *cfg, fmd_ovlp_dist_t **out)
{
    if (!h || !comm || !cfg || !out || !comm->allgather || !comm->exchange || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world) return FMD_E_ARG;
    if (cfg->n_ids == 0 || cfg->n_ids >= 0xffffff00ull || cfg->max_len == 0 || cfg->max_nei == 0 || cfg->min_match < 0 || cfg->root < 0 || cfg->root >= comm->world) return FMD_E_ARG;
    if (cfg->host_table == 2 && comm->rank == cfg->root && !cfg->row_sink) return FMD_E_ARG;   // (a root that keeps no table must say where the rows go)
    *out = nullptr;
    FMD_HIP_TRY(hipSetDevice(h->device));
    fmd_ovlp_dist *d = new fmd_ovlp_dist();
    d->h = h; d->comm = comm; d->cfg = *cfg; d->rank = comm->rank; d->world = comm->world;
    d->stride = 2 * ((cfg->max_len + 3) / 4 * 4);
    d->n_home = shard_size(cfg->n_ids, d->rank, d->world);
    const uint64_t n_max = shard_size(cfg->n_ids, 0, d->world);   // the largest shard
    d->two_pass = fmd_ovlp_two_pass_ok(h, (size_t)(n_max + n_max / 8 + 65536), cfg->min_match, cfg->max_len);
    if (!d->two_pass) d->cfg.key_shard = 0;
    if (d->world == 1) d->cfg.key_shard = 0;
    // key shard: the keys are hashes, a rank's share of them is n / world +- a few sigma; room for 1/8 more, and a collective
    // fall-back to the id shard for the step in which any rank would exceed it (a read set that is one repeat)
    d->cap_rows = d->cfg.key_shard ? d->n_home + d->n_home / 8 + 65536 : d->n_home;
    // pieces: the same number on every rank
    {
        const uint64_t bmax = cfg->batch ? cfg->batch : 20000000ull;
        uint64_t P = cfg->pieces ? cfg->pieces : 4;
        const uint64_t rows_max = d->cfg.key_shard ? n_max + n_max / 8 + 65536 : n_max;
        while (rows_max * 2 / (P + 1) + 2 > bmax && P < 4096) ++P;
        if (!cfg->pieces) while (P > 1 && n_max / P < 262144) --P;
        if (P > 4096) P = 4096;
        d->pieces = (int)P;
        d->piece_max = (size_t)(rows_max * 2 / (P + 1) + 2);   // the first piece is the largest: 2 / (P + 1) of the rows
    }
    d->var_cap_piece = fmd_ovlp_pack_max_bytes(d->piece_max, cfg->max_nei, d->stride);
    int rc = FMD_OK;
    size_t dry_dev = 0, dry_host = 0;
    const bool dry = dist_dry();
#define NEED(buf, bytes) do { if (rc == FMD_OK) { const size_t b_ = (bytes); rc = (buf).need(b_); dry_dev += b_; \
                              if (dry && (b_ >= ((size_t)1 << 28) || rc != FMD_OK)) fprintf(stderr, "[M::fmd_ovlp_dist_new] rank %d/%d: %-24s %8.2f GB of HBM%s\n", d->rank, d->world, #buf, b_ / 1e9, rc == FMD_OK ? "" : "  <- FAILED"); } } while (0)
#define NEEDH(buf, bytes) do { if (rc == FMD_OK) { const size_t b_ = (bytes); rc = (buf).need(b_); dry_host += b_; \
                               if (dry && (b_ >= ((size_t)1 << 28) || rc != FMD_OK)) fprintf(stderr, "[M::fmd_ovlp_dist_new] rank %d/%d: %-24s %8.2f GB of pinned host memory%s\n", d->rank, d->world, #buf, b_ / 1e9, rc == FMD_OK ? "" : "  <- FAILED"); } } while (0)
    NEED(d->ids_home, d->n_home * 8);
    NEED(d->rec, d->cap_rows * sizeof(fmd_ovlp_rec_t));
    NEED(d->nei, d->cap_rows * (size_t)cfg->max_nei * sizeof(fmd_intv_t));
    NEED(d->seq, d->cap_rows * (size_t)d->stride);
    NEED(d->order, d->cap_rows * 4);
    NEED(d->iota, d->cap_rows * 4);
    if (d->two_pass) {
        NEED(d->keys, d->cap_rows * 4);
        NEED(d->park_home, d->n_home * sizeof(FmdWalkPark));
        NEED(d->park_send, d->cap_rows * sizeof(FmdWalkPark));    // the parked strands in the order of pass 2 (key shard: first the send buffer of the all-to-all)
        NEED(d->ids_loc, d->cap_rows * 8);
        if (d->cfg.key_shard) NEED(d->park_loc, d->cap_rows * sizeof(FmdWalkPark));
    }
    {
        size_t wb = fmd_ovlp_work_bytes(d->piece_max, cfg->max_len, cfg->min_match);
        if (d->two_pass) { const size_t hb = fmd_ovlp_head_work_bytes((size_t)d->cap_rows); if (hb > wb) wb = hb; }
        NEED(d->work, wb);
    }
    NEED(d->pack_work, fmd_ovlp_pack_work_bytes(d->piece_max));
    NEED(d->cnt_dev, (size_t)(d->world + 2) * 8 * (size_t)(d->world + 1));     // (a row of the key exchange's counts: W + 1 counts and a status word)
    NEED(d->sizes_dev, 24 * (size_t)(d->world + 1));                           // (a piece: rows, variable bytes, status)
    NEED(d->split_dev, 4 * ((size_t)d->world * (d->world + 2) + 2));
    NEEDH(d->split_host, 4 * ((size_t)d->world * (d->world + 2) + 2));
    NEEDH(d->cnt_host, (size_t)(d->world + 2) * 8 * (size_t)(d->world + 1));
    NEEDH(d->sizes_host, 24 * (size_t)(d->world + 1));
    // where the table lives at the root
    const bool root = d->rank == cfg->root;
    d->on_host = 0;
    if (root) {
        const uint64_t n = cfg->n_ids;
        const size_t table_dev = n * (sizeof(fmd_ovlp_rec_t) + 4 + 8 + 4) + n * (size_t)(cfg->max_nei * 8 + 72);   // fixed parts + a usual variable part
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        d->on_host = cfg->host_table > 0 || (cfg->host_table < 0 && table_dev + ((size_t)6 << 30) > free_b) ? 1 : 0;
        d->sink = cfg->host_table == 2;
        NEED(d->t_row_of, n * 4);
        if (!d->on_host) { NEED(d->t_prec, n * sizeof(fmd_ovlp_rec_t)); NEED(d->t_ids, n * 4); NEED(d->t_vaddr, n * 8); NEED(d->t_off, (size_t)d->world * (d->piece_max + 1) * 8); }
        else {
            if (!d->sink) {
                NEEDH(d->th_prec, n * sizeof(fmd_ovlp_rec_t));
                NEEDH(d->th_ids, n * 4);
                NEEDH(d->th_vaddr, n * 8);
                NEEDH(d->th_row_of, n * 4);
            } else   // no table: two sets of pinned landing buffers, a piece of every peer each; the sink folds set k while set k ^ 1 fills
                for (int k = 0; k < 2 && rc == FMD_OK; ++k) {
                    d->hin[k].resize((size_t)d->world);
                    for (int q = 0; q < d->world && rc == FMD_OK; ++q) {
                        HStage &s = d->hin[k][(size_t)q];
                        NEEDH(s.pid, d->piece_max * 4); NEEDH(s.prec, d->piece_max * sizeof(fmd_ovlp_rec_t)); NEEDH(s.off, (d->piece_max + 1) * 8); NEEDH(s.var, d->var_cap_piece);
                    }
                }
            for (int k = 0; k < 2 && rc == FMD_OK; ++k) {
                d->in[k].resize((size_t)d->world);
                for (int q = 0; q < d->world && rc == FMD_OK; ++q) {
                    Stage &s = d->in[k][(size_t)q];
                    NEED(s.pid, d->piece_max * 4); NEED(s.prec, d->piece_max * sizeof(fmd_ovlp_rec_t)); NEED(s.off, (d->piece_max + 1) * 8); NEED(s.var, d->var_cap_piece); NEED(s.vaddr, d->piece_max * 8);
                }
            }
        }
        d->var.host = d->on_host != 0;
        d->var.chunk_bytes = d->var_cap_piece > ((size_t)256 << 20) ? up256(d->var_cap_piece) : ((size_t)256 << 20);
        if (dry && rc == FMD_OK && !d->sink) {   // the arena as a whole step leaves it: the usual variable part (neighbours + bases, as table_dev above) of every row
            const size_t want = (size_t)n * (size_t)(cfg->max_nei * 8 + 72);
            size_t got = 0;
            while (got < want) {
                d->var.cur = d->var.chunks.size(); d->var.used = 0;
                if (!d->var.take(d->var.chunk_bytes)) { rc = FMD_E_NOMEM; break; }
                got += d->var.chunk_bytes;
            }
            d->var.reset();
            (d->on_host ? dry_host : dry_dev) += got;
            fprintf(stderr, "[M::fmd_ovlp_dist_new] rank %d/%d: %-24s %8.2f GB of %s in %zu chunks%s\n", d->rank, d->world, "variable parts (arena)", got / 1e9,
                    d->on_host ? "pinned host memory" : "HBM", d->var.chunks.size(), rc == FMD_OK ? "" : "  <- FAILED");
        }
    }
    for (int k = 0; k < (root && d->on_host ? 2 : (root ? 0 : 1)); ++k) { NEED(d->out[k].pid, d->piece_max * 4); NEED(d->out[k].prec, d->piece_max * sizeof(fmd_ovlp_rec_t)); NEED(d->out[k].off, (d->piece_max + 1) * 8); NEED(d->out[k].var, d->var_cap_piece); }
#undef NEED
#undef NEEDH
    if (dry) fprintf(stderr, "[M::fmd_ovlp_dist_new] rank %d/%d of a job over %llu ids (%s, %d pieces of at most %zu rows, table %s): %.2f GB of HBM, %.2f GB of pinned host memory: %s\n", d->rank, d->world,
                     (unsigned long long)cfg->n_ids, d->cfg.key_shard ? "key shard" : "id shard", d->pieces, d->piece_max, root ? (d->sink ? "none: rows go to the sink" : d->on_host ? "in pinned host memory" : "in HBM") : "elsewhere",
                     dry_dev / 1e9, dry_host / 1e9, rc == FMD_OK ? "every allocation succeeded" : "OUT OF MEMORY");
    if (rc == FMD_OK) {
        int lo = 0, hi = 0;
        // the second stream at the compute stream's priority: measured on one GPU (tools/dist_one_rank.py), 10^8 strands, pass 2 takes 242 ms beside
        // the pack at equal priority (243 without any pack) and 265 ms when the pack's stream is the device's highest; FMD_DIST_HIGH_PRIORITY is the A/B knob
        bool ok = getenv("FMD_DIST_HIGH_PRIORITY") && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&d->sm, hipStreamNonBlocking, hi) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); ok = hipStreamCreateWithFlags(&d->sm, hipStreamNonBlocking) == hipSuccess; }
        ok = ok && hipStreamCreateWithFlags(&d->s3, hipStreamNonBlocking) == hipSuccess;
        d->done.resize((size_t)d->pieces, nullptr);
        for (int p = 0; p < d->pieces && ok; ++p) ok = hipEventCreateWithFlags(&d->done[(size_t)p], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < 8 && ok; ++k) ok = hipEventCreate(&d->ev[k]) == hipSuccess;
        for (int k = 0; k < 2 && ok; ++k) ok = hipEventCreateWithFlags(&d->staged[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&d->drained[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) { fmd_set_hip_error(hipGetLastError(), "overlap job on N GPUs: streams and events"); rc = FMD_E_HIP; }
    }
    if (rc != FMD_OK) { fmd_ovlp_dist_free(d); return rc; }
    k_fill_ids64<<<nblk(d->n_home, 256), 256>>>(d->n_home, (uint64_t)d->rank, (uint64_t)d->world, (uint64_t *)d->ids_home.p);
    k_iota32<<<nblk(d->cap_rows, 256), 256>>>((size_t)d->cap_rows, (uint32_t *)d->iota.p);
    FMD_HIP_TRY(hipDeviceSynchronize());
    memset(&d->last, 0, sizeof(d->last));
    *out = d;
    return FMD_OK;
}

// A failure that all ranks see (the reference's workers cannot fail apart: unitig.c:394-404 joins them).  A rank that fails on its own -- out of memory,
// a capacity exceeded, a HIP error -- must not leave the step while its peers wait for it in the next collective: it carries the code in `lerr`, skips
// its own work, keeps taking part, and the next all-gather delivers the code to everybody (a status word beside the counts of the key exchange, beside
// the sizes of every piece).  All ranks then read the SAME gathered words and leave with the same code -- the first failing rank's, in rank order --
// before anything further is posted.  (A transport that fails itself -- an error out of RCCL -- is beyond this: that code is returned as it comes.)
// FMD_DIST_INJECT=<rank>:<where>[:<piece>] makes a rank fail on purpose (tests): where = head | keys | pack | arena.
static int dist_inject(const fmd_ovlp_dist *d, const char *where, int piece)
{
    const char *e = getenv("FMD_DIST_INJECT");
    if (!e) return FMD_OK;
    int r = -1, pc = -1;
    char w[16] = {0};
    if (sscanf(e, "%d:%15[a-z]:%d", &r, w, &pc) < 2 || r != d->rank || strcmp(w, where) != 0) return FMD_OK;
    if (pc >= 0 && piece >= 0 && pc != piece) return FMD_OK;
    fprintf(stderr, "[W::fmd_ovlp_dist_step] rank %d fails on purpose at `%s' (FMD_DIST_INJECT)\n", d->rank, where);
    return FMD_E_NOMEM;
}
// the verdict over `W` gathered status words (stride in u64 between ranks): the first failing rank's code, the same on every rank
static int dist_verdict(const fmd_ovlp_dist *d, const uint64_t *words, size_t stride, const char *what, int piece)
{
    for (int q = 0; q < d->world; ++q) {
        const int code = (int)(int64_t)words[(size_t)q * stride];
        if (code != FMD_OK) {
            if (piece >= 0) fprintf(stderr, "[E::fmd_ovlp_dist_step] rank %d: rank %d reports `%s' at %s, piece %d: every rank leaves the step with this code\n", d->rank, q, fmd_strerror(code), what, piece);
            else fprintf(stderr, "[E::fmd_ovlp_dist_step] rank %d: rank %d reports `%s' at %s: every rank leaves the step with this code\n", d->rank, q, fmd_strerror(code), what);
            return code;
        }
    }
    return FMD_OK;
}
// one status word per rank, all-gathered on `st` (8 bytes each): used where no other collective is at hand to carry it
static int dist_agree(fmd_ovlp_dist *d, hipStream_t st, int local, const char *what)
{
    if (d->world == 1) return local;
    unsigned long long *dev = (unsigned long long *)d->sizes_dev.p;     // (free between the pieces: [3] mine, then [W][3])
    uint64_t *host = (uint64_t *)d->sizes_host.p;
    const unsigned long long mine = (unsigned long long)(int64_t)local;
    if (hipMemcpyAsync(dev, &mine, 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); /* the all-gather still runs */ }
    int rc = d->comm->allgather(d->comm->ctx, st, dev, dev + 3, 8);
    if (rc != FMD_OK) return rc;
    FMD_HIP_TRY(hipMemcpyAsync(host, dev + 3, (size_t)d->world * 8, hipMemcpyDeviceToHost, st));
    FMD_HIP_TRY(hipStreamSynchronize(st));
    return dist_verdict(d, host, 1, what, -1);
}
#define DIST_LOCAL(expr) do { if (lerr == FMD_OK && (expr) != hipSuccess) { fmd_set_hip_error(hipGetLastError(), "overlap job on N GPUs"); lerr = FMD_E_HIP; } } while (0)

// the all-to-all of the parked strands; on return park_loc / ids_loc / keys / order describe the rows this rank computes.
// *fell_back = 1: some rank's share would not fit (decided the same way on every rank): the step runs on the id shard.
// Returns a code EVERY rank returns (a rank's own failure before the exchange travels in the status slot of the counts), or -- *local_err -- a failure
// of this rank alone after the last collective of the exchange, which the caller carries into the pieces' all-gathers.
static int key_exchange(fmd_ovlp_dist *d, hipStream_t sc, uint64_t *rows_out, std::vector<uint64_t> &rows_of_rank, int *fell_back, uint64_t *sent_rows, int *local_err)
{
    const int W = d->world, me = d->rank;
    const size_t n = d->n_home;
    const size_t CW = (size_t)W + 2;                                         // a rank's row of counts: [W] to each rank, [1] special, [1] status
    unsigned long long *cnt = (unsigned long long *)d->cnt_dev.p;           // [W + 2] mine, then [W][W + 2] everybody's
    int lerr = FMD_OK;
    // the key ranges: everybody's quantiles, the median of each
    uint32_t *q_dev = (uint32_t *)d->split_dev.p, *q_all = q_dev + W, *split_dev = q_all + (size_t)W * W;   // [W] mine, [W][W] everybody's, [W + 1] the boundaries
    uint32_t *q_host = (uint32_t *)d->split_host.p, *split = q_host + (size_t)W * W;
    k_ks_quantiles<<<(W + 63) / 64, 64, 0, sc>>>(n, (const uint32_t *)d->keys.p, W, q_dev);
    int rc = d->comm->allgather(d->comm->ctx, sc, q_dev, q_all, (size_t)W * 4);
    if (rc != FMD_OK) return rc;
    DIST_LOCAL(hipMemcpyAsync(q_host, q_all, (size_t)W * W * 4, hipMemcpyDeviceToHost, sc));
    DIST_LOCAL(hipStreamSynchronize(sc));
    split[0] = 0; split[W] = 0xfffffffeu;
    for (int p = 1; p < W; ++p) {
        std::vector<uint32_t> v((size_t)W);
        for (int r = 0; r < W; ++r) v[(size_t)r] = q_host[(size_t)r * W + p];
        std::sort(v.begin(), v.end());
        uint32_t m_ = v[(size_t)(W - 1) / 2];
        if (m_ < split[p - 1]) m_ = split[p - 1];
        if (m_ > 0xfffffffeu) m_ = 0xfffffffeu;
        split[p] = m_;
    }
    DIST_LOCAL(hipMemcpyAsync(split_dev, split, (size_t)(W + 1) * 4, hipMemcpyHostToDevice, sc));
    k_ks_counts<<<(W + 1 + 63) / 64, 64, 0, sc>>>(n, (const uint32_t *)d->keys.p, W, split_dev, cnt);
    if (lerr == FMD_OK) lerr = dist_inject(d, "keys", -1);
    { const unsigned long long stw = (unsigned long long)(int64_t)lerr; if (hipMemcpyAsync(cnt + W + 1, &stw, 8, hipMemcpyHostToDevice, sc) != hipSuccess || hipStreamSynchronize(sc) != hipSuccess) (void)hipGetLastError(); }
    rc = d->comm->allgather(d->comm->ctx, sc, cnt, cnt + CW, CW * 8);
    if (rc != FMD_OK) return rc;
    uint64_t *mat = (uint64_t *)d->cnt_host.p;
    FMD_HIP_TRY(hipMemcpyAsync(mat, cnt + CW, (size_t)W * CW * 8, hipMemcpyDeviceToHost, sc));
    FMD_HIP_TRY(hipStreamSynchronize(sc));
    rc = dist_verdict(d, mat + W + 1, CW, "the key exchange", -1);
    if (rc != FMD_OK) return rc;
    // rows every rank ends up with
    *fell_back = 0;
    for (int q = 0; q < W; ++q) {
        uint64_t m = mat[(size_t)q * CW + W];
        for (int s = 0; s < W; ++s) m += mat[(size_t)s * CW + q];
        rows_of_rank[(size_t)q] = m;
        const uint64_t nq = shard_size(d->cfg.n_ids, q, W), capq = nq + nq / 8 + 65536;
        if (m > capq || piece_begin(m, 1, d->pieces) + 1 > d->piece_max) *fell_back = 1;
    }
    if (*fell_back) return FMD_OK;
    // my segments in park_send: [to 0 | to 1 | ... | to W-1 | special]; what I receive, in rank order, then my own special rows.  Nothing that can fail
    // on one rank alone stands between the verdict above and the exchange: the local copies follow it (they touch other parts of the buffers).
    std::vector<fmd_comm_op_t> ops;
    uint64_t soff = 0, roff = 0;
    FmdWalkPark *send = (FmdWalkPark *)d->park_send.p, *loc = (FmdWalkPark *)d->park_loc.p;
    *sent_rows = 0;
    struct Own { uint64_t to, from, rows; };
    std::vector<Own> own;
    for (int q = 0; q < W; ++q) {
        const uint64_t sc_ = mat[(size_t)me * CW + q], rc_ = mat[(size_t)q * CW + me];
        if (q == me) { if (sc_) own.push_back(Own{roff, soff, sc_}); }
        else {
            if (sc_) { ops.push_back(fmd_comm_op_t{0, q, send + soff, (size_t)sc_ * sizeof(FmdWalkPark)}); *sent_rows += sc_; }
            if (rc_) ops.push_back(fmd_comm_op_t{1, q, loc + roff, (size_t)rc_ * sizeof(FmdWalkPark)});
        }
        soff += sc_; roff += rc_;
    }
    { const uint64_t sp = mat[(size_t)me * CW + W]; if (sp) own.push_back(Own{roff, soff, sp}); roff += sp; }
    rc = d->comm->exchange(d->comm->ctx, sc, (int)ops.size(), ops.data());
    if (rc != FMD_OK) return rc;
    for (const Own &o : own) DIST_LOCAL(hipMemcpyAsync(loc + o.to, send + o.from, o.rows * sizeof(FmdWalkPark), hipMemcpyDeviceToDevice, sc));
    const uint64_t m = roff;
    *rows_out = m;
    if (m && lerr == FMD_OK) {
        // the order of pass 2 over the rows as they arrived (W sorted runs): the same sort again, then the rows into that order (the send
        // buffer is free again: the exchange precedes this on the stream)
        uint8_t *w = (uint8_t *)d->work.p;
        const size_t tb = fmd_park_sort_temp_bytes(m);
        uint32_t *keys_a = (uint32_t *)w, *vals_a = (uint32_t *)(w + up256(m * 4));
        void *tmp = w + 2 * up256(m * 4);
        if (2 * up256(m * 4) + tb > d->work.bytes) lerr = FMD_E_NOMEM;
        else lerr = fmd_park_sort(sc, m, loc, keys_a, (uint32_t *)d->keys.p, vals_a, (uint32_t *)d->order.p, tmp, tb);
        if (lerr == FMD_OK) {
            k_park_permute<<<nblk(m, 64), 256, 0, sc>>>(m, (const uint32_t *)d->order.p, loc, send);
            k_ks_unpack<<<nblk(m, 256), 256, 0, sc>>>(m, send, (uint64_t *)d->ids_loc.p, (fmd_ovlp_rec_t *)d->rec.p);
        }
    }
    *local_err = lerr;
    return FMD_OK;
}

extern "C" int fmd_ovlp_dist_step(fmd_ovlp_dist_t *d, void *stream_, fmd_ovlp_dist_stats_t *stats)
{
    if (!d) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(d->h->device));
    hipStream_t sc = (hipStream_t)stream_, sm = d->sm, s3 = d->s3;
    const fmd_ovlp_dist_cfg_t &cfg = d->cfg;
    const int W = d->world, me = d->rank, root = cfg.root, P = d->pieces;
    const bool is_root = me == root;
    fmd_ovlp_dist_stats_t S;
    memset(&S, 0, sizeof(S));
    S.pieces = P; S.on_host = d->sink ? 2 : d->on_host; S.two_pass = d->two_pass;
    // root with a row sink: the piece that sits in landing set k goes to the sink, peer by peer, on this thread (the sink has its own threads) -- while the
    // GPUs compute and pack the next piece.  A sink that fails is a failure of this rank: it travels with the next status word like any other.
    for (int k = 0; k < 2; ++k) for (HStage &hs : d->hin[k]) hs.rows = 0;
    auto run_sink = [&](int k) -> int {
        if (!d->sink) return FMD_OK;
        bool any = false;
        for (const HStage &hs : d->hin[k]) any = any || hs.rows != 0;
        if (!any) return FMD_OK;
        if (hipEventSynchronize(d->drained[k]) != hipSuccess) { fmd_set_hip_error(hipGetLastError(), "overlap job on N GPUs: a piece on its way to the sink"); return FMD_E_HIP; }
        int src = FMD_OK;
        for (HStage &hs : d->hin[k]) {
            if (hs.rows && src == FMD_OK && d->cfg.row_sink(d->cfg.sink_ctx, hs.rows, (const uint32_t *)hs.pid.p, (const fmd_ovlp_rec_t *)hs.prec.p, (const uint64_t *)hs.off.p, (const uint8_t *)hs.var.p, d->cfg.max_nei) != 0) {
                fprintf(stderr, "[E::fmd_ovlp_dist_step] rank %d: the row sink turned %llu rows down\n", d->rank, (unsigned long long)hs.rows);
                src = FMD_E_IO;
            }
            hs.rows = 0;
        }
        return src;
    };
    const double t_begin = now_s();
    int rc = FMD_OK;
    int lerr = FMD_OK;      // a failure of this rank alone: carried, not returned (see dist_verdict above)
    FMD_HIP_TRY(hipEventRecord(d->ev[0], sc));
    FMD_HIP_TRY(hipStreamWaitEvent(sm, d->ev[0], 0));
    // ---- pass 1 on the id shard
    uint64_t rows = d->n_home;
    std::vector<uint64_t> rows_of_rank((size_t)W);
    for (int q = 0; q < W; ++q) rows_of_rank[(size_t)q] = shard_size(cfg.n_ids, q, W);
    FmdWalkPark *park = (FmdWalkPark *)d->park_home.p;
    const uint64_t *row_ids = nullptr;          // nullptr: row j of this rank is id me + W * j
    int key_shard = 0;
    if (d->two_pass) {
        if (d->n_home) lerr = fmd_ovlp_head_dev(d->h, sc, (size_t)d->n_home, (const uint64_t *)d->ids_home.p, cfg.min_match, cfg.max_len, (fmd_ovlp_rec_t *)d->rec.p, park,
                                                (uint32_t *)d->keys.p, (uint32_t *)d->order.p, d->work.p, d->work.bytes);
        if (lerr == FMD_OK) lerr = dist_inject(d, "head", -1);
        DIST_LOCAL(hipEventRecord(d->ev[1], sc));
        // the parked strands in key order, each with its id (and, if it ended inside the head, what its record needs): the send buffer of the
        // all-to-all, and -- without one -- already the rows of pass 2
        if (d->n_home && lerr == FMD_OK)
            k_ks_gather<<<nblk(d->n_home, 64), 256, 0, sc>>>((size_t)d->n_home, (const uint32_t *)d->order.p, (const FmdWalkPark *)d->park_home.p, (const uint64_t *)d->ids_home.p,
                                                             (const fmd_ovlp_rec_t *)d->rec.p, (FmdWalkPark *)d->park_send.p);
        int fell_back = 1;
        if (cfg.key_shard) {
            // the exchange is made of collectives: nobody enters it unless everybody's pass 1 went through
            rc = dist_agree(d, sc, lerr, "pass 1");
            if (rc != FMD_OK) return rc;
            uint64_t m = 0, sent = 0;
            rc = key_exchange(d, sc, &m, rows_of_rank, &fell_back, &sent, &lerr);
            if (rc != FMD_OK) return rc;
            if (!fell_back) { rows = m; key_shard = 1; S.key_rows_sent = sent; }
            else for (int q = 0; q < W; ++q) rows_of_rank[(size_t)q] = shard_size(cfg.n_ids, q, W);
        }
        if (fell_back && d->n_home && lerr == FMD_OK) k_ks_unpack<<<nblk(d->n_home, 256), 256, 0, sc>>>((size_t)d->n_home, (const FmdWalkPark *)d->park_send.p, (uint64_t *)d->ids_loc.p, (fmd_ovlp_rec_t *)d->rec.p);
        park = (FmdWalkPark *)d->park_send.p;
        row_ids = (const uint64_t *)d->ids_loc.p;
    } else {
        DIST_LOCAL(hipEventRecord(d->ev[1], sc));
    }
    DIST_LOCAL(hipEventRecord(d->ev[2], sc));
    S.key_shard = key_shard;
    d->n_rows = rows;
    d->loc_ids = row_ids ? row_ids : (const uint64_t *)d->ids_home.p;
    // ---- pass 2 + fm6_get_nei: every piece queued on the compute stream now; the loop below follows with pack + transfer
    for (int p = 0; p < P && lerr == FMD_OK; ++p) {
        const uint64_t b = piece_begin(rows, p, P), np = piece_begin(rows, p + 1, P) - b;
        if (np) {
            if (d->two_pass)
                lerr = fmd_ovlp_tail_dev(d->h, sc, (size_t)np, (const uint32_t *)d->iota.p + b, park, cfg.min_match, cfg.max_len, cfg.max_nei, (fmd_ovlp_rec_t *)d->rec.p,
                                         (fmd_intv_t *)d->nei.p, (uint8_t *)d->seq.p, d->stride, d->work.p, d->work.bytes);
            else
                lerr = fmd_ovlp_dev(d->h, sc, (size_t)np, (const uint64_t *)d->ids_home.p + b, cfg.min_match, cfg.max_len, cfg.max_nei, (fmd_ovlp_rec_t *)d->rec.p + b,
                                    (fmd_intv_t *)d->nei.p + b * cfg.max_nei, (uint8_t *)d->seq.p + b * (size_t)d->stride, d->stride, d->work.p, d->work.bytes);
        }
        DIST_LOCAL(hipEventRecord(d->done[(size_t)p], sc));
    }
    DIST_LOCAL(hipEventRecord(d->ev[3], sc));
    // ---- the pieces leave
    std::vector<uint64_t> rowbase((size_t)W + 1, 0);
    for (int q = 0; q < W; ++q) rowbase[(size_t)q + 1] = rowbase[(size_t)q] + rows_of_rank[(size_t)q];
    if (is_root) { d->var.reset(); DIST_LOCAL(hipMemsetAsync(d->t_row_of.p, 0xff, cfg.n_ids * 4, sm)); }
    unsigned long long *sz_dev = (unsigned long long *)d->sizes_dev.p;   // [3] mine = {rows, variable bytes, status}, then [W][3]
    uint64_t *sz_host = (uint64_t *)d->sizes_host.p;
    for (int p = 0; p < P && !(W == 1 && getenv("FMD_DIST_NO_PACK")); ++p) {   // (A/B knob: pass 2 alone)
        const uint64_t b = piece_begin(rows, p, P), np = piece_begin(rows, p + 1, P) - b;
        const int k = p & 1;
        // ---- this rank's own part of piece p: everything in it that can fail comes BEFORE the all-gather of the sizes, which carries the outcome
        uint32_t *o_pid = nullptr; fmd_ovlp_rec_t *o_prec = nullptr; uint64_t *o_off = nullptr; uint8_t *o_var = nullptr; uint64_t o_cap = 0;
        const uint64_t my_row0 = rowbase[(size_t)me] + b;
        if (lerr == FMD_OK) lerr = dist_inject(d, "pack", p);
        DIST_LOCAL(hipStreamWaitEvent(sm, d->done[(size_t)p], 0));
        if (p == P - 1) DIST_LOCAL(hipEventRecord(d->ev[4], sm));
        if (lerr == FMD_OK) {
            // where this rank's piece is packed to: straight into the table (root, table in HBM) or into the out buffers
            if (is_root && !d->on_host) {
                o_pid = (uint32_t *)d->t_ids.p + my_row0; o_prec = (fmd_ovlp_rec_t *)d->t_prec.p + my_row0; o_off = (uint64_t *)d->t_off.p + (size_t)me * (d->piece_max + 1);
                o_cap = fmd_ovlp_pack_max_bytes((size_t)np, cfg.max_nei, d->stride);
                o_var = (uint8_t *)d->var.take(o_cap ? o_cap : 256);
                if (!o_var) lerr = FMD_E_NOMEM;
            } else {
                Stage &os = d->out[is_root ? k : 0];
                if (is_root) DIST_LOCAL(hipStreamWaitEvent(sm, d->drained[k], 0));   // (host table: set k -- the root's own piece and the peers' staging -- has left for the host)
                o_pid = (uint32_t *)os.pid.p; o_prec = (fmd_ovlp_rec_t *)os.prec.p; o_off = (uint64_t *)os.off.p; o_var = (uint8_t *)os.var.p; o_cap = d->var_cap_piece;
            }
        }
        if (lerr == FMD_OK)
            lerr = fmd_ovlp_pack_rows_dev(d->h, sm, (size_t)np, (const uint32_t *)d->iota.p + b, row_ids, (uint64_t)me, (uint64_t)W, (const fmd_ovlp_rec_t *)d->rec.p, (const fmd_intv_t *)d->nei.p,
                                          cfg.max_nei, (const uint8_t *)d->seq.p, d->stride, o_pid, o_prec, o_off, o_var, o_cap, d->pack_work.p, d->pack_work.bytes);
        if (is_root && d->sink && p >= 1 && lerr == FMD_OK) lerr = run_sink(k ^ 1);   // piece p - 1, while piece p is being packed
        if (is_root && lerr == FMD_OK) {
            // the root's arena must hold whatever the peers send: room for their worst case is made NOW, while a failure can still travel with the sizes
            uint64_t np_max = 0;
            for (int q = 0; q < W; ++q) { const uint64_t nq = piece_begin(rows_of_rank[(size_t)q], p + 1, P) - piece_begin(rows_of_rank[(size_t)q], p, P); if (q != me || d->on_host) np_max = nq > np_max ? nq : np_max; }
            const size_t worst = fmd_ovlp_pack_max_bytes((size_t)np_max, cfg.max_nei, d->stride);
            if (dist_inject(d, "arena", p) != FMD_OK || (!d->sink && !d->var.ensure((size_t)(d->on_host ? W : W - 1), worst ? worst : 256))) lerr = FMD_E_NOMEM;   // (a sink: the landing sets hold a worst case each)
        }
        // sizes of everybody's piece p, and how everybody fared
        if (lerr == FMD_OK) DIST_LOCAL(hipMemcpyAsync(sz_dev + 1, o_off + np, 8, hipMemcpyDeviceToDevice, sm));
        {
            const unsigned long long head[3] = {lerr == FMD_OK ? (unsigned long long)np : 0ull, 0ull, (unsigned long long)(int64_t)lerr};
            bool ok = hipMemcpyAsync(sz_dev, &head[0], 8, hipMemcpyHostToDevice, sm) == hipSuccess && hipMemcpyAsync(sz_dev + 2, &head[2], 8, hipMemcpyHostToDevice, sm) == hipSuccess;
            if (lerr != FMD_OK) ok = ok && hipMemcpyAsync(sz_dev + 1, &head[1], 8, hipMemcpyHostToDevice, sm) == hipSuccess;
            ok = ok && hipStreamSynchronize(sm) == hipSuccess;
            if (!ok) (void)hipGetLastError();      // (a device that cannot even take this is beyond the protocol: the all-gather below reports or hangs with it)
        }
        if (W > 1) {
            rc = d->comm->allgather(d->comm->ctx, sm, sz_dev, sz_dev + 3, 24);
            if (rc != FMD_OK) return rc;
            FMD_HIP_TRY(hipMemcpyAsync(sz_host, sz_dev + 3, (size_t)W * 24, hipMemcpyDeviceToHost, sm));
        } else FMD_HIP_TRY(hipMemcpyAsync(sz_host, sz_dev, 24, hipMemcpyDeviceToHost, sm));
        FMD_HIP_TRY(hipStreamSynchronize(sm));
        // ---- the verdict, from the same words on every rank; then -- also on every rank, from the same words -- the row counts
        rc = dist_verdict(d, sz_host + 2, 3, "pass 2 / the pack", p);
        if (rc != FMD_OK) return rc;
        for (int q = 0; q < W; ++q) {
            const uint64_t want = piece_begin(rows_of_rank[(size_t)q], p + 1, P) - piece_begin(rows_of_rank[(size_t)q], p, P);
            if (sz_host[3 * q] != want) { fprintf(stderr, "[E::fmd_ovlp_dist_step] rank %d sends %llu rows in piece %d, %llu expected\n", q, (unsigned long long)sz_host[3 * q], p, (unsigned long long)want); return FMD_E_ARG; }
        }
        S.rows_computed += np;
        if (is_root && !d->on_host) d->var.shrink_last(o_cap ? o_cap : 256, sz_host[3 * me + 1] ? sz_host[3 * me + 1] : 256);   // (the worst case was reserved for the root's own piece)
        if (!is_root) {
            const uint64_t vb = sz_host[3 * me + 1];
            fmd_comm_op_t ops[4] = {{0, root, o_pid, (size_t)np * 4}, {0, root, o_prec, (size_t)np * sizeof(fmd_ovlp_rec_t)}, {0, root, o_off, (size_t)(np + 1) * 8}, {0, root, o_var, (size_t)vb}};
            if (np) { rc = d->comm->exchange(d->comm->ctx, sm, 4, ops); if (rc != FMD_OK) return rc; }
            S.rows_sent += np; S.bytes_sent += np * (4 + sizeof(fmd_ovlp_rec_t)) + (np + 1) * 8 + vb;
            continue;
        }
        // ---- root: receive every peer's piece p, place it (the arena has the room: ensure() above)
        std::vector<fmd_comm_op_t> ops;
        struct Placed { const uint32_t *pid; const uint64_t *off; uint64_t var_base, row0, np; uint64_t *vaddr_dst; Stage *st; uint64_t vb; int q; };
        std::vector<Placed> placed;
        for (int q = 0; q < W; ++q) {
            const uint64_t nq = sz_host[3 * q], vb = sz_host[3 * q + 1];
            const uint64_t bq = piece_begin(rows_of_rank[(size_t)q], p, P);
            if (!nq) continue;
            const uint64_t row0 = rowbase[(size_t)q] + bq;
            if (!d->on_host) {
                uint64_t *offq = (uint64_t *)d->t_off.p + (size_t)q * (d->piece_max + 1);
                uint8_t *vq = o_var;
                if (q != me) {
                    vq = (uint8_t *)d->var.take(vb ? vb : 256);
                    if (!vq) return FMD_E_NOMEM;        // (cannot happen after ensure(): a peer's variable part is no larger than its worst case)
                    ops.push_back(fmd_comm_op_t{1, q, (uint32_t *)d->t_ids.p + row0, (size_t)nq * 4});
                    ops.push_back(fmd_comm_op_t{1, q, (fmd_ovlp_rec_t *)d->t_prec.p + row0, (size_t)nq * sizeof(fmd_ovlp_rec_t)});
                    ops.push_back(fmd_comm_op_t{1, q, offq, (size_t)(nq + 1) * 8});
                    ops.push_back(fmd_comm_op_t{1, q, vq, (size_t)vb});
                    S.bytes_received += nq * (4 + sizeof(fmd_ovlp_rec_t)) + (nq + 1) * 8 + vb;
                }
                placed.push_back(Placed{(const uint32_t *)d->t_ids.p + row0, offq, (uint64_t)(uintptr_t)vq, row0, nq, (uint64_t *)d->t_vaddr.p + row0, nullptr, vb, q});
            } else {
                Stage *st = &d->in[k][(size_t)q];
                uint8_t *hv = d->sink ? (uint8_t *)d->hin[k][(size_t)q].var.p : (uint8_t *)d->var.take(vb ? vb : 256);   // the row's final place in pinned host memory (a sink: where it waits for the sink)
                if (!hv) return FMD_E_NOMEM;            // (as above)
                const uint32_t *pid_src; const uint64_t *off_src;
                if (q != me) {
                    ops.push_back(fmd_comm_op_t{1, q, st->pid.p, (size_t)nq * 4});
                    ops.push_back(fmd_comm_op_t{1, q, st->prec.p, (size_t)nq * sizeof(fmd_ovlp_rec_t)});
                    ops.push_back(fmd_comm_op_t{1, q, st->off.p, (size_t)(nq + 1) * 8});
                    ops.push_back(fmd_comm_op_t{1, q, st->var.p, (size_t)vb});
                    S.bytes_received += nq * (4 + sizeof(fmd_ovlp_rec_t)) + (nq + 1) * 8 + vb;
                    pid_src = (const uint32_t *)st->pid.p; off_src = (const uint64_t *)st->off.p;
                } else { pid_src = o_pid; off_src = o_off; }
                placed.push_back(Placed{pid_src, off_src, (uint64_t)(uintptr_t)hv, row0, nq, (uint64_t *)st->vaddr.p, st, vb, q});
            }
        }
        if (!ops.empty()) { rc = d->comm->exchange(d->comm->ctx, sm, (int)ops.size(), ops.data()); if (rc != FMD_OK) return rc; }
        for (const Placed &pl : placed)
            k_place<<<nblk(pl.np, 256), 256, 0, sm>>>((size_t)pl.np, pl.pid, pl.off, pl.var_base, pl.row0, pl.vaddr_dst, (uint32_t *)d->t_row_of.p, cfg.n_ids);
        if (d->on_host) {   // staging set k -> the pinned table, on the copy stream
            FMD_HIP_TRY(hipEventRecord(d->staged[k], sm));
            FMD_HIP_TRY(hipStreamWaitEvent(s3, d->staged[k], 0));
            for (const Placed &pl : placed) {
                const bool mine = pl.pid == o_pid;
                const void *src_prec = mine ? (const void *)o_prec : pl.st->prec.p, *src_var = mine ? (const void *)o_var : pl.st->var.p;
                if (d->sink) {   // ids, records and the offsets as they were packed (the sink reads a piece as fmd_ovlp_pack_rows_dev wrote it: off[] into var)
                    HStage &hs = d->hin[k][(size_t)pl.q];
                    FMD_HIP_TRY(hipMemcpyAsync(hs.pid.p, pl.pid, pl.np * 4, hipMemcpyDeviceToHost, s3));
                    FMD_HIP_TRY(hipMemcpyAsync(hs.prec.p, src_prec, pl.np * sizeof(fmd_ovlp_rec_t), hipMemcpyDeviceToHost, s3));
                    FMD_HIP_TRY(hipMemcpyAsync(hs.off.p, pl.off, (pl.np + 1) * 8, hipMemcpyDeviceToHost, s3));
                    hs.rows = pl.np;
                } else {
                FMD_HIP_TRY(hipMemcpyAsync((uint32_t *)d->th_ids.p + pl.row0, pl.pid, pl.np * 4, hipMemcpyDeviceToHost, s3));
                FMD_HIP_TRY(hipMemcpyAsync((fmd_ovlp_rec_t *)d->th_prec.p + pl.row0, src_prec, pl.np * sizeof(fmd_ovlp_rec_t), hipMemcpyDeviceToHost, s3));
                FMD_HIP_TRY(hipMemcpyAsync((uint64_t *)d->th_vaddr.p + pl.row0, pl.vaddr_dst, pl.np * 8, hipMemcpyDeviceToHost, s3));
                }
                if (pl.vb) FMD_HIP_TRY(hipMemcpyAsync((void *)(uintptr_t)pl.var_base, src_var, pl.vb, hipMemcpyDeviceToHost, s3));
            }
            FMD_HIP_TRY(hipEventRecord(d->drained[k], s3));
        }
    }
    if (lerr != FMD_OK) return lerr;       // (W == 1 with FMD_DIST_NO_PACK: nobody to tell)
    if (W == 1 && getenv("FMD_DIST_NO_PACK")) { FMD_HIP_TRY(hipStreamWaitEvent(sm, d->done[(size_t)P - 1], 0)); FMD_HIP_TRY(hipEventRecord(d->ev[4], sm)); }
    FMD_HIP_TRY(hipEventRecord(d->ev[5], sm));
    // ---- the end of this rank's part in the gather (the end of its compute is ev[3] on the compute stream)
    FMD_HIP_TRY(hipStreamSynchronize(sm));
    if (is_root && d->on_host && !d->sink) FMD_HIP_TRY(hipMemcpyAsync(d->th_row_of.p, d->t_row_of.p, cfg.n_ids * 4, hipMemcpyDeviceToHost, s3));
    FMD_HIP_TRY(hipEventRecord(d->ev[6], s3));
    FMD_HIP_TRY(hipStreamSynchronize(s3));
    if (cfg.host_table == 2) {   // what is left in the landing sets (the last piece), and how the sink fared with it: one more status word, so that this too is everybody's code
        int serr = FMD_OK;
        if (is_root) { serr = run_sink(P & 1); if (serr == FMD_OK) serr = run_sink((P - 1) & 1); }
        rc = dist_agree(d, sm, serr, "the row sink");
        if (rc != FMD_OK) return rc;
    }
    const double t_end = now_s();
    FMD_HIP_TRY(hipStreamWaitEvent(sc, d->ev[5], 0));   // the caller's stream owns the table
    float ms = 0;
    if (hipEventElapsedTime(&ms, d->ev[0], d->ev[1]) == hipSuccess) S.head_ms = ms;
    if (hipEventElapsedTime(&ms, d->ev[1], d->ev[2]) == hipSuccess) S.key_exchange_ms = ms;
    if (hipEventElapsedTime(&ms, d->ev[2], d->ev[3]) == hipSuccess) S.tail_ms = ms;
    if (hipEventElapsedTime(&ms, d->ev[4], d->ev[5]) == hipSuccess) S.last_piece_pack_send_ms = ms;
    (void)hipGetLastError();
    // what of the gather did not hide under this rank's compute: from the end of its last kernel to the end of its last transfer
    // (root with a host table: to the end of the last copy to the host)
    if (hipEventElapsedTime(&ms, d->ev[3], d->ev[is_root && d->on_host ? 6 : 5]) == hipSuccess) S.gather_exposed_ms = ms > 0 ? ms : 0;
    S.step_ms = (t_end - t_begin) * 1e3;
    d->last = S;
    if (stats) *stats = S;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "overlap job on N GPUs"); return FMD_E_HIP; }
    return FMD_OK;
}

extern "C" int fmd_ovlp_dist_table(fmd_ovlp_dist_t *d, fmd_ovlp_dist_table_t *t)
{
    if (!d || !t) return FMD_E_ARG;
    memset(t, 0, sizeof(*t));
    if (d->rank != d->cfg.root) return FMD_E_ARG;
    if (d->sink) return FMD_E_ARG;       // (no table was kept: the rows went to cfg.row_sink)
    t->on_host = d->on_host;
    t->n_rows = d->cfg.n_ids;
    if (d->on_host) { t->prec = (const fmd_ovlp_rec_t *)d->th_prec.p; t->ids = (const uint32_t *)d->th_ids.p; t->vaddr = (const uint64_t *)d->th_vaddr.p; t->row_of_id = (const uint32_t *)d->th_row_of.p; }
    else { t->prec = (const fmd_ovlp_rec_t *)d->t_prec.p; t->ids = (const uint32_t *)d->t_ids.p; t->vaddr = (const uint64_t *)d->t_vaddr.p; t->row_of_id = (const uint32_t *)d->t_row_of.p; }
    return FMD_OK;
}

extern "C" int fmd_ovlp_dist_local(fmd_ovlp_dist_t *d, uint64_t *n_rows, const uint64_t **d_ids, const fmd_ovlp_rec_t **d_rec, const fmd_intv_t **d_nei, const uint8_t **d_seq, uint32_t *seq_stride)
{
    if (!d) return FMD_E_ARG;
    if (n_rows) *n_rows = d->n_rows;
    if (d_ids) *d_ids = d->loc_ids;
    if (d_rec) *d_rec = (const fmd_ovlp_rec_t *)d->rec.p;
    if (d_nei) *d_nei = (const fmd_intv_t *)d->nei.p;
    if (d_seq) *d_seq = (const uint8_t *)d->seq.p;
    if (seq_stride) *seq_stride = d->stride;
    return FMD_OK;
}
