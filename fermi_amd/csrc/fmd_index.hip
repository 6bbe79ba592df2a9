// fmd_index.hip -- index residency: upload a fermi .fmd (RLD\2 or RLE\6) or a plain BWT and
// transcode it ON THE GPU into the fixed-stride rank-block layout of fmd_wave.h.
// Replaces rld_restore / rld_restore_mmap / rld_destroy (rld.c:288, :327, :81) for the device.
#include "fmd_prim.h"
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include <unistd.h>
#include <sys/types.h>
#include "fmd_internal.h"

// ---------------------------------------------------------------------------------- errors
static thread_local char g_hip_err[256] = "";
void fmd_set_hip_error(hipError_t e, const char *what)
{
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();   // the code travels in the return value and the text here: the runtime's own "last error" is not left set for the next HIP call of the process
                               // (another library's launch check) to trip over -- a failed hipMalloc in fmd_dev_open_file ended a torch kernel launch minutes later
}
extern "C" const char *fmd_last_hip_error(void) { return g_hip_err; }
extern "C" const char *fmd_strerror(int code)
{
    switch (code) {
    case FMD_OK: return "ok";
    case FMD_E_NODEV: return "no usable HIP device (libfmdhip has no CPU fallback)";
    case FMD_E_ARG: return "bad argument";
    case FMD_E_FORMAT: return "not a fermi .fmd (RLD\\2 with asize 6 / sbits 3, or RLE\\6)";
    case FMD_E_IO: return "file I/O failed";
    case FMD_E_NOMEM: return "out of host or device memory";
    case FMD_E_HIP: return "HIP runtime error (see fmd_last_hip_error)";
    case FMD_E_OVERFLOW: return "fixed-capacity device list overflowed";
    default: return "unknown error";
    }
}
extern "C" int fmd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------- transcode: plain BWT
// one thread per 32-position chunk
__global__ void k_bwt_to_planes(const uint8_t *__restrict__ bwt, uint64_t n, uint4 *__restrict__ blocks, uint64_t n_chunks)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = c * 32;
        uint32_t a = 0, b = 0, d = 0;
        if (p0 + 32 <= n) {
            const uint4 *src = (const uint4 *)(bwt + p0); // hipMalloc'd + 32-byte stride: aligned
            const uint4 v0 = src[0], v1 = src[1];
            const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t s = (w[i] >> (8 * j)) & 7;
                    a |= (s & 1) << (4 * i + j); b |= ((s >> 1) & 1) << (4 * i + j); d |= ((s >> 2) & 1) << (4 * i + j);
                }
        } else {
            for (int i = 0; i < 32 && p0 + i < n; ++i) {
                const uint32_t s = bwt[p0 + i] & 7;
                a |= (s & 1) << i; b |= ((s >> 1) & 1) << i; d |= ((s >> 2) & 1) << i;
            }
        }
        uint4 *dst = blocks + fmd_word_u4(c); // 32-position word c -> its block and chunk
        dst->x = a; dst->y = b; dst->z = d;
    }
}

// --------------------------------------------------------- transcode: scatter of (sym,len) runs
// OR `len` copies of symbol `sym` into the planes starting at BWT position `pos`.
__device__ __forceinline__ void fmd_or_run(uint32_t *words, uint64_t pos, uint64_t len, uint32_t sym)
{
    if (sym == 0) return; // '$' = all-zero planes
    uint64_t p = pos, end = pos + len;
    while (p < end) {
        const uint32_t bit = (uint32_t)p & 31;
        const uint64_t take = (end - p < 32 - bit) ? end - p : 32 - bit;
        const uint32_t m = (take == 32 ? 0xffffffffu : ((1u << take) - 1u)) << bit;
        uint32_t *w = words + fmd_word_u4(p >> 5) * 4; // chunk = 4 words
        if (take == 32) {                   // whole word is ours
            if (sym & 1) w[0] = m;
            if (sym & 2) w[1] = m;
            if (sym & 4) w[2] = m;
        } else {
            if (sym & 1) atomicOr(w + 0, m);
            if (sym & 2) atomicOr(w + 1, m);
            if (sym & 4) atomicOr(w + 2, m);
        }
        p += take;
    }
}

// RLE\6 stream: byte = len<<3 | sym, len 1..31 (ropebwt.c:132-136; reader rld.c:295-308)
__global__ void k_rle6_len(const uint8_t *__restrict__ runs, uint64_t n, uint64_t *__restrict__ len)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) len[i] = runs[i] >> 3;
}
__global__ void k_rle6_scatter(const uint8_t *__restrict__ runs, uint64_t n, const uint64_t *__restrict__ start,
                               uint32_t *__restrict__ words, uint64_t *__restrict__ sym_total)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = runs[i] >> 3, c = runs[i] & 7;
        if (l) fmd_or_run(words, start[i], l, c);
        if (i == n - 1) *sym_total = start[i] + l;
    }
}

// RLD\2 payload (rld.c:111-175 writer; rld.h:77-94 decoder): 64-byte blocks, header = counts of
// the PREVIOUS block as 7 x u16 or (bit 31 of the first u32 set) 7 x u32, then MSB-first
// Elias-delta run codes.
__device__ __forceinline__ uint64_t rld_hdr_size(const uint64_t *blk)
{
    const uint32_t w0 = (uint32_t)blk[0];
    return (w0 >> 31) ? (w0 & 0x7fffffffu) : (w0 & 0xffffu);
}
__global__ void k_rld_sizes(const uint64_t *__restrict__ w, uint64_t n_rld_blocks, uint64_t *__restrict__ size)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_rld_blocks; b += (uint64_t)gridDim.x * blockDim.x) size[b] = rld_hdr_size(w + (b + 1) * 8); // next header describes block b
}
__device__ __forceinline__ uint64_t rld_peek(const uint64_t *w, uint32_t bit /*0..511*/)
{
    const uint32_t i = bit >> 6, off = bit & 63;
    if (i >= 8) return 0;
    uint64_t x = w[i] << off;
    if (off && i + 1 < 8) x |= w[i + 1] >> (64 - off);
    return x;
}
// A run that would end past the symbol count of the header is not written: the file is corrupt or truncated, and
// sym_total[1] tells the host (FMD_E_FORMAT) -- the decoded lengths are never trusted to stay inside the index.
__global__ void k_rld_scatter(const uint64_t *__restrict__ w, uint64_t n_rld_blocks, const uint64_t *__restrict__ start,
                              uint32_t *__restrict__ words, uint64_t n_sym, uint64_t *__restrict__ sym_total)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_rld_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t *blk = w + b * 8;
        uint32_t bit = (((uint32_t)blk[0] >> 31) ? 4u : 2u) * 64u;
        uint64_t pos = start[b];
        for (;;) {
            const uint64_t x = rld_peek(blk, bit);
            uint64_t len; uint32_t sym;
            if (x >> 63) { len = 1; sym = (uint32_t)(x >> 60) & 7; bit += 4; }
            else {
                const int z = x ? __clzll((long long)x) : 64;
                if (z >= 6) break;                                   // zero padding: block exhausted
                const int gw = 2 * z + 1, nlow = (int)(x >> (64 - gw)) - 1;
                len = ((x << gw) >> (64 - nlow)) | (1ull << nlow);
                sym = (uint32_t)((x << (gw + nlow)) >> 61);
                bit += (uint32_t)(gw + nlow + 3);
            }
            if (pos > n_sym || len > n_sym - pos) { sym_total[1] = 1; break; }
            fmd_or_run(words, pos, len, sym);
            pos += len;
        }
        if (b == n_rld_blocks - 1) sym_total[0] = pos;
    }
}

// -------------------------------------------------------------- per-block symbol counts + meta
// Counts fit a byte for 96-position blocks and a 16-bit word for 256-position ones; the six running
// sums are then taken one symbol at a time through ONE 8-byte-per-block buffer, so finishing an index
// costs 20 (14) bytes per block on top of the index itself -- the 1.4e11-symbol index (1.5e9 blocks)
// must fit next to its own 94 GB.
typedef uint8_t fmd_bc_t;
__global__ void k_block_counts(const uint4 *__restrict__ blocks, uint64_t n_blocks, fmd_bc_t *__restrict__ bc)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t n[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < FMD_BLK_OWN_CHUNKS; ++c) {   // (overlapped blocks: the look-ahead chunk belongs to the next block's count)
            const uint4 v = blocks[b * FMD_BLK_U4 + c];
            n[0] += __builtin_popcount(~v.z & ~v.y & ~v.x); n[1] += __builtin_popcount(~v.z & ~v.y & v.x);
            n[2] += __builtin_popcount(~v.z & v.y & ~v.x);  n[3] += __builtin_popcount(~v.z & v.y & v.x);
            n[4] += __builtin_popcount(v.z & ~v.y & ~v.x);  n[5] += __builtin_popcount(v.z & ~v.y & v.x);
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) bc[(uint64_t)s * n_blocks + b] = (fmd_bc_t)n[s];
    }
}
// overlapped blocks: the planes of the first chunk of block b + 1 repeated as the third chunk of block b (the transcoders write
// every 32-position word once, into the block that owns it)
__global__ void k_fill_lookahead(uint4 *__restrict__ blocks, uint64_t n_blocks)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (b + 1 < n_blocks) v = blocks[(b + 1) * FMD_BLK_U4];
        uint4 *dst = blocks + b * FMD_BLK_U4 + FMD_BLK_OWN_CHUNKS;
        dst->x = v.x; dst->y = v.y; dst->z = v.z;
    }
}
// the count of symbol s before each block into its place in the block's meta words (fmd_wave.h)
__global__ void k_write_meta_sym(uint4 *__restrict__ blocks, uint64_t n_blocks, const uint64_t *__restrict__ acc, int s)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = acc[b];
        const uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32) & 0xff;
        // meta_0..2 in the .w of the three plane chunks, meta_3..6 = the fourth uint4; N is not stored
        uint4 *m = blocks + b * 4;
        if (s < 3) m[s].w = lo; else if (s == 3) m[3].x = lo; else if (s == 4) m[3].y = lo;
        if (s < 4) m[3].z = (m[3].z & ~(0xffu << (8 * s))) | hi << (8 * s);
        else if (s == 4) m[3].w = hi;
    }
}

// ------------------------------------------------------------------ prefix table (FmdIndexView::ptab)
// level d from level d-1: the string c S (c prepended) is one backward extension of S by c.
// thread = index of the new string; its top two bits are c - 1.
__global__ void k_ptab_level(FmdIndexView ix, int d, const uint4 *__restrict__ prev, uint4 *__restrict__ next)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ull << (2 * d))) return;
    const int c = (int)(i >> (2 * (d - 1))) + 1;
    uint64_t k, l;
    if (d == 1) { k = ix.cnt[c]; l = ix.cnt[c + 1] - 1; }
    else {
        const uint4 e = prev[i & ((1ull << (2 * (d - 1))) - 1)];
        k = (uint64_t)e.y << 32 | e.x; l = (uint64_t)e.w << 32 | e.z;
        if (k <= l) { // rank_c(k - 1), rank_c(l)  (k >= 1 in a real index: position 0 holds a sentinel suffix)
            uint32_t b, o;
            uint64_t rk = 0;
            if (k) { fmd_split(k - 1, b, o); rk = fmd_block_rank1(ix.blocks + (size_t)b * FMD_BLK_U4, 0, o + 1, c, b); }
            fmd_split(l, b, o);
            const uint64_t rl = fmd_block_rank1(ix.blocks + (size_t)b * FMD_BLK_U4, 0, o + 1, c, b);
            k = ix.cnt[c] + rk; l = ix.cnt[c] + rl - 1;
        }
    }
    if (k > l) { k = 1; l = 0; }
    next[i] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)l, (uint32_t)(l >> 32));
}

static int build_ptab(fmd_dev *h)
{
    // as deep as keeps the table well below the index: 4^d <= n/8, at most 14 (4.3 GB; 12 = 268 MB until round 6), at least 2 -- and no deeper than the
    // tail table can say: its 8-byte entry holds 2 d bits of bases beside the row (FmdIndexView::tail), so the index must have fewer than 2^(64 - 2 d) symbols.
    // Every base the tables take is a base the head does not walk, and the first ones are the expensive ones -- the interval is still wider than a block,
    // up to three lines and two wave steps a base: pass 1 of the sorted job 36 -> 27 ms per 10^8 strands from depth 12 to 14 (profiles/r6_ptab).
    int d = 2;
    while (d < 14 && (1ull << (2 * (d + 1))) <= h->mcnt[0] / 8 && h->mcnt[0] < (1ull << (64 - 2 * (d + 1))) - 1) ++d;
    if (getenv("FMD_PTAB_DEPTH")) { d = atoi(getenv("FMD_PTAB_DEPTH")); if (d < 1) return FMD_OK; if (d > 15) d = 15; while (d > 2 && ((1ull << (2 * d)) > h->mcnt[0] || h->mcnt[0] >= (1ull << (64 - 2 * d)) - 1)) --d; }   // (15: 17 GB, by request only)
    uint4 *a = nullptr, *b = nullptr;
    const uint64_t n = 1ull << (2 * d);
    if (hipMalloc((void **)&a, n * 16) != hipSuccess) { (void)hipGetLastError(); return FMD_E_NOMEM; }      // (the error is the caller's to report: none is left behind for the next HIP call of the process to trip over)
    if (hipMalloc((void **)&b, (n / 4 ? n / 4 : 1) * 16) != hipSuccess) { (void)hipGetLastError(); hipFree(a); return FMD_E_NOMEM; }
    // levels alternate between b (odd distance from the last) and a, so that level d lands in a
    FmdIndexView ix = fmd_view(h);
    uint4 *cur = nullptr;
    for (int lv = 1; lv <= d; ++lv) {
        uint4 *dst = ((d - lv) & 1) ? b : a;
        const uint64_t m = 1ull << (2 * lv);
        k_ptab_level<<<(unsigned)((m + 255) / 256), 256>>>(ix, lv, cur, dst);
        cur = dst;
    }
    hipError_t e = hipDeviceSynchronize();
    hipFree(b);
    if (e != hipSuccess) { hipFree(a); fmd_set_hip_error(e, "prefix table"); return FMD_E_HIP; }
    h->ptab = a; h->ptab_d = d;
    h->bytes += n * 16;
    return FMD_OK;
}

// ------------------------------------------------------------------ tail table (FmdIndexView::tail)
#define FMD_TAIL_NONE (~0ull)
static unsigned nblk(uint64_t n, unsigned per);
__global__ void k_tail_table(FmdIndexView ix, int d, unsigned long long *__restrict__ tail)
{
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < ix.n_seq; id += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = id, tfw = 0;
        bool ok = true;
        for (int j = 0; j < d && ok; ++j) {   // the LF step of fm_retrieve (exact.c:63-66): base = BWT[k], k = cnt[c] + rank_c(k) - 1
            uint32_t b, o;
            uint64_t r[6];
            fmd_split(k, b, o);
            const int c = fmd_block_rank6<true>(ix.blocks + (size_t)b * FMD_BLK_U4, 0, o + 1, r, b);
            if (c < 1 || c > 4) { ok = false; break; }
            k = ix.cnt[c] + r[c] - 1;
            tfw |= (uint64_t)(c - 1) << (2 * j);
        }
        tail[id] = ok ? (k | tfw << (64 - 2 * d)) : FMD_TAIL_NONE;      // (FmdIndexView::tail: the row in the low 64 - 2 d bits)
    }
}
static int build_tail(fmd_dev *h)
{
    const char *e = getenv("FMD_TAIL_TABLE");
    if ((e && atoi(e) == 0) || !h->ptab || h->ptab_d < 2 || h->ptab_d > 15 || h->mcnt[0] >= (1ull << (64 - 2 * h->ptab_d)) - 1 || h->mcnt[1] == 0) return FMD_OK;   // 2 d bits of bases beside a row of 64 - 2 d bits
    unsigned long long *t = nullptr;
    if (hipMalloc((void **)&t, h->mcnt[1] * 8) != hipSuccess) { (void)hipGetLastError(); return FMD_OK; }   // no room: the walk takes its steps itself
    k_tail_table<<<nblk(h->mcnt[1], 256), 256>>>(fmd_view(h), h->ptab_d, t);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { hipFree(t); fmd_set_hip_error(err, "tail table"); return FMD_E_HIP; }
    h->tail = t;
    h->bytes += h->mcnt[1] * 8;
    return FMD_OK;
}

// --------------------------------------------------------------------------------- host side
struct FmdWiden { __host__ __device__ uint64_t operator()(fmd_bc_t v) const { return (uint64_t)v; } };
static int scan_counts(const fmd_bc_t *d_in, uint64_t *d_out, uint64_t n, hipStream_t st)
{
    rocprim::transform_iterator<const fmd_bc_t *, FmdWiden, uint64_t> in(d_in, FmdWiden());
    void *tmp = nullptr; size_t tmp_bytes = 0;
    FMD_HIP_TRY(fmd_exclusive_sum(nullptr, tmp_bytes, in, d_out, (size_t)n, st));
    FMD_HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    hipError_t e = fmd_exclusive_sum(tmp, tmp_bytes, in, d_out, (size_t)n, st);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(tmp);
    FMD_HIP_TRY(e); FMD_HIP_TRY(e2);
    return FMD_OK;
}

static int scan_u64(uint64_t *d_in, uint64_t *d_out, uint64_t n, hipStream_t st)
{
    void *tmp = nullptr; size_t tmp_bytes = 0;
    FMD_HIP_TRY(fmd_exclusive_sum(nullptr, tmp_bytes, d_in, d_out, (size_t)n, st));
    FMD_HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    hipError_t e = fmd_exclusive_sum(tmp, tmp_bytes, d_in, d_out, (size_t)n, st);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(tmp);
    FMD_HIP_TRY(e); FMD_HIP_TRY(e2);
    return FMD_OK;
}

// blocks for n items, t threads each: at most 2^31 threads per launch (the dispatch packet counts work-items in 32
// bits; the kernels above loop with a grid stride)
static inline unsigned nblk(uint64_t n, unsigned t)
{
    const uint64_t b = (n + t - 1) / t, cap = (1ull << 31) / t;
    return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

static int dev_alloc_index(int device, uint64_t n_sym, fmd_dev **out)
{
    int ndev = fmd_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return FMD_E_NODEV;
    if (n_sym == 0 || n_sym >= (1ull << 40)) return FMD_E_ARG; // 40-bit absolute counts
    if ((n_sym + FMD_BLK_STRIDE - 1) / FMD_BLK_STRIDE + 1 >= 0xffffffffull) return FMD_E_ARG; // 32-bit block numbers
    FMD_HIP_TRY(hipSetDevice(device));
    fmd_dev *h = (fmd_dev *)calloc(1, sizeof(fmd_dev));
    if (!h) return FMD_E_NOMEM;
    hipDeviceProp_t prop;
    FMD_HIP_TRY(hipGetDeviceProperties(&prop, device));
    h->device = device;
    h->n_cu = prop.multiProcessorCount;
    h->n_blocks = (n_sym + FMD_BLK_STRIDE - 1) / FMD_BLK_STRIDE + 1; // +1 pad block
    h->bytes = h->n_blocks * FMD_BLK_BYTES;
    hipError_t e = hipMalloc((void **)&h->blocks, h->bytes);
    if (e != hipSuccess) { fmd_set_hip_error(e, "hipMalloc(index)"); free(h); return FMD_E_NOMEM; }
    e = hipMalloc((void **)&h->queues, FMD_N_QUEUES * sizeof(uint32_t));
    if (e != hipSuccess) { fmd_set_hip_error(e, "hipMalloc(queues)"); hipFree(h->blocks); free(h); return FMD_E_NOMEM; }
    e = hipMalloc((void **)&h->stat, FMD_STAT_SLOTS * FMD_STAT_STRIDE * 8);
    if (e != hipSuccess) { fmd_set_hip_error(e, "hipMalloc(stat)"); hipFree(h->queues); hipFree(h->blocks); free(h); return FMD_E_NOMEM; }
    hipMemset(h->blocks, 0, h->bytes);
    hipMemset(h->queues, 0, FMD_N_QUEUES * sizeof(uint32_t));
    hipMemset(h->stat, 0, FMD_STAT_SLOTS * FMD_STAT_STRIDE * 8);
    *out = h;
    return FMD_OK;
}

// counts -> scan -> meta; also fills cnt/mcnt from the device counts
static int finish_index(fmd_dev *h)
{
    const uint64_t nb = h->n_blocks;
    fmd_bc_t *bc = nullptr; uint64_t *acc = nullptr;
    FMD_HIP_TRY(hipMalloc((void **)&bc, 6 * nb * sizeof(fmd_bc_t)));
    hipError_t e = hipMalloc((void **)&acc, nb * 8);
    if (e != hipSuccess) { hipFree(bc); fmd_set_hip_error(e, "hipMalloc(scan)"); return FMD_E_NOMEM; }
    k_fill_lookahead<<<nblk(nb, 256), 256>>>(h->blocks, nb);
    k_block_counts<<<nblk(nb, 256), 256>>>(h->blocks, nb, bc);
    int rc = FMD_OK;
    uint64_t last[6] = {0, 0, 0, 0, 0, 0};
    for (int s = 0; s < 6 && rc == FMD_OK; ++s) { // one symbol at a time through the same buffer
        rc = scan_counts(bc + (uint64_t)s * nb, acc, nb, 0);
        if (rc != FMD_OK) break;
        k_write_meta_sym<<<nblk(nb, 256), 256>>>(h->blocks, nb, acc, s);
        // marginal count = prefix at the pad block
        if (hipMemcpy(&last[s], acc + (nb - 1), 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FMD_E_HIP;
    }
    if (rc == FMD_OK) {
        // positions past the end are '$'-coded zeros: correct mcnt[1] for them
        const uint64_t pad = (nb - 1) * FMD_BLK_STRIDE - h->mcnt[0];
        last[0] -= pad;
        h->mcnt[1] = last[0];
        for (int s = 1; s < 6; ++s) h->mcnt[s + 1] = last[s];
        h->cnt[0] = 0;
        for (int s = 1; s < 7; ++s) h->cnt[s] = h->cnt[s - 1] + h->mcnt[s];
        e = hipDeviceSynchronize();
        if (e != hipSuccess) { fmd_set_hip_error(e, "transcode"); rc = FMD_E_HIP; }
        else if (h->cnt[6] != h->mcnt[0]) rc = FMD_E_FORMAT;
    }
    hipFree(bc); hipFree(acc);
    if (rc == FMD_OK) rc = build_ptab(h);
    if (rc == FMD_OK) rc = build_tail(h);
    return rc;
}

// for the in-place builder (fmd_build.hip): an empty index of n symbols, slices of the BWT OR-ed into its planes in any
// number of calls (positions ascending, one slice after the other on one stream), then the counts
int fmd_index_alloc(int device, uint64_t n_sym, fmd_dev **out)
{
    int rc = dev_alloc_index(device, n_sym, out);
    if (rc == FMD_OK) (*out)->mcnt[0] = n_sym;
    return rc;
}
int fmd_index_finish(fmd_dev *h) { return finish_index(h); }
// one thread per 32-position word that the slice [first, first + m) touches; border words are shared with the
// neighbouring slices, which are written before / after this launch on the same stream: plain OR
__global__ void k_slice_to_planes(const uint8_t *__restrict__ slice, uint64_t first, uint64_t m, uint4 *__restrict__ blocks)
{
    const uint64_t w0 = first >> 5, w1 = (first + m - 1) >> 5;
    for (uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= w1; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = w << 5;
        uint32_t a = 0, b = 0, d = 0;
        for (int i = 0; i < 32; ++i) {
            const uint64_t p = p0 + (uint64_t)i;
            if (p < first || p >= first + m) continue;
            const uint32_t s = slice[p - first] & 7;
            a |= (s & 1) << i; b |= ((s >> 1) & 1) << i; d |= ((s >> 2) & 1) << i;
        }
        uint4 *dst = blocks + fmd_word_u4(w);
        dst->x |= a; dst->y |= b; dst->z |= d;
    }
}
int fmd_index_put_slice(fmd_dev *h, hipStream_t st, const uint8_t *d_slice, uint64_t first, uint64_t m)
{
    if (m == 0) return FMD_OK;
    if (first + m > h->mcnt[0]) return FMD_E_ARG;
    const uint64_t words = ((first + m - 1) >> 5) - (first >> 5) + 1;
    k_slice_to_planes<<<nblk(words, 256), 256, 0, st>>>(d_slice, first, m, h->blocks);
    return hipGetLastError() == hipSuccess ? FMD_OK : FMD_E_HIP;
}

extern "C" int fmd_dev_open_bwt_dev(int device, const uint8_t *d_bwt, uint64_t n, fmd_dev_t **out)
{
    if (!d_bwt || !out) return FMD_E_ARG;
    fmd_dev *h = nullptr;
    int rc = dev_alloc_index(device, n, &h);
    if (rc) return rc;
    h->mcnt[0] = n;
    const uint64_t n_chunks = (n + 31) / 32;
    k_bwt_to_planes<<<nblk(n_chunks, 256), 256>>>(d_bwt, n, h->blocks, n_chunks);
    rc = finish_index(h);
    if (rc) { fmd_dev_close(h); return rc; }
    *out = h;
    return FMD_OK;
}

extern "C" int fmd_dev_open_bwt(int device, const uint8_t *bwt, uint64_t n, fmd_dev_t **out)
{
    if (!bwt || !out) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    uint8_t *d = nullptr;
    FMD_HIP_TRY(hipMalloc((void **)&d, n + 64));
    hipError_t e = hipMemcpy(d, bwt, n, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? fmd_dev_open_bwt_dev(device, d, n, out) : FMD_E_HIP;
    hipFree(d);
    return rc;
}

extern "C" int fmd_dev_open_rle6(int device, const uint8_t *runs, uint64_t n_bytes, fmd_dev_t **out)
{
    if (!runs || !out || n_bytes == 0) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    uint8_t *d_runs = nullptr; uint64_t *d_len = nullptr, *d_start = nullptr, *d_tot = nullptr;
    int rc = FMD_OK;
    fmd_dev *h = nullptr;
    uint64_t n_sym = 0;
    FMD_HIP_TRY(hipMalloc((void **)&d_runs, n_bytes));
    if (hipMalloc((void **)&d_len, n_bytes * 8) != hipSuccess || hipMalloc((void **)&d_start, n_bytes * 8) != hipSuccess ||
        hipMalloc((void **)&d_tot, 8) != hipSuccess) { rc = FMD_E_NOMEM; goto done; }
    hipMemcpy(d_runs, runs, n_bytes, hipMemcpyHostToDevice);
    k_rle6_len<<<nblk(n_bytes, 256), 256>>>(d_runs, n_bytes, d_len);
    rc = scan_u64(d_len, d_start, n_bytes, 0);
    if (rc) goto done;
    { // total = start[last] + len[last]
        uint64_t a, b;
        hipMemcpy(&a, d_start + n_bytes - 1, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&b, d_len + n_bytes - 1, 8, hipMemcpyDeviceToHost);
        n_sym = a + b;
    }
    rc = dev_alloc_index(device, n_sym, &h);
    if (rc) goto done;
    h->mcnt[0] = n_sym;
    k_rle6_scatter<<<nblk(n_bytes, 256), 256>>>(d_runs, n_bytes, d_start, (uint32_t *)h->blocks, d_tot);
    rc = finish_index(h);
done:
    hipFree(d_runs); hipFree(d_len); hipFree(d_start); hipFree(d_tot);
    if (rc) { if (h) fmd_dev_close(h); return rc; }
    *out = h;
    return FMD_OK;
}

// The payload words of an RLD\2 file, already in device memory ((n_words / 8 + 1) * 64 bytes, zero behind the payload): the index.  Frees d_w.
static int open_rld_words_dev(int device, uint64_t *d_w, uint64_t n_words, const uint64_t mcnt[7], fmd_dev_t **out)
{
    // blocks 0 .. last/8-1 carry payload; the block at word `last` is header-only (rld.h:64)
    const uint64_t n_rld = n_words / 8;
    uint64_t *d_size = nullptr, *d_start = nullptr, *d_tot = nullptr;
    int rc = FMD_OK;
    fmd_dev *h = nullptr;
    if (hipMalloc((void **)&d_size, n_rld * 8) != hipSuccess || hipMalloc((void **)&d_start, n_rld * 8) != hipSuccess ||
        hipMalloc((void **)&d_tot, 16) != hipSuccess) { rc = FMD_E_NOMEM; goto done; }
    hipMemset(d_tot, 0, 16);
    k_rld_sizes<<<nblk(n_rld, 256), 256>>>(d_w, n_rld, d_size);
    rc = scan_u64(d_size, d_start, n_rld, 0);
    if (rc) goto done;
    rc = dev_alloc_index(device, mcnt[0], &h);
    if (rc) goto done;
    h->mcnt[0] = mcnt[0];
    k_rld_scatter<<<nblk(n_rld, 64), 64>>>(d_w, n_rld, d_start, (uint32_t *)h->blocks, mcnt[0], d_tot);
    {
        uint64_t tot[2] = {0, 0};
        hipMemcpy(tot, d_tot, 16, hipMemcpyDeviceToHost);
        if (tot[1] || tot[0] != mcnt[0]) { rc = FMD_E_FORMAT; goto done; }
    }
    rc = finish_index(h);
    if (rc == FMD_OK)
        for (int s = 1; s < 7; ++s) if (h->mcnt[s] != mcnt[s]) rc = FMD_E_FORMAT; // header vs decoded stream
done:
    hipFree(d_w); hipFree(d_size); hipFree(d_start); hipFree(d_tot);
    if (rc) { if (h) fmd_dev_close(h); return rc; }
    *out = h;
    return FMD_OK;
}

extern "C" int fmd_dev_open_rld(int device, const uint64_t *payload, uint64_t n_words, const uint64_t mcnt[7], fmd_dev_t **out)
{
    if (!payload || !out || !mcnt || n_words < 10) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    const uint64_t n_rld = n_words / 8;
    if (n_rld == 0) return FMD_E_FORMAT;
    uint64_t *d_w = nullptr;
    FMD_HIP_TRY(hipMalloc((void **)&d_w, (n_rld + 1) * 64));
    hipMemset(d_w, 0, (n_rld + 1) * 64);
    hipMemcpy(d_w, payload, n_words * 8, hipMemcpyHostToDevice);
    return open_rld_words_dev(device, d_w, n_words, mcnt, out);
}

// The payload of a large .fmd goes from the file to the device in pieces, several threads each reading a piece into its own pinned buffer and sending it
// on its own stream: one fread into fresh memory and one pageable copy of the whole (2.5 GB for 5*10^7 raw reads) were most of "index load + transcode".
static int upload_payload(int device, int fd, off_t at, uint64_t bytes, uint8_t *d_dst)
{
    size_t CH = (size_t)32 << 20;
    { const char *e = getenv("FMD_LOAD_CHUNK"); if (e && atoll(e) >= 64) CH = (size_t)atoll(e) / 64 * 64; }   // (tests: many pieces of a small file)
    const int T = 4;
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    const uint64_t n_ch = (bytes + CH - 1) / CH;
    auto work = [&]() {
        void *stage = nullptr; hipStream_t st = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&stage, CH, hipHostMallocDefault) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { failed = FMD_E_HIP; if (stage) hipHostFree(stage); return; }
        for (;;) {
            const uint64_t c = next.fetch_add(1);
            if (c >= n_ch || failed) break;
            const uint64_t off = c * CH, len = bytes - off < CH ? bytes - off : CH;
            uint64_t got = 0;
            while (got < len) { const ssize_t k = pread(fd, (char *)stage + got, len - got, at + (off_t)(off + got)); if (k <= 0) { failed = FMD_E_IO; break; } got += (uint64_t)k; }
            if (failed) break;
            if (hipMemcpyAsync(d_dst + off, stage, len, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { failed = FMD_E_HIP; break; }
        }
        hipStreamDestroy(st); hipHostFree(stage);
    };
    std::vector<std::thread> th;
    for (int k = 1; k < T && (uint64_t)k < n_ch; ++k) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return failed.load();
}

// .fmd file: header = "RLD\2", u32 asize<<16|sbits, u64 0, u64 n_bytes, u64 n_frames, u64 mcnt[1..6]
// (rld.c:242-263); anything else is treated as a raw run-length byte stream after a 4-byte
// magic, as rld_restore does (rld.c:295-308).
extern "C" int fmd_dev_open_file(int device, const char *fn, fmd_dev_t **out)
{
    if (!fn || !out) return FMD_E_ARG;
    FILE *fp = fopen(fn, "rb");
    if (!fp) return FMD_E_IO;
    char magic[4];
    int rc;
    if (fread(magic, 1, 4, fp) != 4) { fclose(fp); return FMD_E_FORMAT; }
    if (memcmp(magic, "RLD\2", 4) == 0) {
        uint32_t a; uint64_t hdr[3], mcnt[7];
        if (fread(&a, 4, 1, fp) != 1 || fread(hdr, 8, 3, fp) != 3 || fread(mcnt + 1, 8, 6, fp) != 6) { fclose(fp); return FMD_E_FORMAT; }
        if ((a >> 16) != 6 || (a & 0xffff) != 3 || (hdr[1] & 7)) { fclose(fp); return FMD_E_FORMAT; }
        mcnt[0] = 0;
        for (int s = 1; s < 7; ++s) mcnt[0] += mcnt[s];
        const uint64_t n_words = hdr[1] / 8;
        {   // the header's payload size against the file itself, before anything is sized by it
            const long at = ftell(fp);
            fseek(fp, 0, SEEK_END);
            const long sz = ftell(fp);
            fseek(fp, at, SEEK_SET);
            if (at < 0 || sz < at || hdr[1] > (uint64_t)(sz - at)) { fclose(fp); return FMD_E_FORMAT; }
        }
        if (n_words < 10 || n_words / 8 == 0) { fclose(fp); return FMD_E_FORMAT; }
        if (fmd_device_count() <= 0) { fclose(fp); return FMD_E_NODEV; }
        if (hipSetDevice(device) != hipSuccess) { fclose(fp); fmd_set_hip_error(hipGetLastError(), "hipSetDevice"); return FMD_E_HIP; }
        {
            const uint64_t n_rld = n_words / 8;
            const off_t at = (off_t)ftell(fp);
            uint64_t *d_w = nullptr;
            if (hipMalloc((void **)&d_w, (n_rld + 1) * 64) != hipSuccess) { fclose(fp); (void)hipGetLastError(); return FMD_E_NOMEM; }
            hipMemset((uint8_t *)d_w + n_words * 8, 0, (n_rld + 1) * 64 - n_words * 8);   // behind the payload
            rc = upload_payload(device, fileno(fp), at, n_words * 8, (uint8_t *)d_w);
            fclose(fp); // the rank frames that follow are not needed: the device layout has none
            if (rc) { hipFree(d_w); return rc; }
            return open_rld_words_dev(device, d_w, n_words, mcnt, out);
        }
    } else {
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        if (sz <= 4) { fclose(fp); return FMD_E_FORMAT; }
        fseek(fp, 4, SEEK_SET);
        uint8_t *buf = (uint8_t *)malloc((size_t)sz - 4);
        if (!buf) { fclose(fp); return FMD_E_NOMEM; }
        if (fread(buf, 1, (size_t)sz - 4, fp) != (size_t)sz - 4) { free(buf); fclose(fp); return FMD_E_IO; }
        fclose(fp);
        rc = fmd_dev_open_rle6(device, buf, (uint64_t)sz - 4, out);
        free(buf);
        return rc;
    }
}

// ---- inspection: what `fermi chkbwt -p / -r` and the tests need --------------------------------
// one thread per 32-position chunk: the plane words back into one nt6 byte per position
__global__ void k_planes_to_bwt(const uint4 *__restrict__ blocks, uint64_t first, uint64_t n, uint8_t *__restrict__ out)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // chunk relative to first/32
    const uint64_t p0 = (first & ~31ull) + c * 32;
    if (p0 >= first + n) return;
    const uint4 v = blocks[fmd_word_u4(p0 >> 5)];
    for (int i = 0; i < 32; ++i) {
        const uint64_t p = p0 + i;
        if (p >= first && p < first + n) out[p - first] = (uint8_t)(((v.x >> i) & 1) | ((v.y >> i) & 1) << 1 | ((v.z >> i) & 1) << 2);
    }
}

extern "C" int fmd_dev_export_bwt(fmd_dev_t *h, uint64_t first, uint64_t n, uint8_t *bwt)
{
    if (!h || (n && !bwt) || first > h->mcnt[0] || n > h->mcnt[0] - first) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    uint8_t *d = nullptr;
    FMD_HIP_TRY(hipMalloc((void **)&d, n));
    const uint64_t n_chunks = (first + n + 31) / 32 - first / 32;
    k_planes_to_bwt<<<nblk(n_chunks, 256), 256>>>(h->blocks, first, n, d);
    hipError_t e = hipMemcpy(bwt, d, n, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) { fmd_set_hip_error(e, "export"); return FMD_E_HIP; }
    return FMD_OK;
}

// rank self-check (chkbwt -r, cmd.c:86-101): for every position k the six counts of BWT[0..k] taken
// from the block's absolute counts + popcounts must equal those of k-1 plus the symbol at k, and the
// last position must give the marginal counts.  One thread per position, blocks read in order.
__global__ void k_check_rank(FmdIndexView ix, unsigned long long *__restrict__ bad /* [0] count, [1] first position */)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < ix.n_sym; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t a[6], b[6] = {0, 0, 0, 0, 0, 0};
        uint32_t bn, off;
        fmd_split(k, bn, off);
        const int c = fmd_block_rank6<true>(ix.blocks + (size_t)bn * FMD_BLK_U4, 0, off + 1, a, bn);
        if (k) { fmd_split(k - 1, bn, off); fmd_block_rank6<false>(ix.blocks + (size_t)bn * FMD_BLK_U4, 0, off + 1, b, bn); }
        bool ok = c >= 0 && c < 6;
        for (int j = 0; j < 6; ++j) ok = ok && a[j] == b[j] + (j == c ? 1u : 0u);
        if (k == ix.n_sym - 1) for (int j = 0; j < 6; ++j) ok = ok && a[j] == ix.cnt[j + 1] - ix.cnt[j];
        if (!ok) { atomicAdd(&bad[0], 1ull); atomicMin(&bad[1], (unsigned long long)k); }
    }
}

extern "C" int fmd_dev_check_rank(fmd_dev_t *h, uint64_t *n_bad, uint64_t *first_bad)
{
    if (!h || !n_bad || !first_bad) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    unsigned long long *d = nullptr, init[2] = {0, ~0ull}, res[2];
    FMD_HIP_TRY(hipMalloc((void **)&d, 16));
    FMD_HIP_TRY(hipMemcpy(d, init, 16, hipMemcpyHostToDevice));
    const uint64_t n = h->mcnt[0];
    if (n) k_check_rank<<<nblk(n, 256), 256>>>(fmd_view(h), d);
    hipError_t e = hipMemcpy(res, d, 16, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) { fmd_set_hip_error(e, "check_rank"); return FMD_E_HIP; }
    *n_bad = res[0]; *first_bad = res[1];
    return FMD_OK;
}

extern "C" void fmd_dev_close(fmd_dev_t *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    hipFree(h->blocks);
    hipFree(h->ptab);
    hipFree(h->tail);
    hipFree(h->pair);
    hipFree(h->pair_tab);
    hipFree(h->queues);
    hipFree(h->stat);
    for (size_t i = 0; i < sizeof(h->scratch) / sizeof(h->scratch[0]); ++i) if (h->scratch[i].p) hipFree(h->scratch[i].p);
    if (h->slow_ready) {
        hipStreamDestroy(h->slow_stream);
        for (int i = 0; i < 2; ++i) hipEventDestroy(h->slow_ev[i]);
    }
    if (h->aux_ready) {
        hipStreamDestroy(h->aux_stream);
        for (int i = 0; i <= FMD_OVLP_MAX_PARTS; ++i) hipEventDestroy(h->aux_ev[i]);
    }
    free(h);
}

extern "C" int fmd_dev_info(const fmd_dev_t *h, fmd_info_t *info)
{
    if (!h || !info) return FMD_E_ARG;
    memcpy(info->cnt, h->cnt, sizeof(h->cnt));
    memcpy(info->mcnt, h->mcnt, sizeof(h->mcnt));
    info->n_blocks = h->n_blocks;
    info->hbm_bytes = h->bytes;
    info->device = h->device;
    return FMD_OK;
}

// Rank blocks (and other random lines) the kernels launched on this handle have requested since the last reset.
// The shipped library does not count (returns zeros and counting = 0); libfmdhip_count.so, the same sources built with
// -DFMD_COUNT_LINES=1, does.  Synchronises the device.
extern "C" int fmd_dev_line_count(fmd_dev_t *h, uint64_t lines[2], int reset, int *counting)
{
    if (!h || !lines) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    unsigned long long host[FMD_STAT_SLOTS * FMD_STAT_STRIDE];
    FMD_HIP_TRY(hipDeviceSynchronize());
    FMD_HIP_TRY(hipMemcpy(host, h->stat, sizeof(host), hipMemcpyDeviceToHost));
    lines[0] = lines[1] = 0;
    for (int i = 0; i < FMD_STAT_SLOTS; ++i) { lines[0] += host[i * FMD_STAT_STRIDE]; lines[1] += host[i * FMD_STAT_STRIDE + 1]; }
    if (reset) FMD_HIP_TRY(hipMemset(h->stat, 0, sizeof(host)));
    if (counting) *counting = FMD_COUNT_LINES;
    return FMD_OK;
}

// the same with the 128-byte two-base blocks as a third number (fmd_pair.hip; lines[2])
extern "C" int fmd_dev_line_count3(fmd_dev_t *h, uint64_t lines[3], int reset, int *counting)
{
    if (!h || !lines) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    unsigned long long host[FMD_STAT_SLOTS * FMD_STAT_STRIDE];
    FMD_HIP_TRY(hipDeviceSynchronize());
    FMD_HIP_TRY(hipMemcpy(host, h->stat, sizeof(host), hipMemcpyDeviceToHost));
    lines[0] = lines[1] = lines[2] = 0;
    for (int i = 0; i < FMD_STAT_SLOTS; ++i) { lines[0] += host[i * FMD_STAT_STRIDE]; lines[1] += host[i * FMD_STAT_STRIDE + 1]; lines[2] += host[i * FMD_STAT_STRIDE + 2]; }
    if (reset) FMD_HIP_TRY(hipMemset(h->stat, 0, sizeof(host)));
    if (counting) *counting = FMD_COUNT_LINES;
    return FMD_OK;
}

extern "C" int fmd_dev_malloc(int device, size_t bytes, void **d_ptr)
{
    if (!d_ptr) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) { fmd_set_hip_error(e, "hipMalloc"); return FMD_E_NOMEM; }
    return FMD_OK;
}
extern "C" int fmd_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    FMD_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    FMD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return FMD_OK;
}
extern "C" int fmd_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    FMD_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    FMD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return FMD_OK;
}

extern "C" int fmd_dev_sync(const fmd_dev_t *h, void *stream)
{
    if (!h) return FMD_E_ARG;
    FMD_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return FMD_OK;
}

static void scratch_lock(fmd_dev *h) { int e = 0; while (!__atomic_compare_exchange_n(&h->scratch_lock, &e, 1, false, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) e = 0; }
static void scratch_unlock(fmd_dev *h) { __atomic_store_n(&h->scratch_lock, 0, __ATOMIC_RELEASE); }
void *fmd_scratch_acquire(fmd_dev *h, size_t bytes)
{
    const int N = (int)(sizeof(h->scratch) / sizeof(h->scratch[0]));
    if (bytes == 0) bytes = 16;
    scratch_lock(h);
    int best = -1, empty = -1, victim = -1;
    for (int i = 0; i < N; ++i) {
        if (!h->scratch[i].p) { if (empty < 0) empty = i; continue; }
        if (h->scratch[i].busy) continue;
        if (h->scratch[i].bytes >= bytes && (best < 0 || h->scratch[i].bytes < h->scratch[best].bytes)) best = i;
        if (victim < 0 || h->scratch[i].bytes < h->scratch[victim].bytes) victim = i;
    }
    if (best >= 0 && h->scratch[best].bytes <= 2 * bytes + ((size_t)64 << 20)) { // do not hand a 26 GB buffer to a 1 MB request
        h->scratch[best].busy = 1;
        void *p = h->scratch[best].p;
        scratch_unlock(h);
        return p;
    }
    if (empty < 0 && victim >= 0) { hipFree(h->scratch[victim].p); h->scratch[victim].p = nullptr; empty = victim; } // table full: drop the smallest idle one
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        for (int i = 0; i < N; ++i) if (h->scratch[i].p && !h->scratch[i].busy) { hipFree(h->scratch[i].p); h->scratch[i].p = nullptr; if (empty < 0) empty = i; } // make room, try once more
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    }
    if (p && empty >= 0) { h->scratch[empty].p = p; h->scratch[empty].bytes = bytes; h->scratch[empty].busy = 1; }
    else if (p) { /* no slot: an untracked buffer, freed on release */ }
    scratch_unlock(h);
    return p;
}
void fmd_scratch_release(fmd_dev *h, void *p)
{
    if (!p) return;
    const int N = (int)(sizeof(h->scratch) / sizeof(h->scratch[0]));
    scratch_lock(h);
    for (int i = 0; i < N; ++i) if (h->scratch[i].p == p) { h->scratch[i].busy = 0; scratch_unlock(h); return; }
    scratch_unlock(h);
    hipFree(p);
}

// The buffers the handle keeps between calls (the work areas of the host-buffer entries: fmd_*_batch, the table jobs) go back to the device; the next call
// that needs one allocates it again.  Synchronises the device first.  -> bytes released.
extern "C" uint64_t fmd_dev_trim(fmd_dev_t *h)
{
    if (!h || hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    (void)hipDeviceSynchronize();
    const int N = (int)(sizeof(h->scratch) / sizeof(h->scratch[0]));
    uint64_t freed = 0;
    scratch_lock(h);
    for (int i = 0; i < N; ++i)
        if (h->scratch[i].p && !h->scratch[i].busy) { hipFree(h->scratch[i].p); freed += h->scratch[i].bytes; h->scratch[i].p = nullptr; h->scratch[i].bytes = 0; }
    scratch_unlock(h);
    (void)hipGetLastError();
    return freed;
}

uint32_t *fmd_next_queue(fmd_dev *h, hipStream_t stream)
{
    const uint32_t i = __atomic_fetch_add(&h->queue_next, 1u, __ATOMIC_RELAXED) % FMD_N_QUEUES;
    hipMemsetAsync(h->queues + i, 0, sizeof(uint32_t), stream);
    return h->queues + i;
}

int fmd_grid_for(const fmd_dev *h, size_t n_items)
{
    const size_t waves_needed = (n_items + 63) / 64;
    size_t per_cu = (160 * 1024) / (FMD_WAVE_LDS_U4 * 16);  // two dense slots per wave: 10 waves per CU with 128-byte blocks
    if (per_cu > 16) per_cu = 16;
    const size_t resident = (size_t)h->n_cu * per_cu;
    size_t g = waves_needed < resident ? waves_needed : resident;
    return (int)(g ? g : 1);
}

int fmd_grid_for_lds(const fmd_dev *h, size_t n_items, size_t lds_bytes)
{
    const size_t waves_needed = (n_items + 63) / 64;
    size_t per_cu = (160 * 1024) / (lds_bytes ? lds_bytes : 1);
    static const size_t cap = getenv("FMD_WAVES_PER_CU") ? (size_t)atoi(getenv("FMD_WAVES_PER_CU")) : 16;
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    const size_t resident = (size_t)h->n_cu * per_cu;
    size_t g = waves_needed < resident ? waves_needed : resident;
    return (int)(g ? g : 1);
}
