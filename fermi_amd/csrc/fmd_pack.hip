// fmd_pack.hip -- compact form of a finished overlap batch: what leaves the GPU.
//
// fmd_ovlp_dev works on fixed-stride rows (64-byte record, max_nei x 32-byte neighbours, seq_stride bytes of
// sequence: 448 bytes per strand at the defaults).  That is the right shape for the kernels and the wrong one for
// a link: the unitig walk (unitig.c:227-317, host) reads from a row its record, its n_nei neighbours (one, unless
// the read set forks) and len + ext_len bases.  fmd_ovlp_pack_dev rewrites a batch as
//     prec[i]        the record, unchanged except for FMD_OVLP_F_PACK4 in `flags`
//     var + off[i]   min(n_nei, max_nei) neighbours (32 bytes each), then len + ext_len bases, 4 per byte
//                    (2 per byte when the row holds a base other than A/C/G/T: FMD_OVLP_F_PACK4), padded to 8 bytes;
//                    nothing at all for rows the walk never opens (status != 0 or an overflowed record)
// ~125 bytes per strand on 100-bp reads.  The same rows travel over PCIe to the host walk (fmdh_ovlp_table_build)
// and over xGMI to rank 0 in the multi-process form (the one RCCL exchange of the pipeline).
#include <stdlib.h>
#include <time.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <errno.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "fmd_prim.h"
#include "fmd_kernel_common.h"

// bytes of row i's variable part; *pack4 = the sequence needs 4 bits per base
__device__ __forceinline__ uint32_t pack_row_bytes(const fmd_ovlp_rec_t &r, uint32_t max_nei, const uint8_t *seq, uint32_t seq_stride, bool &pack4)
{
    pack4 = false;
    if (r.status != 0 || (r.flags & FMD_OVLP_F_OVERFLOW)) return 0;
    uint32_t nb = (uint32_t)r.len + (uint32_t)r.ext_len;
    if (nb > seq_stride) nb = seq_stride;
    for (uint32_t j = 0; j < nb && !pack4; j += 4) { // rows are 4-byte aligned (the host entry checks seq_stride % 4 == 0)
        const uint32_t w = *(const uint32_t *)(seq + j);
#pragma unroll
        for (int b = 0; b < 4; ++b) { const uint32_t c = (w >> (8 * b)) & 0xff; if (j + b < nb && (c < 1 || c > 4)) pack4 = true; }
    }
    const uint32_t nn = (uint32_t)r.n_nei < max_nei ? (uint32_t)r.n_nei : max_nei;
    const uint32_t sb = pack4 ? (nb + 1) / 2 : (nb + 3) / 4;
    return nn * 32 + ((sb + 7) & ~7u);
}

// ---- the two pack kernels: 8 lanes per row, the sequence rows read 8 bytes per lane and step (coalesced 64-byte bursts), bases
// compressed in registers (SWAR): the pack runs beside the discovery kernels of the next piece (fmd_ovlp_dist.hip) and must be light
__device__ __forceinline__ uint64_t bytes_outside_1_4(uint64_t w)   // bit 7 of every byte that is not 1, 2, 3 or 4 (the others zero)
{
    const uint64_t H = 0x8080808080808080ull, hi = w & 0xF8F8F8F8F8F8F8F8ull;          // any of bits 3..7 set: >= 8
    const uint64_t big = ((hi | (hi << 1) | (hi << 2) | (hi << 3) | (hi << 4)) & H);      // -> bit 7 of such bytes
    const uint64_t lo = w & 0x0707070707070707ull;                                      // low 3 bits: 0..7, no carries between bytes below
    const uint64_t ge5 = ((lo + 0x0303030303030303ull) & 0x0808080808080808ull) << 4;    // b + 3 >= 8  <=>  b >= 5
    const uint64_t zero = (~(lo + 0x0707070707070707ull) & 0x0808080808080808ull) << 4;  // b + 7 < 8   <=>  b == 0
    return big | ge5 | zero;
}
__device__ __forceinline__ uint32_t squeeze2(uint64_t w)   // 8 bases (one per byte, codes 1..4) -> 16 bits, first base lowest: (b - 1) & 3 each
{
    uint64_t y = (w - 0x0101010101010101ull) & 0x0303030303030303ull;   // (bytes 1..4: no borrow crosses a byte; other bytes are masked by the caller)
    y = (y | (y >> 6)) & 0x000F000F000F000Full;
    y = (y | (y >> 12)) & 0x000000FF000000FFull;
    y = (y | (y >> 24)) & 0xFFFFull;
    return (uint32_t)y;
}
// sizes[t] = bytes of output row t's variable part, p4[t] = its sequence needs 4 bits per base.  (rows != nullptr: output row t is row
// rows[t] of the fixed-stride arrays -- a piece of a sorted job, fmd_ovlp_pack_rows_dev)
__global__ void k_pack_sizes(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, uint32_t max_nei, const uint8_t *__restrict__ seq, uint32_t seq_stride,
                             uint64_t *__restrict__ sizes, uint8_t *__restrict__ p4, const uint32_t *__restrict__ rows)
{
    const size_t step = (size_t)gridDim.x * (blockDim.x >> 3);
    const int l8 = threadIdx.x & 7;
    for (size_t t0 = (size_t)blockIdx.x * (blockDim.x >> 3); t0 <= n; t0 += step) {   // (whole groups stay together for the shuffles)
        const size_t t = t0 + (threadIdx.x >> 3);
        uint64_t bad = 0, bytes = 0;
        if (t < n) {
            const size_t i = rows ? (size_t)rows[t] : t;
            const int32_t status = rec[i].status; const uint32_t flags = rec[i].flags;
            if (status == 0 && !(flags & FMD_OVLP_F_OVERFLOW)) {
                uint32_t nb = (uint32_t)rec[i].len + (uint32_t)rec[i].ext_len;
                if (nb > seq_stride) nb = seq_stride;
                const uint8_t *s = seq + i * (size_t)seq_stride;
                if (((uintptr_t)s & 7) == 0)
                    for (uint32_t w = l8; 8 * w < nb; w += 8) {
                        uint64_t v = *(const uint64_t *)(s + 8 * w);
                        const uint32_t left = nb - 8 * w;
                        if (left < 8) v = (v & ((1ull << (8 * left)) - 1)) | (0x0101010101010101ull << (8 * left));   // bytes past the end count as 'A'
                        bad |= bytes_outside_1_4(v);
                    }
                else
                    for (uint32_t j = l8; j < nb; j += 8) { const uint32_t c = s[j]; bad |= (uint64_t)(c < 1 || c > 4); }
                bytes = 1;
            }
        }
        uint32_t b32 = (uint32_t)((bad | (bad >> 32)) != 0);
        b32 |= __shfl_xor(b32, 1); b32 |= __shfl_xor(b32, 2); b32 |= __shfl_xor(b32, 4);
        if (l8 == 0 && t <= n) {
            uint64_t sz = 0;
            if (bytes) {
                const size_t i = rows ? (size_t)rows[t] : t;
                uint32_t nb = (uint32_t)rec[i].len + (uint32_t)rec[i].ext_len;
                if (nb > seq_stride) nb = seq_stride;
                const uint32_t nn = (uint32_t)rec[i].n_nei < max_nei ? (uint32_t)rec[i].n_nei : max_nei;
                const uint32_t sb = b32 ? (nb + 1) / 2 : (nb + 3) / 4;
                sz = nn * 32 + ((sb + 7) & ~7u);
            }
            sizes[t] = sz;
            if (t < n) p4[t] = (uint8_t)(bytes ? b32 : 0);
        }
    }
}

// one 8-lane group per row: record (64 bytes = 8 lanes x 8), neighbours, then the packed bases 8 bytes per lane
__global__ void k_pack_rows(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, const fmd_intv_t *__restrict__ nei, uint32_t max_nei,
                            const uint8_t *__restrict__ seq, uint32_t seq_stride, fmd_ovlp_rec_t *__restrict__ prec,
                            const uint64_t *__restrict__ off, const uint8_t *__restrict__ p4, uint8_t *__restrict__ var, uint64_t var_cap, const uint32_t *__restrict__ rows,
                            const uint64_t *__restrict__ row_ids, uint64_t id_first, uint64_t id_step, uint32_t *__restrict__ pid)
{
    const size_t step = (size_t)gridDim.x * (blockDim.x >> 3);
    const int l8 = threadIdx.x & 7;
    for (size_t t = (size_t)blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); t < n; t += step) {
        const size_t i = rows ? (size_t)rows[t] : t;   // source row; everything written goes to row t
        if (pid && l8 == 0) pid[t] = (uint32_t)(row_ids ? row_ids[i] : id_first + id_step * i);
        const uint64_t o = off[t], sz = off[t + 1] - o;
        const bool fits = o + sz <= var_cap;
        const bool pack4 = p4[t] != 0;
        uint64_t w7;
        {   // the record: lane l8 copies word l8; flags live in word 7 (low half)
            uint64_t w = ((const uint64_t *)(rec + i))[l8];
            if (l8 == 7 && pack4) w |= (uint64_t)FMD_OVLP_F_PACK4;
            ((uint64_t *)(prec + t))[l8] = w;
            w7 = w;
        }
        if (!sz || !fits) continue;
        // len, ext_len, n_nei from the record words the lanes hold: word 4 = len | status << 32, word 6 = ext_len | n_nei << 32
        const int g0 = (int)(threadIdx.x & 63u & ~7u);
        const uint64_t w4 = __shfl(w7, g0 + 4), w6 = __shfl(w7, g0 + 6);
        uint32_t nb = (uint32_t)w4 + (uint32_t)w6;
        if (nb > seq_stride) nb = seq_stride;
        const uint32_t nnr = (uint32_t)(w6 >> 32);
        const uint32_t nn = nnr < max_nei ? nnr : max_nei;
        const uint32_t sb8 = (uint32_t)sz - nn * 32;          // padded sequence bytes
        uint8_t *dst = var + o;
        for (uint32_t w = l8; w < nn * 4; w += 8) ((uint64_t *)dst)[w] = ((const uint64_t *)(nei + i * (size_t)max_nei))[w];
        dst += nn * 32;
        const uint8_t *s = seq + i * (size_t)seq_stride;
        const bool al = ((uintptr_t)s & 7) == 0;
        for (uint32_t w = l8; w < sb8 / 8; w += 8) {   // output word w = bases [32w, 32w+32) (2-bit) or [16w, 16w+16) (4-bit)
            uint64_t v = 0;
            if (pack4) {
                for (int b = 0; b < 16; ++b) { const uint32_t j = 16 * w + b; if (j < nb) v |= (uint64_t)(s[j] & 15) << (4 * b); }
            } else if (al) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t j = 32 * w + 8 * q;
                    if (j < nb) {
                        uint64_t x = *(const uint64_t *)(s + j);
                        const uint32_t left = nb - j;
                        if (left < 8) x = (x & ((1ull << (8 * left)) - 1)) | (0x0101010101010101ull << (8 * left));
                        v |= (uint64_t)squeeze2(x) << (16 * q);
                    }
                }
            } else {
                for (int b = 0; b < 32; ++b) { const uint32_t j = 32 * w + b; if (j < nb) v |= (uint64_t)((s[j] - 1) & 3) << (2 * b); }
            }
            ((uint64_t *)dst)[w] = v;
        }
    }
}

extern "C" size_t fmd_ovlp_pack_max_bytes(size_t n, uint32_t max_nei, uint32_t seq_stride)
{
    return n * ((size_t)max_nei * 32 + (((size_t)seq_stride + 1) / 2 + 7) / 8 * 8);
}

extern "C" size_t fmd_ovlp_pack_work_bytes(size_t n)
{
    size_t tb = 0;
    fmd_exclusive_sum(nullptr, tb, (uint64_t *)nullptr, (uint64_t *)nullptr, n + 1);
    return (((n + 1) * 8 + 255) & ~(size_t)255) + ((tb + 255) & ~(size_t)255) + ((n + 255) & ~(size_t)255) + 256;   // sizes, the scan's storage, one flag byte per row
}

static int pack_core(fmd_dev_t *h, hipStream_t st, size_t n, const uint32_t *d_rows, const uint64_t *d_row_ids, uint64_t id_first, uint64_t id_step,
                     const fmd_ovlp_rec_t *d_rec, const fmd_intv_t *d_nei, uint32_t max_nei, const uint8_t *d_seq, uint32_t seq_stride, uint32_t *d_pid,
                     fmd_ovlp_rec_t *d_prec, uint64_t *d_off, uint8_t *d_var, uint64_t var_cap, void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_rec || !d_nei || !d_seq || !d_prec || !d_off || !d_var || !d_work)) || max_nei == 0 || (seq_stride & 7)) return FMD_E_ARG;   /* the rows are read as aligned 8-byte words */
    if (work_bytes < fmd_ovlp_pack_work_bytes(n)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    if (n == 0) { FMD_HIP_TRY(hipMemsetAsync(d_off, 0, 8, st)); return FMD_OK; }
    uint64_t *sizes = (uint64_t *)d_work;
    void *tmp = (uint8_t *)d_work + (((n + 1) * 8 + 255) & ~(size_t)255);
    size_t tb = 0;
    FMD_HIP_TRY(fmd_exclusive_sum(nullptr, tb, sizes, d_off, n + 1, st));
    uint8_t *p4 = (uint8_t *)tmp + ((tb + 255) & ~(size_t)255);
    size_t blocks = (n + 1 + 31) / 32;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    k_pack_sizes<<<(unsigned)blocks, 256, 0, st>>>(n, d_rec, max_nei, d_seq, seq_stride, sizes, p4, d_rows);
    FMD_HIP_TRY(fmd_exclusive_sum(tmp, tb, sizes, d_off, n + 1, st));
    blocks = (n + 31) / 32;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    k_pack_rows<<<(unsigned)blocks, 256, 0, st>>>(n, d_rec, d_nei, max_nei, d_seq, seq_stride, d_prec, d_off, p4, d_var, var_cap, d_rows, d_row_ids, id_first, id_step, d_pid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "overlap pack kernels"); return FMD_E_HIP; }
    return FMD_OK;
}

extern "C" int fmd_ovlp_pack_dev(fmd_dev_t *h, void *stream_, size_t n, const fmd_ovlp_rec_t *d_rec, const fmd_intv_t *d_nei, uint32_t max_nei,
                                 const uint8_t *d_seq, uint32_t seq_stride, fmd_ovlp_rec_t *d_prec, uint64_t *d_off, uint8_t *d_var, uint64_t var_cap,
                                 void *d_work, size_t work_bytes)
{
    return pack_core(h, (hipStream_t)stream_, n, nullptr, nullptr, 0, 1, d_rec, d_nei, max_nei, d_seq, seq_stride, nullptr, d_prec, d_off, d_var, var_cap, d_work, work_bytes);
}

extern "C" int fmd_ovlp_pack_rows_dev(fmd_dev_t *h, void *stream_, size_t n, const uint32_t *d_rows, const uint64_t *d_row_ids, uint64_t id_first, uint64_t id_step,
                                      const fmd_ovlp_rec_t *d_rec, const fmd_intv_t *d_nei, uint32_t max_nei, const uint8_t *d_seq, uint32_t seq_stride,
                                      uint32_t *d_pid, fmd_ovlp_rec_t *d_prec, uint64_t *d_off, uint8_t *d_var, uint64_t var_cap, void *d_work, size_t work_bytes)
{
    if (n && (!d_rows || !d_pid)) return FMD_E_ARG;
    return pack_core(h, (hipStream_t)stream_, n, d_rows, d_row_ids, id_first, id_step, d_rec, d_nei, max_nei, d_seq, seq_stride, d_pid, d_prec, d_off, d_var, var_cap, d_work, work_bytes);
}

// ------------------------------------------------------------------------------------------------
// The link pass over a COMPLETE table resident on one device (rows = sequence ids 0 .. n-1): what the unitig walk needs
// to step from a read to the next without touching neighbour lists, and check_left_simple's verdict for every edge
// from the lfork of the neighbour's reverse strand (include/fmd_hip.h).  The host does the same over sharded tables
// (fmdh_ovlp_table_link); on one GPU this is two streaming kernels instead of 0.4 s of host threads.
//   row_of[k]  = smallest id whose `$read$` interval starts at k (identical reads share one interval), ~0 = none
//   link[i]    = {row_of[nei.x0], row_of[nei.x1]} of the unique neighbour, {~0, ~0} otherwise
//   rec[i].reserved: 0 / 1 where lfork decides the edge; 2 (untouched) elsewhere -- those ids are appended to und[], and so are the
//   rows whose neighbour has no row yet (flagged records)
__device__ __forceinline__ int lfork_decide_dev(uint16_t lfork, int rbeg)   // fmd_lfork_decide of include/fmd_hip.h (a host inline there)
{
    const int r = lfork & 0x7fff;
    if (rbeg <= r || r == (int)FMD_LFORK_ALL) return 0;
    return (lfork & 0x8000) ? -1 : 1;
}
// Round 6: the row map carries each row's lfork.  An edge used to cost three random accesses: row_of[x0], row_of[x1] (4 bytes each, mostly from the
// Infinity Cache: the map of 10^8 rows is 400 MB) and then -- DEPENDENT on the second -- the record of the neighbour's reverse strand for its lfork, two
// bytes of a 64-byte line from DRAM (PMC: 2.13 x the bytes the kernel asks for, profiles/r5_final).  The map is now 8 bytes per position, row << 32 | lfork:
// atomicMin on the whole word still elects the smallest row (the row sits in the high half), and the edge's second look-up brings the lfork with the row.
// Two random 8-byte reads per edge, no dependent one; row_of[] as the ABI promises it is a streaming pass over the map.
__global__ void k_link_rows(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, unsigned long long *__restrict__ map)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const fmd_ovlp_rec_t *r = rec + i;
        if (r->status == 0 && !(r->flags & FMD_OVLP_F_OVERFLOW) && r->k[0] < n) atomicMin(map + r->k[0], (unsigned long long)i << 32 | (unsigned long long)r->lfork);
    }
}
// The round-5 form, kept for tables beside which the 8-byte map does not fit (config 5: 1.4*10^9 rows beside a 153 GB index): a 4-byte row map in the
// caller's row_of[], and the neighbour's reverse strand's record read for its lfork.
__global__ void k_link_rows32(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, uint32_t *__restrict__ row_of)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const fmd_ovlp_rec_t *r = rec + i;
        if (r->status == 0 && !(r->flags & FMD_OVLP_F_OVERFLOW) && r->k[0] < n) atomicMin(row_of + r->k[0], (uint32_t)i);
    }
}
__global__ void k_link_edges32(size_t n, fmd_ovlp_rec_t *__restrict__ rec, const uint64_t *__restrict__ nei_x01, uint32_t nei_stride,
                               const uint32_t *__restrict__ row_of, fmd_ovlp_link_t *__restrict__ link, uint64_t *__restrict__ und,
                               unsigned long long *__restrict__ n_und, int force_exact)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        fmd_ovlp_rec_t *r = rec + i;
        fmd_ovlp_link_t l; l.nxt = l.rev = 0xffffffffu;
        if (r->status == 0 && r->n_nei == 1 && r->rbeg >= 0 && !(r->flags & FMD_OVLP_F_OVERFLOW)) {
            const uint64_t x0 = nei_x01[i * (size_t)nei_stride], x1 = nei_x01[i * (size_t)nei_stride + 1];
            if (x0 < n) l.nxt = row_of[x0];
            if (x1 < n) l.rev = row_of[x1];
            const bool miss = (x0 < n && l.nxt == 0xffffffffu) || (x1 < n && l.rev == 0xffffffffu);
            int d = 0;
            if (r->reserved == 2) {
                d = 1;
                if (l.rev != 0xffffffffu && !force_exact) d = lfork_decide_dev(rec[l.rev].lfork, r->rbeg);
                if (d != 1) r->reserved = (uint16_t)(d < 0 ? 1 : 0);
            }
            if (d == 1 || miss) und[atomicAdd(n_und, 1ull)] = i;
        }
        link[i] = l;
    }
}
__global__ void k_link_row_of(size_t n, const unsigned long long *__restrict__ map, uint32_t *__restrict__ row_of)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) row_of[i] = (uint32_t)(map[i] >> 32);   // (~0 stays ~0)
}
__global__ void k_link_edges(size_t n, fmd_ovlp_rec_t *__restrict__ rec, const uint64_t *__restrict__ nei_x01, uint32_t nei_stride,
                             const unsigned long long *__restrict__ map, fmd_ovlp_link_t *__restrict__ link, uint64_t *__restrict__ und,
                             unsigned long long *__restrict__ n_und, int force_exact)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        fmd_ovlp_rec_t *r = rec + i;
        fmd_ovlp_link_t l; l.nxt = l.rev = 0xffffffffu;
        if (r->status == 0 && r->n_nei == 1 && r->rbeg >= 0 && !(r->flags & FMD_OVLP_F_OVERFLOW)) {
            // (round 5 tried to spare the third random line of an edge -- the reverse strand of a neighbour that is the only read of its sequence is the row beside it,
            // and its record, read for lfork anyway, can confirm that: 3.71 -> 4.33 ms per 2*10^7 rows and 9.9 -> 12.3 GB fetched: a dependent record line costs
            // more than the look-up it spares.  Reverted; round 6 moved the lfork into the map instead.)
            const uint64_t x0 = nei_x01[i * (size_t)nei_stride], x1 = nei_x01[i * (size_t)nei_stride + 1];
            unsigned long long m1 = ~0ull;
            if (x0 < n) l.nxt = (uint32_t)(map[x0] >> 32);
            if (x1 < n) { m1 = map[x1]; l.rev = (uint32_t)(m1 >> 32); }
            // a neighbour without a row: its record is flagged (a capacity was exceeded) and the caller computes it again, larger -- the
            // edge is reported with the undecided ones, and the caller links it when that row is there (host/ovlp_table.c: table_patch_links)
            const bool miss = (x0 < n && l.nxt == 0xffffffffu) || (x1 < n && l.rev == 0xffffffffu);
            int d = 0;
            if (r->reserved == 2) {
                d = 1;
                if (l.rev != 0xffffffffu && !force_exact) d = lfork_decide_dev((uint16_t)m1, r->rbeg);
                if (d != 1) r->reserved = (uint16_t)(d < 0 ? 1 : 0);
            }
            if (d == 1 || miss) und[atomicAdd(n_und, 1ull)] = i;
        }
        link[i] = l;
    }
}
extern "C" int fmd_ovlp_link_dev(fmd_dev_t *h, void *stream_, size_t n, fmd_ovlp_rec_t *d_rec, const uint64_t *d_nei_x01, uint32_t nei_stride_u64,
                                 uint32_t *d_row_of, fmd_ovlp_link_t *d_link, uint64_t *d_undecided, uint64_t *d_n_undecided)
{
    if (!h || (n && (!d_rec || !d_nei_x01 || !d_row_of || !d_link || !d_undecided || !d_n_undecided)) || nei_stride_u64 < 2) return FMD_E_ARG;
    if (n >= 0xffffffffull) return FMD_E_ARG;                     // 32-bit row numbers
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    FMD_HIP_TRY(hipMemsetAsync(d_n_undecided, 0, 8, st));
    if (n == 0) return FMD_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    const int force_exact = getenv("FMD_CHECK_LEFT_EXACT") != nullptr;
    unsigned long long *map = getenv("FMD_LINK_MAP32") ? nullptr : (unsigned long long *)fmd_scratch_acquire(h, n * 8);   // (kept by the handle between calls; FMD_LINK_MAP32: the A/B switch)
    if (!map) {   // no room for 8 bytes per position (or asked not to): the 4-byte map in the caller's array, three look-ups per edge
        (void)hipGetLastError();
        FMD_HIP_TRY(hipMemsetAsync(d_row_of, 0xff, n * 4, st));
        k_link_rows32<<<(unsigned)blocks, 256, 0, st>>>(n, d_rec, d_row_of);
        k_link_edges32<<<(unsigned)blocks, 256, 0, st>>>(n, d_rec, d_nei_x01, nei_stride_u64, d_row_of, d_link, d_undecided, (unsigned long long *)d_n_undecided, force_exact);
        hipError_t e32 = hipGetLastError();
        if (e32 != hipSuccess) { fmd_set_hip_error(e32, "link kernels"); return FMD_E_HIP; }
        return FMD_OK;
    }
    if (hipMemsetAsync(map, 0xff, n * 8, st) != hipSuccess) { fmd_set_hip_error(hipGetLastError(), "link kernels"); fmd_scratch_release(h, map); return FMD_E_HIP; }
    k_link_rows<<<(unsigned)blocks, 256, 0, st>>>(n, d_rec, map);
    k_link_edges<<<(unsigned)blocks, 256, 0, st>>>(n, d_rec, d_nei_x01, nei_stride_u64, map, d_link, d_undecided, (unsigned long long *)d_n_undecided, force_exact);
    k_link_row_of<<<(unsigned)blocks, 256, 0, st>>>(n, map, d_row_of);
    hipError_t e = hipGetLastError();
    // the map goes back to the handle's cache when the stream has passed the kernels that read it
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    fmd_scratch_release(h, map);
    if (e != hipSuccess) { fmd_set_hip_error(e, "link kernels"); return FMD_E_HIP; }
    return FMD_OK;
}

// ------------------------------------------------------------------------------------------------
// Host form: the whole table of one shard of sequence ids, packed, pipelined.
//
// ids = first, first + step, ... (the reference's worker interleave, unitig.c:333, 398-399) or an explicit list.
// Chunks of 2^chunk_shift rows go through  fmd_ovlp_dev -> fmd_ovlp_check_left_dev -> fmd_ovlp_pack_dev  on one
// stream; the packed rows of chunk c cross PCIe on a second stream while chunk c+1 is computed (two sets of
// packed buffers, one set of everything else).  ~125 bytes per strand leave the GPU instead of 448.
__global__ void k_fill_ids(size_t n, uint64_t first, uint64_t step, uint64_t *ids)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ids[i] = first + step * i;
}

namespace {
struct DevMem { // from the handle's buffer cache: the same sizes come back call after call
    fmd_dev *h = nullptr;
    void *p = nullptr;
    int alloc(fmd_dev *h_, size_t bytes) { h = h_; p = fmd_scratch_acquire(h, bytes); return p ? FMD_OK : FMD_E_NOMEM; }
    ~DevMem() { if (p) fmd_scratch_release(h, p); }
};
struct Stream {
    hipStream_t s = nullptr;
    int make() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? FMD_OK : FMD_E_HIP; }
    ~Stream() { if (s) hipStreamDestroy(s); }
};
struct Event {
    hipEvent_t e = nullptr;
    int make() { return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? FMD_OK : FMD_E_HIP; }
    ~Event() { if (e) hipEventDestroy(e); }
};
struct Pinned { // hipHostRegister for a scope (best effort: pageable copies otherwise)
    void *p = nullptr;
    void pin(void *ptr, size_t bytes) { if (bytes >= ((size_t)8 << 20) && !getenv("FMD_NO_PIN") && !getenv("FMD_TABLE_DIR") && hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess) p = ptr; else (void)hipGetLastError(); }   // (FMD_TABLE_DIR: the table is not to be held in RAM)
    ~Pinned() { if (p) hipHostUnregister(p); }
};
}

// ---- host memory of the tables (include/fmd_hip.h: fmd_table_alloc) --------------------------------------------------------
static std::mutex g_tab_mu;
static std::unordered_map<void *, size_t> g_tab_maps;   // blocks that are file pages: address -> mapped bytes

extern "C" void *fmd_table_alloc(size_t bytes)
{
    const size_t huge = (size_t)2 << 20;
    if (bytes == 0) bytes = 1;
    const char *dir = getenv("FMD_TABLE_DIR"), *mn = getenv("FMD_TABLE_DIR_MIN");
    const size_t dir_min = mn && atoll(mn) > 0 ? (size_t)atoll(mn) : 16 * huge;
    if (dir && *dir && bytes >= dir_min) {
        const size_t page = 4096, asz = (bytes + page - 1) / page * page;
        std::string path = std::string(dir) + "/fmdtab.XXXXXX";
        const int fd = mkstemp(&path[0]);
        int err = errno;   // (of the call that failed, for the warning below)
        if (fd >= 0) {
            unlink(path.c_str());
            // the space is RESERVED (posix_fallocate), not just promised (ftruncate): a device that fills up is an allocation
            // failure here and the anonymous-memory fallback below, not a SIGBUS at first touch in the middle of the walk
            void *p = MAP_FAILED;
            err = posix_fallocate(fd, 0, (off_t)asz);
            if (err == 0) { p = mmap(nullptr, asz, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (p == MAP_FAILED) err = errno; }
            close(fd);
            if (p != MAP_FAILED) {
#ifdef MADV_RANDOM
                madvise(p, asz, MADV_RANDOM);   // the walk reads rows at random: no read-ahead
#endif
                std::lock_guard<std::mutex> lk(g_tab_mu);
                g_tab_maps[p] = asz;
                return p;
            }
        }
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "[W::%s] FMD_TABLE_DIR=%s: no file pages (%s); anonymous memory instead\n", __func__, dir, strerror(err)); }
    }
    if (bytes < 16 * huge) return malloc(bytes);
    void *p = nullptr;
    const size_t asz = (bytes + huge - 1) / huge * huge;
    if (posix_memalign(&p, huge, asz)) return nullptr;
#ifdef MADV_HUGEPAGE
    madvise(p, asz, MADV_HUGEPAGE);   // first touch of 4 KiB pages costs more than the copy that fills a 9 GB table, and random reads miss the TLB less
#endif
    return p;
}

extern "C" void fmd_table_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_tab_mu);
        auto it = g_tab_maps.find(p);
        if (it != g_tab_maps.end()) { munmap(p, it->second); g_tab_maps.erase(it); return; }
    }
    free(p);
}

extern "C" void fmd_ovlp_packed_free(uint8_t **chunks, size_t n_chunks)
{
    if (!chunks) return;
    for (size_t c = 0; c < n_chunks; ++c) { fmd_table_free(chunks[c]); chunks[c] = nullptr; }
}

// rows of all chunks kept on the device for the link pass at the end (whole-table form only)
struct KeepRows { fmd_ovlp_rec_t *rec; uint64_t *nei01; };
__global__ void k_keep_nei01(size_t n, const fmd_intv_t *__restrict__ nei, uint32_t max_nei, uint64_t *__restrict__ out)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) { out[2 * i] = nei[i * (size_t)max_nei].x[0]; out[2 * i + 1] = nei[i * (size_t)max_nei].x[1]; }
}
__global__ void k_take_reserved(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, uint8_t *__restrict__ out)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) out[i] = (uint8_t)rec[i].reserved;
}

// where the packed chunks go when the caller does not want them kept: a callback over two pinned staging sets (the streamed forms below)
struct RowSink { fmd_ovlp_rows_fn fn; void *ctx; };
struct PinnedBuf { // hipHostMalloc for a scope
    void *p = nullptr;
    int alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) == hipSuccess ? FMD_OK : FMD_E_NOMEM; }
    ~PinnedBuf() { if (p) hipHostFree(p); }
};

static int packed_batch_core(fmd_dev_t *h, const uint64_t *ids, uint64_t first, uint64_t step, size_t n, int min_match, uint32_t max_len,
                             uint32_t max_nei, int with_check_left, fmd_ovlp_rec_t *rec, uint64_t *off, uint32_t chunk_shift, uint8_t **chunks,
                             const KeepRows *keep, const RowSink *sink)
{
    if (!h || (n && !sink && (!rec || !off || !chunks)) || chunk_shift < 10 || chunk_shift > 26 || max_len == 0 || max_nei == 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    const bool timing = getenv("FMD_TIMING") != nullptr;
    auto now = [] { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    const double t_begin = now();
    double t_alloc = 0, t_wait = 0, t_host = 0, t_sink = 0;
    // computed (and packed) in chunks of CH rows; a sink takes them in pieces of SUB <= CH rows (its staging buffers are SUB rows large)
    const size_t SUB = (size_t)1 << chunk_shift;
    const size_t CH = sink && chunk_shift < 22 ? (size_t)1 << 22 : SUB, m = n < CH ? n : CH, n_chunks = (n + CH - 1) / CH;
    const uint32_t stride = 2 * ((max_len + 3) / 4 * 4);
    const size_t wb0 = fmd_ovlp_work_bytes(m, max_len, min_match), wb1 = fmd_ovlp_pack_work_bytes(m), wb = wb0 > wb1 ? wb0 : wb1;
    const size_t cap = fmd_ovlp_pack_max_bytes(m, max_nei, stride);
    if (!sink) for (size_t c = 0; c < n_chunks; ++c) chunks[c] = nullptr;
    DevMem d_ids, d_rec, d_nei, d_seq, d_work, d_prec[2], d_off[2], d_var[2];
    Stream s_cmp, s_cpy;
    Event done[2], copied[2];
    // Large tables: ALL rows from one sorted job (fmd_ovlp_sorted_dev: the order that keeps neighbours on the genome in flight together)
    // into fixed-stride arrays for the whole table, which the chunks are then packed from in id order.  Taken when those arrays + the
    // job's work area fit beside what is in HBM already; otherwise chunk by chunk in id order, as before.
    size_t sorted_batch = 0, sorted_wb = 0;
    size_t sort_min = (size_t)1 << 21;   // below this a table has no locality to find (FMD_PACKED_SORT_MIN: tests force the path on small tables)
    { const char *e = getenv("FMD_PACKED_SORT_MIN"); if (e && atoll(e) > 0) sort_min = (size_t)atoll(e); }
    if (n >= sort_min && !getenv("FMD_PACKED_UNSORTED")) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        const size_t rows = n * (8 + sizeof(fmd_ovlp_rec_t) + max_nei * sizeof(fmd_intv_t) + (size_t)stride), pk = 2 * (m * sizeof(fmd_ovlp_rec_t) + (m + 1) * 8 + cap);
        for (size_t bt = (size_t)1 << 24; bt >= ((size_t)1 << 12); bt >>= 1) {
            const size_t b2 = bt < n ? bt : n, w2 = fmd_ovlp_sorted_work_bytes(n, b2, max_len, min_match);
            // (the area also serves the per-chunk steps that follow the job: check_left of a chunk of m rows needs wb0, the pack wb1)
            const size_t w = w2 > wb ? w2 : wb;
            if (rows + pk + w + ((size_t)2 << 30) <= free_b) { sorted_batch = b2; sorted_wb = w; break; }
        }
    }
    const size_t mr = sorted_batch ? n : m;   // rows the fixed-stride arrays hold
    if (d_ids.alloc(h, mr * 8) || d_rec.alloc(h, mr * sizeof(fmd_ovlp_rec_t)) || d_nei.alloc(h, mr * max_nei * sizeof(fmd_intv_t)) || d_seq.alloc(h, mr * (size_t)stride) ||
        d_work.alloc(h, sorted_batch ? sorted_wb : wb)) return FMD_E_NOMEM;
    for (int k = 0; k < 2; ++k)
        if (d_prec[k].alloc(h, m * sizeof(fmd_ovlp_rec_t)) || d_off[k].alloc(h, (m + 1) * 8) || d_var[k].alloc(h, cap) || done[k].make() || copied[k].make()) return FMD_E_NOMEM;
    if (s_cmp.make() || s_cpy.make()) return FMD_E_HIP;
    Pinned pin_rec, pin_off;
    PinnedBuf st_rec[2], st_adj[2], st_var[2], st_off;        // the sink's staging sets (pieces of SUB rows), and the offsets of a whole chunk
    size_t st_var_cap[2] = {0, 0};
    Event sub_done[2];
    const size_t sub_m = m < SUB ? m : SUB;
    if (!sink) {
        pin_rec.pin(rec, n * sizeof(fmd_ovlp_rec_t));
        pin_off.pin(off, n * 8);
    } else {
        if (st_off.alloc((m + 1) * 8)) return FMD_E_NOMEM;
        for (int k = 0; k < 2; ++k) {
            st_var_cap[k] = sub_m * 96 + 4096;     // (grown when a piece is larger: the worst case is max_nei * 32 + seq_stride / 2 + 8 per row, the usual 60-70)
            if (st_rec[k].alloc(sub_m * sizeof(fmd_ovlp_rec_t)) || st_adj[k].alloc((sub_m + 1) * 8) || st_var[k].alloc(st_var_cap[k]) || sub_done[k].make()) return FMD_E_NOMEM;
        }
    }
    uint64_t *tot = nullptr;            // pinned landing place of the two chunk totals
    FMD_HIP_TRY(hipHostMalloc((void **)&tot, 2 * sizeof(uint64_t), hipHostMallocDefault));
    t_alloc = now() - t_begin;
    int rc = FMD_OK;
    std::vector<void *> registered;
    auto fail = [&](int code) { rc = code; };
    double t_job = 0;
    if (sorted_batch) {   // every row of the table, once
        const double t0 = now();
        if (ids) { if (hipMemcpyAsync(d_ids.p, ids, n * 8, hipMemcpyHostToDevice, s_cmp.s) != hipSuccess) fail(FMD_E_HIP); }
        else k_fill_ids<<<1024, 256, 0, s_cmp.s>>>(n, first, step, (uint64_t *)d_ids.p);
        if (rc == FMD_OK && (hipMemsetAsync(d_seq.p, 0, n * (size_t)stride, s_cmp.s) != hipSuccess || hipMemsetAsync(d_nei.p, 0, n * max_nei * sizeof(fmd_intv_t), s_cmp.s) != hipSuccess)) fail(FMD_E_HIP);
        if (rc == FMD_OK)
            rc = fmd_ovlp_sorted_dev(h, s_cmp.s, n, (uint64_t *)d_ids.p, min_match, max_len, max_nei, (fmd_ovlp_rec_t *)d_rec.p, (fmd_intv_t *)d_nei.p, (uint8_t *)d_seq.p, stride,
                                     d_work.p, sorted_wb, sorted_batch);
        if (rc == FMD_OK && timing) { hipStreamSynchronize(s_cmp.s); t_job = now() - t0; }
    }
    // chunk c: compute on s_cmp into set c & 1; its copy-out is issued one iteration later
    for (size_t c = 0; c <= n_chunks && rc == FMD_OK; ++c) {
        if (c < n_chunks) {
            const int k = (int)(c & 1);
            const size_t b = c * CH, nc = n - b < CH ? n - b : CH;
            const size_t row0 = sorted_batch ? b : 0;   // the chunk's first row in the fixed-stride arrays
            fmd_ovlp_rec_t *c_rec = (fmd_ovlp_rec_t *)d_rec.p + row0;
            fmd_intv_t *c_nei = (fmd_intv_t *)d_nei.p + row0 * max_nei;
            uint8_t *c_seq = (uint8_t *)d_seq.p + row0 * (size_t)stride;
            if (c >= 2 && hipStreamWaitEvent(s_cmp.s, copied[k].e, 0) != hipSuccess) { fail(FMD_E_HIP); break; } // set k has left the GPU
            if (!sorted_batch) {
                if (ids) { if (hipMemcpyAsync(d_ids.p, ids + b, nc * 8, hipMemcpyHostToDevice, s_cmp.s) != hipSuccess) { fail(FMD_E_HIP); break; } }
                else k_fill_ids<<<1024, 256, 0, s_cmp.s>>>(nc, first + step * b, step, (uint64_t *)d_ids.p);
                if (hipMemsetAsync(d_seq.p, 0, nc * (size_t)stride, s_cmp.s) != hipSuccess || hipMemsetAsync(d_nei.p, 0, nc * max_nei * sizeof(fmd_intv_t), s_cmp.s) != hipSuccess) { fail(FMD_E_HIP); break; }
                rc = fmd_ovlp_dev(h, s_cmp.s, nc, (uint64_t *)d_ids.p, min_match, max_len, max_nei, c_rec, c_nei, c_seq, stride, d_work.p, wb);
            }
            if (rc == FMD_OK && with_check_left)
                rc = fmd_ovlp_check_left_dev(h, s_cmp.s, nc, min_match, max_len, c_rec, c_seq, stride, d_work.p, sorted_batch ? sorted_wb : wb);
            if (rc == FMD_OK)
                rc = fmd_ovlp_pack_dev(h, s_cmp.s, nc, c_rec, c_nei, max_nei, c_seq, stride,
                                       (fmd_ovlp_rec_t *)d_prec[k].p, (uint64_t *)d_off[k].p, (uint8_t *)d_var[k].p, cap, d_work.p, sorted_batch ? sorted_wb : wb);
            if (rc != FMD_OK) break;
            if (keep) { // the fixed-stride record and the first neighbour's coordinates stay on the device for the link pass
                if (hipMemcpyAsync(keep->rec + b, c_rec, nc * sizeof(fmd_ovlp_rec_t), hipMemcpyDeviceToDevice, s_cmp.s) != hipSuccess) { fail(FMD_E_HIP); break; }
                k_keep_nei01<<<1024, 256, 0, s_cmp.s>>>(nc, (const fmd_intv_t *)c_nei, max_nei, keep->nei01 + 2 * b);
            }
            if (hipMemcpyAsync(tot + k, (uint64_t *)d_off[k].p + nc, 8, hipMemcpyDeviceToHost, s_cmp.s) != hipSuccess || hipEventRecord(done[k].e, s_cmp.s) != hipSuccess) { fail(FMD_E_HIP); break; }
        }
        if (sink && c >= 1) { // chunk c - 1 to the caller, piece by piece, while chunk c is computed: piece j crosses PCIe while the caller has piece j - 1
            const size_t p = c - 1, b = p * CH, np = n - b < CH ? n - b : CH;
            const int k = (int)(p & 1);
            double t0 = now();
            if (hipEventSynchronize(done[k].e) != hipSuccess) { fail(FMD_E_HIP); break; }
            t_wait += now() - t0; t0 = now();
            if (tot[k] > cap) { fail(FMD_E_OVERFLOW); break; }
            if (hipMemcpyAsync(st_off.p, d_off[k].p, (np + 1) * 8, hipMemcpyDeviceToHost, s_cpy.s) != hipSuccess || hipStreamSynchronize(s_cpy.s) != hipSuccess) { fail(FMD_E_HIP); break; }
            const uint64_t *offs = (const uint64_t *)st_off.p;
            const size_t n_sub = (np + SUB - 1) / SUB;
            t_host += now() - t0;
            for (size_t j = 0; j <= n_sub && rc == FMD_OK; ++j) {
                if (j < n_sub) {
                    const size_t r0 = j * SUB, nr = np - r0 < SUB ? np - r0 : SUB;
                    const int ks = (int)(j & 1);
                    const uint64_t b0 = offs[r0], bytes = offs[r0 + nr] - b0;
                    t0 = now();
                    if (bytes > st_var_cap[ks]) {   // (the caller finished with this set two pieces ago)
                        hipHostFree(st_var[ks].p); st_var[ks].p = nullptr;
                        st_var_cap[ks] = bytes + bytes / 4;
                        if (st_var[ks].alloc(st_var_cap[ks])) { fail(FMD_E_NOMEM); break; }
                    }
                    if (hipMemcpyAsync(st_rec[ks].p, (const fmd_ovlp_rec_t *)d_prec[k].p + r0, nr * sizeof(fmd_ovlp_rec_t), hipMemcpyDeviceToHost, s_cpy.s) != hipSuccess ||
                        (bytes && hipMemcpyAsync(st_var[ks].p, (const uint8_t *)d_var[k].p + b0, bytes, hipMemcpyDeviceToHost, s_cpy.s) != hipSuccess) ||
                        hipEventRecord(sub_done[ks].e, s_cpy.s) != hipSuccess) { fail(FMD_E_HIP); break; }
                    uint64_t *adj = (uint64_t *)st_adj[ks].p;
                    for (size_t i = 0; i <= nr; ++i) adj[i] = offs[r0 + i] - b0;
                    t_host += now() - t0;
                }
                if (j >= 1) {
                    const size_t q = j - 1, r0 = q * SUB, nr = np - r0 < SUB ? np - r0 : SUB;
                    const int ks = (int)(q & 1);
                    t0 = now();
                    if (hipEventSynchronize(sub_done[ks].e) != hipSuccess) { fail(FMD_E_HIP); break; }
                    t_wait += now() - t0; t0 = now();
                    const int src = sink->fn(sink->ctx, (uint64_t)(b + r0), nr, (const fmd_ovlp_rec_t *)st_rec[ks].p, (const uint64_t *)st_adj[ks].p, (const uint8_t *)st_var[ks].p, offs[r0 + nr] - offs[r0]);
                    t_sink += now() - t0;
                    if (src) { fail(src); break; }
                }
            }
            if (rc == FMD_OK && hipEventRecord(copied[k].e, s_cpy.s) != hipSuccess) { fail(FMD_E_HIP); break; }
        }
        if (!sink && c >= 1) { // copy chunk c - 1 out while chunk c runs
            const size_t p = c - 1, b = p * CH, np = n - b < CH ? n - b : CH;
            const int k = (int)(p & 1);
            double t0 = now();
            if (hipEventSynchronize(done[k].e) != hipSuccess) { fail(FMD_E_HIP); break; }
            t_wait += now() - t0; t0 = now();
            const uint64_t bytes = tot[k];
            if (bytes > cap) { fail(FMD_E_OVERFLOW); break; }  // cannot happen: cap is the worst case
            // huge pages on request, or file pages under FMD_TABLE_DIR: the walk reads these rows at random
            void *buf = fmd_table_alloc(bytes ? bytes : 64);
            if (!buf) { fail(FMD_E_NOMEM); break; }
            chunks[p] = (uint8_t *)buf;
            // (file pages are not pinned: registering every chunk until the end of the pass would hold the whole table in RAM, which is what FMD_TABLE_DIR is there to avoid)
            if (bytes >= ((size_t)8 << 20) && !getenv("FMD_NO_PIN") && !getenv("FMD_TABLE_DIR") && hipHostRegister(buf, bytes, hipHostRegisterDefault) == hipSuccess) registered.push_back(buf);
            else (void)hipGetLastError();
            if (hipMemcpyAsync(rec + b, d_prec[k].p, np * sizeof(fmd_ovlp_rec_t), hipMemcpyDeviceToHost, s_cpy.s) != hipSuccess ||
                hipMemcpyAsync(off + b, d_off[k].p, np * 8, hipMemcpyDeviceToHost, s_cpy.s) != hipSuccess ||
                (bytes && hipMemcpyAsync(buf, d_var[k].p, bytes, hipMemcpyDeviceToHost, s_cpy.s) != hipSuccess) ||
                hipEventRecord(copied[k].e, s_cpy.s) != hipSuccess) { fail(FMD_E_HIP); break; }
            t_host += now() - t0;
        }
    }
    hipStreamSynchronize(s_cmp.s);
    const double t_tail0 = now();
    hipStreamSynchronize(s_cpy.s);
    if (timing) fprintf(stderr, "[M::%s] %zu rows in %zu chunks%s: device buffers + pinning %.3f s, sorted job %.3f s, waiting for kernels / copies %.3f s, host allocation + copy issue %.3f s, the caller's sink %.3f s, last copy %.3f s, total %.3f s\n",
                        __func__, n, n_chunks, sorted_batch ? " (all rows from one sorted job)" : " (chunk by chunk in id order)", t_alloc, t_job, t_wait, t_host, t_sink, now() - t_tail0, now() - t_begin);
    for (void *q : registered) hipHostUnregister(q);
    hipHostFree(tot);
    if (rc == FMD_OK) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { fmd_set_hip_error(e, "packed overlap batch"); rc = FMD_E_HIP; }
    } else if (rc == FMD_E_HIP) fmd_set_hip_error(hipGetLastError(), "packed overlap batch");
    if (rc != FMD_OK && !sink) fmd_ovlp_packed_free(chunks, n_chunks);
    return rc;
}

extern "C" int fmd_ovlp_packed_batch(fmd_dev_t *h, const uint64_t *ids, uint64_t first, uint64_t step, size_t n, int min_match, uint32_t max_len,
                                     uint32_t max_nei, int with_check_left, fmd_ovlp_rec_t *rec, uint64_t *off, uint32_t chunk_shift, uint8_t **chunks)
{
    return packed_batch_core(h, ids, first, step, n, min_match, max_len, max_nei, with_check_left, rec, off, chunk_shift, chunks, nullptr, nullptr);
}

// The whole table (ids 0 .. n-1) on ONE device: the packed rows as above, then the link pass on the device
// (fmd_ovlp_link_dev) -- row_of[n], link[n], rec[i].reserved set to check_left_simple's verdict wherever lfork decides it,
// and the ids it leaves open in *undecided (malloc'ed, fmd_host_free; *n_undecided of them, ascending).
extern "C" int fmd_ovlp_packed_table(fmd_dev_t *h, size_t n, int min_match, uint32_t max_len, uint32_t max_nei, fmd_ovlp_rec_t *rec, uint64_t *off,
                                     uint32_t chunk_shift, uint8_t **chunks, uint32_t *row_of, fmd_ovlp_link_t *link, uint64_t **undecided, uint64_t *n_undecided)
{
    if (!h || !undecided || !n_undecided || (n && (!rec || !off || !chunks || !row_of || !link))) return FMD_E_ARG;
    *undecided = nullptr; *n_undecided = 0;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffffffull) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    DevMem d_rec_all, d_nei01, d_row_of, d_link, d_und, d_nund, d_res;
    if (d_rec_all.alloc(h, n * sizeof(fmd_ovlp_rec_t)) || d_nei01.alloc(h, n * 16) || d_row_of.alloc(h, n * 4) || d_link.alloc(h, n * sizeof(fmd_ovlp_link_t)) ||
        d_und.alloc(h, n * 8) || d_nund.alloc(h, 8) || d_res.alloc(h, n)) return FMD_E_NOMEM;
    KeepRows keep{(fmd_ovlp_rec_t *)d_rec_all.p, (uint64_t *)d_nei01.p};
    int rc = packed_batch_core(h, nullptr, 0, 1, n, min_match, max_len, max_nei, 0, rec, off, chunk_shift, chunks, &keep, nullptr);
    if (rc) return rc;
    const size_t n_chunks = (n + ((size_t)1 << chunk_shift) - 1) >> chunk_shift;
    rc = fmd_ovlp_link_dev(h, nullptr, n, keep.rec, keep.nei01, 2, (uint32_t *)d_row_of.p, (fmd_ovlp_link_t *)d_link.p, (uint64_t *)d_und.p, (uint64_t *)d_nund.p);
    if (rc == FMD_OK) {
        size_t blocks = (n + 255) / 256;
        if (blocks > (1u << 20)) blocks = 1u << 20;
        k_take_reserved<<<(unsigned)blocks, 256>>>(n, keep.rec, (uint8_t *)d_res.p);
        std::vector<uint8_t> res(n);
        uint64_t nu = 0;
        if (hipMemcpy(row_of, d_row_of.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(link, d_link.p, n * sizeof(fmd_ovlp_link_t), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(res.data(), d_res.p, n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&nu, d_nund.p, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FMD_E_HIP;
        if (rc == FMD_OK) {
            for (size_t i = 0; i < n; ++i) rec[i].reserved = res[i];
            if (nu) {
                uint64_t *u = (uint64_t *)malloc(nu * 8);
                if (!u) rc = FMD_E_NOMEM;
                else if (hipMemcpy(u, d_und.p, nu * 8, hipMemcpyDeviceToHost) != hipSuccess) { free(u); rc = FMD_E_HIP; }
                else { std::sort(u, u + nu); *undecided = u; *n_undecided = nu; }
            }
        }
    }
    if (rc != FMD_OK) { fmd_ovlp_packed_free(chunks, n_chunks); if (rc == FMD_E_HIP) fmd_set_hip_error(hipGetLastError(), "packed table: link pass"); }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Streamed forms: the packed chunks are handed to a callback from two pinned staging sets and nothing of the table stays in host
// memory unless the callback keeps it (host/slim_table.c keeps ~58 bytes per row of the ~125 that cross PCIe).
extern "C" int fmd_ovlp_packed_stream(fmd_dev_t *h, const uint64_t *ids, uint64_t first, uint64_t step, size_t n, int min_match, uint32_t max_len,
                                      uint32_t max_nei, int with_check_left, uint32_t chunk_shift, fmd_ovlp_rows_fn fn, void *ctx)
{
    if (!fn) return FMD_E_ARG;
    RowSink sink{fn, ctx};
    return packed_batch_core(h, ids, first, step, n, min_match, max_len, max_nei, with_check_left, nullptr, nullptr, chunk_shift, nullptr, nullptr, &sink);
}

// The whole table (ids 0 .. n-1) on ONE device as a job in three steps: rows (streamed; the fixed-stride records and the first neighbour's
// coordinates stay on the device), patch (rows the caller computed again: their records replace the flagged ones), link (fmd_ovlp_link_dev
// over the records as they are THEN, its results streamed in pieces).
struct fmd_ovlp_tabjob {
    fmd_dev *h = nullptr;
    size_t n = 0;
    DevMem rec_all, nei01;
};
extern "C" void fmd_ovlp_tabjob_free(fmd_ovlp_tabjob_t *j) { delete j; }
extern "C" int fmd_ovlp_tabjob_rows(fmd_dev_t *h, size_t n, int min_match, uint32_t max_len, uint32_t max_nei, uint32_t chunk_shift,
                                    fmd_ovlp_rows_fn fn, void *ctx, fmd_ovlp_tabjob_t **job)
{
    if (!h || !fn || !job) return FMD_E_ARG;
    *job = nullptr;
    if (n >= 0xffffffffull) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    fmd_ovlp_tabjob *j = new (std::nothrow) fmd_ovlp_tabjob;
    if (!j) return FMD_E_NOMEM;
    j->h = h; j->n = n;
    if (j->rec_all.alloc(h, (n ? n : 1) * sizeof(fmd_ovlp_rec_t)) || j->nei01.alloc(h, (n ? n : 1) * 16)) { delete j; return FMD_E_NOMEM; }
    KeepRows keep{(fmd_ovlp_rec_t *)j->rec_all.p, (uint64_t *)j->nei01.p};
    RowSink sink{fn, ctx};
    const int rc = packed_batch_core(h, nullptr, 0, 1, n, min_match, max_len, max_nei, 0, nullptr, nullptr, chunk_shift, nullptr, &keep, &sink);
    if (rc) { delete j; return rc; }
    *job = j;
    return FMD_OK;
}
__global__ void k_patch_rows(size_t m, const uint64_t *__restrict__ ids, const fmd_ovlp_rec_t *__restrict__ rec, const uint64_t *__restrict__ nei01, size_t n,
                             fmd_ovlp_rec_t *__restrict__ rec_all, uint64_t *__restrict__ nei01_all)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += step) {
        const uint64_t id = ids[i];
        if (id >= n) continue;
        rec_all[id] = rec[i];
        nei01_all[2 * id] = nei01[2 * i]; nei01_all[2 * id + 1] = nei01[2 * i + 1];
    }
}
extern "C" int fmd_ovlp_tabjob_patch(fmd_ovlp_tabjob_t *j, size_t m, const uint64_t *ids, const fmd_ovlp_rec_t *rec, const uint64_t *nei01)
{
    if (!j || (m && (!ids || !rec || !nei01))) return FMD_E_ARG;
    if (m == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(j->h->device));
    DevMem d_ids, d_rec, d_n01;
    if (d_ids.alloc(j->h, m * 8) || d_rec.alloc(j->h, m * sizeof(fmd_ovlp_rec_t)) || d_n01.alloc(j->h, m * 16)) return FMD_E_NOMEM;
    FMD_HIP_TRY(hipMemcpy(d_ids.p, ids, m * 8, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(d_rec.p, rec, m * sizeof(fmd_ovlp_rec_t), hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(d_n01.p, nei01, m * 16, hipMemcpyHostToDevice));
    size_t blocks = (m + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    k_patch_rows<<<(unsigned)blocks, 256>>>(m, (const uint64_t *)d_ids.p, (const fmd_ovlp_rec_t *)d_rec.p, (const uint64_t *)d_n01.p, j->n, (fmd_ovlp_rec_t *)j->rec_all.p, (uint64_t *)j->nei01.p);
    FMD_HIP_TRY(hipDeviceSynchronize());
    return FMD_OK;
}
extern "C" int fmd_ovlp_tabjob_link(fmd_ovlp_tabjob_t *j, fmd_ovlp_links_fn fn, void *ctx, uint64_t **undecided, uint64_t *n_undecided)
{
    if (!j || !fn || !undecided || !n_undecided) return FMD_E_ARG;
    *undecided = nullptr; *n_undecided = 0;
    const size_t n = j->n;
    if (n == 0) return FMD_OK;
    fmd_dev *h = j->h;
    FMD_HIP_TRY(hipSetDevice(h->device));
    DevMem d_row_of, d_link, d_und, d_nund, d_res;
    if (d_row_of.alloc(h, n * 4) || d_link.alloc(h, n * sizeof(fmd_ovlp_link_t)) || d_und.alloc(h, n * 8) || d_nund.alloc(h, 8) || d_res.alloc(h, n)) return FMD_E_NOMEM;
    int rc = fmd_ovlp_link_dev(h, nullptr, n, (fmd_ovlp_rec_t *)j->rec_all.p, (const uint64_t *)j->nei01.p, 2, (uint32_t *)d_row_of.p, (fmd_ovlp_link_t *)d_link.p,
                               (uint64_t *)d_und.p, (uint64_t *)d_nund.p);
    if (rc) return rc;
    size_t blocks = (n + 255) / 256;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    k_take_reserved<<<(unsigned)blocks, 256>>>(n, (const fmd_ovlp_rec_t *)j->rec_all.p, (uint8_t *)d_res.p);
    const size_t piece = (size_t)1 << 22;
    PinnedBuf st_link, st_res;
    if (st_link.alloc(piece * sizeof(fmd_ovlp_link_t)) || st_res.alloc(piece)) return FMD_E_NOMEM;
    for (size_t b = 0; b < n && rc == FMD_OK; b += piece) {
        const size_t m = n - b < piece ? n - b : piece;
        if (hipMemcpy(st_link.p, (const fmd_ovlp_link_t *)d_link.p + b, m * sizeof(fmd_ovlp_link_t), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(st_res.p, (const uint8_t *)d_res.p + b, m, hipMemcpyDeviceToHost) != hipSuccess) { rc = FMD_E_HIP; break; }
        rc = fn(ctx, (uint64_t)b, m, (const fmd_ovlp_link_t *)st_link.p, (const uint8_t *)st_res.p);
    }
    uint64_t nu = 0;
    if (rc == FMD_OK && hipMemcpy(&nu, d_nund.p, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FMD_E_HIP;
    if (rc == FMD_OK && nu) {
        uint64_t *u = (uint64_t *)malloc(nu * 8);
        if (!u) rc = FMD_E_NOMEM;
        else if (hipMemcpy(u, d_und.p, nu * 8, hipMemcpyDeviceToHost) != hipSuccess) { free(u); rc = FMD_E_HIP; }
        else { std::sort(u, u + nu); *undecided = u; *n_undecided = nu; }
    }
    if (rc == FMD_E_HIP) fmd_set_hip_error(hipGetLastError(), "table job: link pass");
    return rc;
}
