// fmd_build.hip -- FMD-index construction on the GPU (the `fermi build` BWT, cmd.c:378-484 +
// build.c:11-50, as plumbing for benchmarks and as a drop-in for index construction of short
// reads).  The indexed text is, per read in input order,  read $ revcomp(read) $  and sentinels
// compare by sequence id (ksa.c:54), so the BWT is the sort of all suffixes "sequence tail up to
// and including its own $", ties broken by sequence id.  Short reads make that a fixed number
// of stable LSD radix passes over 21-symbol (63-bit) key chunks: ceil((maxlen+1)/21) passes of
// rocPRIM's radix sort over (key, text position) pairs -- HBM-streaming work the MI355X does
// in about a second for 2e9 suffixes, instead of SA-IS + merging on the host.
#include "fmd_prim.h"
#include <stdlib.h>
#include <vector>
#include <type_traits>
#include "fmd_internal.h"

// text[T_s ..] for sequence s = 2r (forward) / 2r+1 (reverse complement); one thread per read base
__global__ void k_build_text(size_t n_reads, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ off,
                             uint8_t *__restrict__ text)
{
    // one 64-thread group per read (grid-stride: the grid is capped)
    for (size_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint64_t o = off[r], len = off[r + 1] - o;
        const uint64_t t0 = 2 * (o + r); // both strands of all earlier reads, each with its '$'
        for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) {
            const uint8_t c = reads[o + i];
            text[t0 + i] = c;
            text[t0 + len + 1 + (len - 1 - i)] = (c >= 1 && c <= 4) ? (uint8_t)(5 - c) : c;
        }
        if (threadIdx.x == 0) { text[t0 + len] = 0; text[t0 + 2 * len + 1] = 0; }
    }
}

// distance from text position t to the '$' closing its sequence
struct RemUniform { uint32_t len1; __device__ uint32_t operator()(uint64_t t) const { return len1 - 1 - (uint32_t)(t % len1); } };
struct RemRagged {   // binary search over sequence ends (position of each '$'), n_seq entries
    const uint64_t *send; uint64_t n_seq;
    __device__ uint32_t operator()(uint64_t t) const {
        uint64_t lo = 0, hi = n_seq - 1;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (send[mid] < t) lo = mid + 1; else hi = mid; }
        return (uint32_t)(send[lo] - t);
    }
};

// ---- the 63-bit key of a chunk: up to 21 symbols from text position a, 3 bits each, the first one highest, zeros behind the `mm` that count.
// Byte by byte that is 21 loads per suffix, and the key kernels took twice the time of the radix sorts they feed (9.6 s of the two 5*10^7-read
// builds of a bench run against 5.2 s: profiles/r4_final2/kernel_stats.csv) although a suffix lies in one or two 64-byte lines.  A byte text is
// read as four aligned 8-byte words instead -- the 21 symbols start anywhere in the first -- and eight symbols at a time go from bytes to 3-bit
// fields in registers.  (`n` = symbols of the text: a window that would reach past it takes the byte loop, as does a text that is not 8-byte aligned.)
#include "fmd_keys.inc"

template <class Rem>
__global__ void k_chunk_keys(const uint8_t *__restrict__ text, uint64_t n, uint64_t n_wide, const uint32_t *__restrict__ order, int chunk,
                             Rem rem, uint64_t *__restrict__ keys)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t t = order ? order[i] : i;
        const uint32_t r = rem(t), o0 = 21u * (uint32_t)chunk;
        uint64_t key = 0;
        if (o0 <= r) {
            const uint32_t m = r - o0 + 1 < 21 ? r - o0 + 1 : 21; // symbols up to and including the '$'
            key = chunk_key(text, n_wide, t + o0, m);
        }
        keys[i] = key;
    }
}

__global__ void k_iota32(uint32_t *a, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a[i] = (uint32_t)i;
}

__global__ void k_emit_bwt(const uint8_t *__restrict__ text, const uint32_t *__restrict__ order, uint64_t n, uint8_t *__restrict__ bwt)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t t = order[i];
        bwt[i] = t ? text[t - 1] : 0; // text[t-1] is '$' (=0) exactly when t starts a sequence
    }
}

__global__ void k_seq_ends(size_t n_reads, const uint64_t *__restrict__ off, uint64_t *__restrict__ send)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = off[r], len = off[r + 1] - o, t0 = 2 * (o + r);
        send[2 * r] = t0 + len; send[2 * r + 1] = t0 + 2 * len + 1;
    }
}

// blocks for n items, t threads each: at most 2^31 threads per launch (the dispatch packet counts work-items in 32
// bits; the kernels above loop with a grid stride)
static inline unsigned nblk(uint64_t n, unsigned t)
{
    const uint64_t b = (n + t - 1) / t, cap = (1ull << 31) / t;
    return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

#include <time.h>
static double bt_now(hipStream_t st, bool on) { if (!on) return 0; hipStreamSynchronize(st); struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
struct DevPtr { void *p = nullptr; ~DevPtr() { if (p) hipFree(p); } };
#define DALLOC(buf, bytes) do { hipError_t e__ = hipMalloc(&(buf).p, (bytes) ? (bytes) : 16); \
    if (e__ != hipSuccess) { fmd_set_hip_error(e__, "hipMalloc(" #buf ")"); return FMD_E_NOMEM; } } while (0)


// ------------------------------------------------------------------ >= 2^32 symbols: bucketed
// Same ordering, sorted one PREFIX bucket at a time with 64-bit text positions.  A bucket = the suffixes that share
// their first `depth` symbols (a '$' ends the comparison: what follows it in the text belongs to the next sequence and
// counts as zeros, exactly as in the chunk keys); buckets taken in increasing prefix order are contiguous in the BWT.
// The suffixes starting with '$' are already in order (sequence id = text order); so is every bucket whose prefix holds
// a '$' (all its suffixes are equal up to their '$': ties go by sequence id).  The others take the LSD passes inside
// their bucket only.  Peak memory = text + BWT + 32 bytes per suffix of the largest bucket (+ the sort's scratch):
// depth 1 carries 50 M x 100 bp reads, depth 2-3 the 2.5*10^8 x 100 bp (5*10^10 symbols) that one 288 GB GPU can hold
// next to its own text and BWT.
// the text as the bucketed builder sees it: one byte per symbol, or two symbols per byte (the in-place index builder)
struct Text8 { const uint8_t *p; __device__ __forceinline__ uint32_t operator[](uint64_t t) const { return p[t]; } };
struct Text4 { const uint8_t *p; __device__ __forceinline__ uint32_t operator[](uint64_t t) const { return (p[t >> 1] >> (4 * (uint32_t)(t & 1))) & 15u; } };

template <class Text>
__device__ __forceinline__ uint32_t prefix_code(const Text &text, uint64_t n, int depth, uint64_t t)   // 3 bits per symbol, first symbol on top
{
    uint32_t c = 0;
    for (int j = 0; j < depth; ++j) {
        const uint32_t x = t + (uint64_t)j < n ? text[t + j] : 0u;
        c = c << 3 | x;
        if (x == 0) { c <<= 3 * (depth - 1 - j); break; }
    }
    return c;
}
// sizes of all prefix buckets in ONE pass over the text (a selection pass per bucket just to count cost as much as the
// selection itself: 0.65 s per pass over 5*10^10 symbols, 156 buckets at depth 3)
template <class Text>
__global__ void k_prefix_hist(Text text, uint64_t n, int depth, unsigned long long *__restrict__ hist)
{
    extern __shared__ unsigned int h_lds[];
    const uint32_t bins = 1u << (3 * depth);
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x) h_lds[i] = 0;
    __syncthreads();
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&h_lds[prefix_code(text, n, depth, t)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x) if (h_lds[i]) atomicAdd(&hist[i], (unsigned long long)h_lds[i]);
}

// The positions of one bucket, ascending.  a device-wide select over a counting iterator takes 0.65 s per pass over 5*10^10
// symbols (156 buckets at depth 3, ~800 at depth 4); here a wave owns a tile of 2^16 consecutive positions: it counts its
// matches, an exclusive scan over the tiles gives every tile its place, and the wave writes its matches in order with
// ballot prefixes -- two coalesced sweeps over the text per bucket.
#define SEL_TILE_SHIFT 16
template <class Text>
__global__ void k_tile_count(Text text, uint64_t n, int depth, uint32_t code, uint64_t n_tiles, uint64_t *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    for (uint64_t tile = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += (uint64_t)gridDim.x * (blockDim.x >> 6)) {
        const uint64_t t0 = tile << SEL_TILE_SHIFT, t1 = t0 + (1ull << SEL_TILE_SHIFT) < n ? t0 + (1ull << SEL_TILE_SHIFT) : n;
        uint32_t c = 0;
        for (uint64_t t = t0 + (uint64_t)lane; t < t1; t += 64) c += prefix_code(text, n, depth, t) == code;
        for (int o = 32; o; o >>= 1) c += __shfl_xor((int)c, o);
        if (lane == 0) counts[tile] = c;
    }
}
template <class Text>
__global__ void k_tile_select(Text text, uint64_t n, int depth, uint32_t code, uint64_t n_tiles, const uint64_t *__restrict__ offset, uint64_t *__restrict__ ids)
{
    const int lane = threadIdx.x & 63;
    for (uint64_t tile = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += (uint64_t)gridDim.x * (blockDim.x >> 6)) {
        const uint64_t t0 = tile << SEL_TILE_SHIFT, t1 = t0 + (1ull << SEL_TILE_SHIFT) < n ? t0 + (1ull << SEL_TILE_SHIFT) : n;
        uint64_t base = offset[tile];
        for (uint64_t tb = t0; tb < t1; tb += 64) {           // all 64 lanes take every turn: the ballot needs them
            const uint64_t t = tb + (uint64_t)lane;
            const bool hit = t < t1 && prefix_code(text, n, depth, t) == code;
            const uint64_t m = __ballot(hit);
            if (hit) ids[base + (uint64_t)__popcll(m & ((1ull << lane) - 1))] = t;
            base += (uint64_t)__popcll(m);
        }
    }
}

// The same two sweeps over the 4-bit text of the in-place builder, sixteen positions per 64-bit word.  Symbol by symbol (above) a position costs ~25
// instructions, and config 5 asks 341 buckets x 2 sweeps x 1.4*10^11 positions of them: ~285 of the 350 s its build took.  A prefix of `depth` <= 4
// symbols is `depth` nibbles of the text: nibble j of the word shifted down by j symbols against symbol j of the code, all sixteen positions at once
// (a nibble is zero where the exclusive-or with the replicated symbol is); a code with a '$' in it compares up to that '$' only, as prefix_code
// does.  A lane owns one word of a turn (64 words = 1024 positions per wave instruction, 512 bytes coalesced) and the first symbols of the next;
// the tile at the end of the text, where "the next" may lie outside it, goes symbol by symbol.
struct Match4 { uint64_t pat[4]; int de; };
static Match4 match4_of(uint32_t code, int depth)
{
    Match4 m; m.de = depth;
    for (int j = 0; j < 4; ++j) m.pat[j] = 0;
    for (int j = 0; j < depth; ++j) {
        const uint64_t sy = (code >> (3 * (depth - 1 - j))) & 7u;
        m.pat[j] = sy * 0x1111111111111111ull;
        if (sy == 0) { m.de = j + 1; break; }
    }
    return m;
}
__device__ __forceinline__ uint64_t match16(uint64_t w, uint64_t nx, const Match4 &m)   // bit 4i = position i of the word starts the prefix
{
    const uint64_t ones = 0x1111111111111111ull;
    uint64_t acc = ones;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < m.de) {
            const uint64_t y = j ? (w >> (4 * j)) | (nx << (64 - 4 * j)) : w, z = y ^ m.pat[j];
            acc &= ~(z | z >> 1 | z >> 2 | z >> 3);
        }
    }
    return acc & ones;
}
#define SEL_TILE_WORDS (1u << (SEL_TILE_SHIFT - 4))
__global__ void k_tile_count4(Text4 text, uint64_t n, int depth, uint32_t code, Match4 m4, uint64_t n_tiles, uint64_t *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const uint64_t *__restrict__ p64 = (const uint64_t *)text.p;
    for (uint64_t tile = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += (uint64_t)gridDim.x * (blockDim.x >> 6)) {
        const uint64_t t0 = tile << SEL_TILE_SHIFT, t1 = t0 + (1ull << SEL_TILE_SHIFT) < n ? t0 + (1ull << SEL_TILE_SHIFT) : n;
        uint32_t c = 0;
        if (t0 + (1ull << SEL_TILE_SHIFT) + 4 <= n) {
            const uint64_t w0 = t0 >> 4;
            for (uint32_t it = 0; it < SEL_TILE_WORDS; it += 64) { const uint64_t wi = w0 + it + (uint64_t)lane; c += (uint32_t)__popcll(match16(p64[wi], p64[wi + 1], m4)); }
        } else for (uint64_t t = t0 + (uint64_t)lane; t < t1; t += 64) c += prefix_code(text, n, depth, t) == code;
        for (int o = 32; o; o >>= 1) c += __shfl_xor((int)c, o);
        if (lane == 0) counts[tile] = c;
    }
}
__global__ void k_tile_select4(Text4 text, uint64_t n, int depth, uint32_t code, Match4 m4, uint64_t n_tiles, const uint64_t *__restrict__ offset, uint64_t *__restrict__ ids)
{
    const int lane = threadIdx.x & 63;
    const uint64_t *__restrict__ p64 = (const uint64_t *)text.p;
    for (uint64_t tile = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += (uint64_t)gridDim.x * (blockDim.x >> 6)) {
        const uint64_t t0 = tile << SEL_TILE_SHIFT, t1 = t0 + (1ull << SEL_TILE_SHIFT) < n ? t0 + (1ull << SEL_TILE_SHIFT) : n;
        uint64_t base = offset[tile];
        if (t0 + (1ull << SEL_TILE_SHIFT) + 4 <= n) {
            const uint64_t w0 = t0 >> 4;
            for (uint32_t it = 0; it < SEL_TILE_WORDS; it += 64) {
                const uint64_t wi = w0 + it + (uint64_t)lane;
                uint64_t acc = match16(p64[wi], p64[wi + 1], m4);
                const uint32_t cnt = (uint32_t)__popcll(acc);
                if (__ballot(cnt != 0) == 0) continue;
                uint32_t inc = cnt;                                 // matches of the lanes up to this one: the lanes' words are in text order
                for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += v; }
                uint64_t at = base + inc - cnt;
                while (acc) { const int b = __ffsll((long long)acc) - 1; ids[at++] = (wi << 4) + (uint64_t)(b >> 2); acc &= acc - 1; }
                base += (uint64_t)(uint32_t)__shfl((int)inc, 63);
            }
        } else {
            for (uint64_t tb = t0; tb < t1; tb += 64) {
                const uint64_t t = tb + (uint64_t)lane;
                const bool hit = t < t1 && prefix_code(text, n, depth, t) == code;
                const uint64_t mk = __ballot(hit);
                if (hit) ids[base + (uint64_t)__popcll(mk & ((1ull << lane) - 1))] = t;
                base += (uint64_t)__popcll(mk);
            }
        }
    }
}
// the two sweeps of a bucket: the word-parallel kernels for the 4-bit text (FMD_BUILD_SELECT_BYTES=1: symbol by symbol, the A/B switch), the generic ones otherwise
template <class Text>
static void launch_tile_count(hipStream_t st, unsigned grid, Text text, uint64_t n, int depth, uint32_t code, uint64_t n_tiles, uint64_t *counts)
{
    k_tile_count<<<grid, 256, 0, st>>>(text, n, depth, code, n_tiles, counts);
}
template <class Text>
static void launch_tile_select(hipStream_t st, unsigned grid, Text text, uint64_t n, int depth, uint32_t code, uint64_t n_tiles, const uint64_t *offset, uint64_t *ids)
{
    k_tile_select<<<grid, 256, 0, st>>>(text, n, depth, code, n_tiles, offset, ids);
}
static bool select_words() { static const bool on = !(getenv("FMD_BUILD_SELECT_BYTES") && atoi(getenv("FMD_BUILD_SELECT_BYTES"))); return on; }
template <>
void launch_tile_count<Text4>(hipStream_t st, unsigned grid, Text4 text, uint64_t n, int depth, uint32_t code, uint64_t n_tiles, uint64_t *counts)
{
    if (depth <= 4 && select_words()) k_tile_count4<<<grid, 256, 0, st>>>(text, n, depth, code, match4_of(code, depth), n_tiles, counts);
    else k_tile_count<<<grid, 256, 0, st>>>(text, n, depth, code, n_tiles, counts);
}
template <>
void launch_tile_select<Text4>(hipStream_t st, unsigned grid, Text4 text, uint64_t n, int depth, uint32_t code, uint64_t n_tiles, const uint64_t *offset, uint64_t *ids)
{
    if (depth <= 4 && select_words()) k_tile_select4<<<grid, 256, 0, st>>>(text, n, depth, code, match4_of(code, depth), n_tiles, offset, ids);
    else k_tile_select<<<grid, 256, 0, st>>>(text, n, depth, code, n_tiles, offset, ids);
}

// ---- FMD_BUILD_PARTITION=1 (off by default: tested on the fixtures only): the suffixes that have ENDED before a chunk -- distance to their '$' <= 21 * chunk,
// key 0 -- keep their order in front of the others, which is all a stable sort would do with them; they are 83 % of a bucket of 100-bp reads in the last chunk
// and 62 / 42 / 21 % in the ones before.  Two stable selections over tiles of the id array (count, scan, scatter by ballot prefix), then keys and sort for the rest.
template <class Rem>
__global__ void k_part_count(const uint64_t *__restrict__ ids, uint64_t m, Rem rem, uint32_t o0, uint64_t m_tiles, uint64_t *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    for (uint64_t tile = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < m_tiles; tile += (uint64_t)gridDim.x * (blockDim.x >> 6)) {
        const uint64_t i0 = tile << SEL_TILE_SHIFT, i1 = i0 + (1ull << SEL_TILE_SHIFT) < m ? i0 + (1ull << SEL_TILE_SHIFT) : m;
        uint32_t c = 0;
        for (uint64_t i = i0 + (uint64_t)lane; i < i1; i += 64) c += rem(ids[i]) <= o0;
        for (int o = 32; o; o >>= 1) c += __shfl_xor((int)c, o);
        if (lane == 0) counts[tile] = c;
    }
}
// ended[tile] = ended suffixes in front of the tile (exclusive scan of the counts), z = all of them: out[0, z) the ended ones, out[z, m) the others, both in order
template <class Rem>
__global__ void k_part_scatter(const uint64_t *__restrict__ ids, uint64_t m, Rem rem, uint32_t o0, uint64_t m_tiles, const uint64_t *__restrict__ ended, uint64_t z,
                               uint64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (uint64_t tile = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < m_tiles; tile += (uint64_t)gridDim.x * (blockDim.x >> 6)) {
        const uint64_t i0 = tile << SEL_TILE_SHIFT, i1 = i0 + (1ull << SEL_TILE_SHIFT) < m ? i0 + (1ull << SEL_TILE_SHIFT) : m;
        uint64_t bz = ended[tile], bn = z + (i0 - ended[tile]);
        for (uint64_t ib = i0; ib < i1; ib += 64) {           // all 64 lanes take every turn: the ballots need them
            const uint64_t i = ib + (uint64_t)lane;
            const bool in = i < i1;
            const uint64_t t = in ? ids[i] : 0;
            const bool end = in && rem(t) <= o0;
            const uint64_t mz = __ballot(end), mn = __ballot(in && !end), below = (1ull << lane) - 1;
            if (end) out[bz + (uint64_t)__popcll(mz & below)] = t;
            else if (in) out[bn + (uint64_t)__popcll(mn & below)] = t;
            bz += (uint64_t)__popcll(mz); bn += (uint64_t)__popcll(mn);
        }
    }
}

template <class Text>
__device__ __forceinline__ uint64_t text_key(Text text, uint64_t, uint64_t a, uint32_t mm)   // (4 bits per symbol: symbol by symbol)
{
    uint64_t key = 0;
    for (uint32_t j = 0; j < mm; ++j) key |= (uint64_t)text[a + j] << (3 * (20 - j));
    return key;
}
__device__ __forceinline__ uint64_t text_key(Text8 text, uint64_t n, uint64_t a, uint32_t mm) { return chunk_key(text.p, n, a, mm); }
// the same for the 4-bit text of the in-place builder (symbol t = nibble t & 1 of byte t >> 1: a little-endian stream of nibbles): 21 symbols are
// 84 bits of three aligned 8-byte words; sixteen symbols at a time go from nibbles to 3-bit fields, the first highest
__device__ __forceinline__ uint64_t text_key(Text4 text, uint64_t n, uint64_t a, uint32_t mm) { return chunk_key4(text.p, n, a, mm); }
template <class Text, class Rem>
__global__ void k_chunk_keys64(Text text, uint64_t n_text, uint64_t m, const uint64_t *__restrict__ ids, int chunk, Rem rem, uint64_t *__restrict__ keys)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t t = ids[i];
        const uint32_t r = rem(t), o0 = 21u * (uint32_t)chunk;
        uint64_t key = 0;
        if (o0 <= r) {
            const uint32_t mm = r - o0 + 1 < 21 ? r - o0 + 1 : 21;
            key = text_key(text, n_text, t + o0, mm);
        }
        keys[i] = key;
    }
}
template <class Text>
__global__ void k_emit_bwt64(Text text, const uint64_t *__restrict__ ids, uint64_t m, uint8_t *__restrict__ bwt)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t t = ids[i];
        bwt[i] = t ? (uint8_t)text[t - 1] : 0;   // text[t-1] is '$' (= 0) exactly when t starts a sequence
    }
}

// prefixes of `depth` symbols in increasing order: digits 0..5, nothing but zeros after a zero
static void prefix_codes(int depth, std::vector<uint32_t> &out)
{
    std::vector<uint32_t> cur(1, 0u);
    for (int j = 0; j < depth; ++j) {
        std::vector<uint32_t> nxt;
        for (uint32_t p : cur) {
            const bool ended = j > 0 && ((p & 7u) == 0);          // the last symbol was '$' (or the prefix ended earlier)
            if (ended) { nxt.push_back(p << 3); continue; }
            for (uint32_t c = 0; c < 6; ++c) nxt.push_back(p << 3 | c);
        }
        cur.swap(nxt);
    }
    out.swap(cur);   // generated in increasing numeric order
}

// Sink(d_bwt_slice, first_position, m): the BWT symbols of positions [first, first + m), in a device buffer that is reused
template <class Text, class Sink>
static int build_bucketed(hipStream_t st, Text text, uint64_t n, uint32_t max_len, int uniform_len, RemRagged rr, int depth, uint8_t *bwt_direct, Sink sink)
{
    const int n_chunks = (int)((max_len + 1 + 20) / 21);
    const bool tm = getenv("FMD_TIMING") != nullptr;     // phase times of the build (each phase synchronised: diagnostics only)
    double t_alloc = 0, t_sel = 0, t_keys = 0, t_sort = 0, t_emit = 0, t0 = bt_now(st, tm), t1;
    std::vector<uint32_t> codes;
    prefix_codes(depth, codes);
    uint64_t done = 0, cap = 0;
    DevPtr ids_a, ids_b, keys_a, keys_b, tmp, stmp, slice;     // grown to the largest bucket met so far
    size_t tb = 0, sb = 0;
    const uint64_t n_tiles = (n + (1ull << SEL_TILE_SHIFT) - 1) >> SEL_TILE_SHIFT;
    DevPtr tile_cnt, tile_off;
    DALLOC(tile_cnt, n_tiles * 8); DALLOC(tile_off, n_tiles * 8);
    FMD_HIP_TRY(fmd_exclusive_sum(nullptr, tb, (uint64_t *)tile_cnt.p, (uint64_t *)tile_off.p, (size_t)n_tiles, st));
    DALLOC(tmp, tb);
    const unsigned sel_grid = (unsigned)(n_tiles / 4 + 1 < 65536 ? n_tiles / 4 + 1 : 65536);
    std::vector<unsigned long long> sizes((size_t)1 << (3 * depth), 0ull);
    {
        DevPtr hist;
        DALLOC(hist, sizes.size() * 8);
        FMD_HIP_TRY(hipMemsetAsync(hist.p, 0, sizes.size() * 8, st));
        // a block counts at most 2^32 - 1 positions per bin in LDS: grid-stride over n with 4096 blocks x 1024 threads is far below that
        k_prefix_hist<<<4096, 1024, sizes.size() * 4, st>>>(text, n, depth, (unsigned long long *)hist.p);
        FMD_HIP_TRY(hipMemcpyAsync(sizes.data(), hist.p, sizes.size() * 8, hipMemcpyDeviceToHost, st));
        FMD_HIP_TRY(hipStreamSynchronize(st));
    }
    for (uint32_t code : codes) {
        bool has_end = false;                               // a '$' inside the prefix: the bucket is in order as it is
        for (int j = 0; j < depth; ++j) if (((code >> (3 * (depth - 1 - j))) & 7u) == 0) has_end = true;
        const uint64_t m = sizes[code];                      // from the histogram; the positions follow, ascending
        if (m == 0) continue;
        if (m > cap) {                                      // grow: release first, the arrays are the bulk of the footprint
            hipFree(ids_a.p); hipFree(ids_b.p); hipFree(keys_a.p); hipFree(keys_b.p); hipFree(stmp.p); hipFree(slice.p);
            ids_a.p = ids_b.p = keys_a.p = keys_b.p = stmp.p = slice.p = nullptr;
            cap = m + m / 16;
            DALLOC(ids_a, cap * 8); DALLOC(ids_b, cap * 8); DALLOC(keys_a, cap * 8); DALLOC(keys_b, cap * 8);
            if (!bwt_direct) DALLOC(slice, cap + 64);
            FMD_HIP_TRY(fmd_sort_pairs(nullptr, sb, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p, (uint64_t *)ids_a.p,
                                                          (uint64_t *)ids_b.p, (size_t)cap, 0, 63, st));
            DALLOC(stmp, sb);
        }
        if (tm) { t1 = bt_now(st, tm); t_alloc += t1 - t0; t0 = t1; }
        launch_tile_count(st, sel_grid, text, n, depth, code, n_tiles, (uint64_t *)tile_cnt.p);
        FMD_HIP_TRY(fmd_exclusive_sum(tmp.p, tb, (uint64_t *)tile_cnt.p, (uint64_t *)tile_off.p, (size_t)n_tiles, st));
        launch_tile_select(st, sel_grid, text, n, depth, code, n_tiles, (const uint64_t *)tile_off.p, (uint64_t *)ids_a.p);
        uint64_t *cur = (uint64_t *)ids_a.p;
        if (tm) { t1 = bt_now(st, tm); t_sel += t1 - t0; t0 = t1; }
        if (!has_end) {
            uint64_t *nxt = (uint64_t *)ids_b.p;
            size_t sb_m = 0;
            FMD_HIP_TRY(fmd_sort_pairs(nullptr, sb_m, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p, cur, nxt, (size_t)m, 0, 63, st));
            const uint64_t n_wide = getenv("FMD_BUILD_KEY_BYTES") && atoi(getenv("FMD_BUILD_KEY_BYTES")) ? 0 : n;   // A/B switch: 0 = every key byte by byte (round 3)
            // on everywhere (round 5: config 5's in-place build at depth 4 and the ragged 2*10^6-read set run with it under pytest -m gpu -- rank self-check over
            // every position, md5 of the .fmd against fermi build); FMD_BUILD_PARTITION=0 is the A/B switch
            const char *pe = getenv("FMD_BUILD_PARTITION");
            const bool part = pe ? atoi(pe) != 0 : true;
            for (int ch = n_chunks - 1; ch >= 0; --ch) {
                if (part && ch > 0) {   // (chunk 0: nothing has ended before the first symbol of a bucket without a '$' in its prefix)
                    const uint64_t m_tiles = (m + (1ull << SEL_TILE_SHIFT) - 1) >> SEL_TILE_SHIFT;
                    const uint32_t o0 = 21u * (uint32_t)ch;
                    const RemUniform ru{max_len + 1};
                    if (uniform_len) k_part_count<<<sel_grid, 256, 0, st>>>(cur, m, ru, o0, m_tiles, (uint64_t *)tile_cnt.p);
                    else k_part_count<<<sel_grid, 256, 0, st>>>(cur, m, rr, o0, m_tiles, (uint64_t *)tile_cnt.p);
                    FMD_HIP_TRY(fmd_exclusive_sum(tmp.p, tb, (uint64_t *)tile_cnt.p, (uint64_t *)tile_off.p, (size_t)m_tiles, st));
                    uint64_t last[2] = {0, 0};
                    FMD_HIP_TRY(hipMemcpyAsync(&last[0], (uint64_t *)tile_off.p + (m_tiles - 1), 8, hipMemcpyDeviceToHost, st));
                    FMD_HIP_TRY(hipMemcpyAsync(&last[1], (uint64_t *)tile_cnt.p + (m_tiles - 1), 8, hipMemcpyDeviceToHost, st));
                    FMD_HIP_TRY(hipStreamSynchronize(st));
                    const uint64_t z = last[0] + last[1];
                    if (z == m) continue;                     // every suffix of the bucket ended before this chunk: the order stands
                    if (z > 0) {
                        if (uniform_len) k_part_scatter<<<sel_grid, 256, 0, st>>>(cur, m, ru, o0, m_tiles, (const uint64_t *)tile_off.p, z, nxt);
                        else k_part_scatter<<<sel_grid, 256, 0, st>>>(cur, m, rr, o0, m_tiles, (const uint64_t *)tile_off.p, z, nxt);
                        if (tm) { t1 = bt_now(st, tm); t_sel += t1 - t0; t0 = t1; }
                        if (uniform_len) k_chunk_keys64<<<nblk(m - z, 256), 256, 0, st>>>(text, n_wide, m - z, nxt + z, ch, ru, (uint64_t *)keys_a.p);
                        else k_chunk_keys64<<<nblk(m - z, 256), 256, 0, st>>>(text, n_wide, m - z, nxt + z, ch, rr, (uint64_t *)keys_a.p);
                        if (tm) { t1 = bt_now(st, tm); t_keys += t1 - t0; t0 = t1; }
                        FMD_HIP_TRY(fmd_sort_pairs(stmp.p, sb_m, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p, nxt + z, cur + z, (size_t)(m - z), 0, 63, st));
                        FMD_HIP_TRY(hipMemcpyAsync(cur, nxt, z * 8, hipMemcpyDeviceToDevice, st));   // (the result is in `cur` again: no swap)
                        if (tm) { t1 = bt_now(st, tm); t_sort += t1 - t0; t0 = t1; }
                        continue;
                    }
                }
                if (uniform_len) k_chunk_keys64<<<nblk(m, 256), 256, 0, st>>>(text, n_wide, m, cur, ch, RemUniform{max_len + 1}, (uint64_t *)keys_a.p);
                else k_chunk_keys64<<<nblk(m, 256), 256, 0, st>>>(text, n_wide, m, cur, ch, rr, (uint64_t *)keys_a.p);
                // the first `depth` symbols are equal inside a bucket: chunk 0 sorts on the bits below them only
                const int end_bit = ch == 0 ? 63 - 3 * (depth < 21 ? depth : 21) : 63;
                if (tm) { t1 = bt_now(st, tm); t_keys += t1 - t0; t0 = t1; }
                FMD_HIP_TRY(fmd_sort_pairs(stmp.p, sb_m, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p, cur, nxt, (size_t)m, 0, end_bit, st));
                uint64_t *t = cur; cur = nxt; nxt = t;
                if (tm) { t1 = bt_now(st, tm); t_sort += t1 - t0; t0 = t1; }
            }
        }
        if (bwt_direct) k_emit_bwt64<<<nblk(m, 256), 256, 0, st>>>(text, cur, m, bwt_direct + done);
        else {
            k_emit_bwt64<<<nblk(m, 256), 256, 0, st>>>(text, cur, m, (uint8_t *)slice.p);
            const int rc = sink((const uint8_t *)slice.p, done, m);
            if (rc) return rc;
        }
        FMD_HIP_TRY(hipStreamSynchronize(st));
        if (tm) { t1 = bt_now(st, tm); t_emit += t1 - t0; t0 = t1; }
        done += m;
    }
    if (tm) fprintf(stderr, "[M::fmd_build] %llu symbols, depth %d: histogram + arrays (hipMalloc / hipFree) %.2f s, positions of the buckets %.2f s, keys %.2f s, sorts %.2f s, BWT out %.2f s\n",
                    (unsigned long long)n, depth, t_alloc, t_sel, t_keys, t_sort, t_emit);
    return done == n ? FMD_OK : FMD_E_HIP;
}
struct NoSink { int operator()(const uint8_t *, uint64_t, uint64_t) const { return FMD_OK; } };

// prefix depth of the bucketed builder: the shallowest whose largest bucket (estimated from the A/C/G/T share of the
// text, with room for skew) fits the free memory next to text and BWT
static int bucket_depth(uint64_t n, int fixed_bytes_per_symbol_already_allocated)
{
    (void)fixed_bytes_per_symbol_already_allocated;
    const char *e = getenv("FMD_BUILD_DEPTH");
    if (e && atoi(e) >= 1 && atoi(e) <= 4) return atoi(e);     // 8^4 histogram bins fit a workgroup's LDS
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 3;
    for (int d = 1; d <= 3; ++d) {
        double share = 1.0;
        for (int j = 0; j < d; ++j) share /= 4.0;
        const double need = 34.0 * 1.6 * share * (double)n;     // 32 bytes per suffix + scratch, 1.6x the even share
        if (need < 0.85 * (double)free_b) return d;
    }
    return 4;
}

extern "C" void fmd_dev_free(void *d_ptr) { if (d_ptr) hipFree(d_ptr); }

extern "C" int fmd_build_bwt_dev(int device, void *stream_, size_t n_reads, const uint8_t *d_reads, const uint64_t *d_off,
                                 uint64_t total_bases, uint32_t max_len, int uniform_len, uint8_t **d_bwt_out, uint64_t *n_sym_out)
{
    if (!d_reads || !d_off || !d_bwt_out || !n_sym_out || n_reads == 0 || max_len == 0) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream_;
    const uint64_t n = 2 * (total_bases + n_reads);
    const bool bucketed = n >= 0xffffffffull || getenv("FMD_BUILD_BUCKETED") != nullptr; // 32-bit suffix ids in the one-shot path
    DevPtr text, keys_a, keys_b, ord_a, ord_b, send, tmp;
    uint8_t *bwt = nullptr;
    DALLOC(text, n + 64);
    k_build_text<<<(unsigned)(n_reads < (1u << 24) ? n_reads : (1u << 24)), 64, 0, st>>>(n_reads, d_reads, d_off, (uint8_t *)text.p);
    RemRagged rr{nullptr, 2 * n_reads};
    if (!uniform_len) {
        DALLOC(send, 2 * n_reads * 8);
        k_seq_ends<<<nblk(n_reads, 256), 256, 0, st>>>(n_reads, d_off, (uint64_t *)send.p);
        rr.send = (const uint64_t *)send.p;
    }
    if (bucketed) {
        FMD_HIP_TRY(hipMalloc((void **)&bwt, n + 64));
        int rc = build_bucketed(st, Text8{(const uint8_t *)text.p}, n, max_len, uniform_len, rr, bucket_depth(n, 2), bwt, NoSink());
        if (rc) { hipFree(bwt); return rc; }
        *d_bwt_out = bwt; *n_sym_out = n;
        return FMD_OK;
    }
    DALLOC(keys_a, n * 8); DALLOC(keys_b, n * 8); DALLOC(ord_a, n * 4); DALLOC(ord_b, n * 4);
    size_t tmp_bytes = 0;
    FMD_HIP_TRY(fmd_sort_pairs(nullptr, tmp_bytes, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p,
                                                  (uint32_t *)ord_a.p, (uint32_t *)ord_b.p, (size_t)n, 0, 63, st));
    DALLOC(tmp, tmp_bytes);
    const int n_chunks = (int)((max_len + 1 + 20) / 21);
    uint32_t *cur = (uint32_t *)ord_a.p, *nxt = (uint32_t *)ord_b.p;
    k_iota32<<<nblk(n, 256), 256, 0, st>>>(cur, n); // text order = sequence-id order: the tie-break
    const uint64_t n_wide = getenv("FMD_BUILD_KEY_BYTES") && atoi(getenv("FMD_BUILD_KEY_BYTES")) ? 0 : n;   // A/B switch: 0 = every key byte by byte (round 3)
    for (int c = n_chunks - 1; c >= 0; --c) {
        if (uniform_len) k_chunk_keys<<<nblk(n, 256), 256, 0, st>>>((const uint8_t *)text.p, n, n_wide, cur, c, RemUniform{max_len + 1}, (uint64_t *)keys_a.p);
        else k_chunk_keys<<<nblk(n, 256), 256, 0, st>>>((const uint8_t *)text.p, n, n_wide, cur, c, rr, (uint64_t *)keys_a.p);
        FMD_HIP_TRY(fmd_sort_pairs(tmp.p, tmp_bytes, (uint64_t *)keys_a.p, (uint64_t *)keys_b.p, cur, nxt,
                                                      (size_t)n, 0, 63, st));
        uint32_t *t = cur; cur = nxt; nxt = t;
    }
    FMD_HIP_TRY(hipMalloc((void **)&bwt, n + 64));
    k_emit_bwt<<<nblk(n, 256), 256, 0, st>>>((const uint8_t *)text.p, cur, n, bwt);
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) { hipFree(bwt); fmd_set_hip_error(e, "build"); return FMD_E_HIP; }
    *d_bwt_out = bwt; *n_sym_out = n;
    return FMD_OK;
}

// ------------------------------------------------------------------ the index built in place (no byte BWT, packed text)
// For read sets whose text + BWT do not fit next to the index (7*10^8 x 100 bp: 1.4*10^11 symbols, 141 GB each): the text
// is kept 4 bits per symbol, and every bucket's slice of the BWT goes straight into the planes of the device index
// (fmd_index_put_slice) instead of into a byte array.  Footprint = n/2 (text) + 0.67 n (index) + 34 bytes per suffix of
// the largest bucket.  Reads of ONE length, appended in any number of calls so that they never have to sit in HBM together.
struct fmd_builder {
    int device; uint64_t n_reads, added; uint32_t len;
    uint64_t n_sym;
    uint8_t *text4;      // device: n_sym / 2 bytes (+ pad)
};
__global__ void k_text4_add(uint64_t n, uint32_t len, uint64_t first_read, const uint8_t *__restrict__ reads, uint8_t *__restrict__ text4)
{
    // one thread per PAIR of text positions of the added reads (a read's 2 * (len + 1) symbols start at an even position)
    const uint64_t per = (uint64_t)len + 1, pairs = n * per;   // 2 * per symbols per read = per pairs
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / per, k = 2 * (i - r * per);     // symbols k, k + 1 of read r's block
        const uint8_t *s = reads + r * (uint64_t)len;
        uint32_t v[2];
        for (int j = 0; j < 2; ++j) {
            const uint64_t q = k + (uint64_t)j;                 // 0 .. 2 * len + 1
            uint32_t c;
            if (q < len) c = s[q];
            else if (q == len || q == 2 * (uint64_t)len + 1) c = 0;
            else { const uint32_t b = s[len - 1 - (q - len - 1)]; c = (b >= 1 && b <= 4) ? 5 - b : b; }
            v[j] = c;
        }
        text4[(first_read + r) * per + (k >> 1)] = (uint8_t)(v[0] | v[1] << 4);
    }
}

extern "C" void fmd_builder_free(fmd_builder_t *b)
{
    if (!b) return;
    hipSetDevice(b->device);
    hipFree(b->text4);
    free(b);
}
extern "C" int fmd_builder_new(int device, uint64_t n_reads, uint32_t read_len, fmd_builder_t **out)
{
    if (!out || n_reads == 0 || read_len == 0 || read_len > 0xfffffff0u) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    fmd_builder *b = (fmd_builder *)calloc(1, sizeof(fmd_builder));
    if (!b) return FMD_E_NOMEM;
    b->device = device; b->n_reads = n_reads; b->len = read_len; b->n_sym = 2 * n_reads * ((uint64_t)read_len + 1);
    if (hipMalloc((void **)&b->text4, b->n_sym / 2 + 64) != hipSuccess) { fmd_set_hip_error(hipGetLastError(), "hipMalloc(text)"); free(b); return FMD_E_NOMEM; }
    *out = b;
    return FMD_OK;
}
extern "C" int fmd_builder_add_dev(fmd_builder_t *b, void *stream, uint64_t n, const uint8_t *d_reads)
{
    if (!b || (n && !d_reads) || b->added + n > b->n_reads) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(b->device));
    k_text4_add<<<nblk(n * ((uint64_t)b->len + 1), 256), 256, 0, (hipStream_t)stream>>>(n, b->len, b->added, d_reads, b->text4);
    b->added += n;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_text4_add"); return FMD_E_HIP; }
    return FMD_OK;
}
extern "C" int fmd_builder_finish(fmd_builder_t *b, fmd_dev_t **out)
{
    if (!b || !out || b->added != b->n_reads) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(b->device));
    FMD_HIP_TRY(hipDeviceSynchronize());
    fmd_dev *h = nullptr;
    int rc = fmd_index_alloc(b->device, b->n_sym, &h);
    if (rc) return rc;
    const uint64_t n = b->n_sym;
    hipStream_t st = nullptr;
    RemRagged rr{nullptr, 2 * b->n_reads};
    rc = build_bucketed(st, Text4{b->text4}, n, b->len, 1, rr, bucket_depth(n, 0), nullptr,
                        [&](const uint8_t *slice, uint64_t first, uint64_t m) { return fmd_index_put_slice(h, st, slice, first, m); });
    if (rc == FMD_OK) { hipFree(b->text4); b->text4 = nullptr; rc = fmd_index_finish(h); }   // the text goes before the count scratch comes
    if (rc) { fmd_dev_close(h); return rc; }
    fmd_builder_free(b);
    *out = h;
    return FMD_OK;
}

// Host-pointer form.  reads = nt6 bases of all reads back to back (no sentinels), off[n+1].
// bwt must hold 2 * (off[n] + n) bytes.
extern "C" int fmd_build_bwt(int device, size_t n_reads, const uint8_t *reads, const uint64_t *off, uint8_t *bwt, uint64_t *n_sym)
{
    if (!reads || !off || !bwt || !n_sym || n_reads == 0) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    uint32_t max_len = 0; int uniform = 1;
    for (size_t i = 0; i < n_reads; ++i) {
        const uint64_t l = off[i + 1] - off[i];
        if (l == 0 || l > 0xfffffff0ull) return FMD_E_ARG;
        if (l > max_len) max_len = (uint32_t)l;
        if (l != off[1] - off[0]) uniform = 0;
    }
    DevPtr dr, doff;
    DALLOC(dr, off[n_reads] + 64); DALLOC(doff, (n_reads + 1) * 8);
    FMD_HIP_TRY(hipMemcpy(dr.p, reads, off[n_reads], hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(doff.p, off, (n_reads + 1) * 8, hipMemcpyHostToDevice));
    uint8_t *d_bwt = nullptr;
    int rc = fmd_build_bwt_dev(device, nullptr, n_reads, (uint8_t *)dr.p, (uint64_t *)doff.p, off[n_reads], max_len, uniform, &d_bwt, n_sym);
    if (rc) return rc;
    hipError_t e = hipMemcpy(bwt, d_bwt, *n_sym, hipMemcpyDeviceToHost);
    hipFree(d_bwt);
    FMD_HIP_TRY(e);
    return FMD_OK;
}

// ---------------------------------------------------------------- BWT (device) -> RLE\6 bytes
// `len<<3 | sym`, len <= 31, long runs split: the stream `fermi ropebwt -b` writes
// (ropebwt.c:132-136) and rld_restore re-encodes (rld.c:295-308).
__global__ void k_run_bytes(const uint32_t *__restrict__ run_len, uint64_t n_runs, uint64_t *__restrict__ nb)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (uint64_t)gridDim.x * blockDim.x) nb[i] = (run_len[i] + 30) / 31;
}
__global__ void k_run_emit(const uint8_t *__restrict__ run_sym, const uint32_t *__restrict__ run_len, uint64_t n_runs,
                           const uint64_t *__restrict__ start, uint8_t *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t l = run_len[i];
        uint8_t *p = out + start[i];
        const uint8_t c = run_sym[i];
        for (; l > 31; l -= 31) *p++ = (uint8_t)(31 << 3 | c);
        *p = (uint8_t)(l << 3 | c);
    }
}

extern "C" int fmd_bwt_to_rle6(int device, const uint8_t *d_bwt, uint64_t n, uint8_t **h_rle6, uint64_t *n_bytes)
{
    if (!d_bwt || !h_rle6 || !n_bytes || n == 0) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    // chunks of 2^28 symbols: a run cut at a chunk border just becomes two adjacent runs of the same
    // symbol, which every reader of this stream merges (rld_enc, rld.c:177-184).  The device buffers are sized for the worst chunk (every
    // symbol its own run) ONCE: allocating and releasing 22 GB per 2^30 symbols was 1 s per chunk, the kernels are milliseconds.
    const uint64_t CH = 1ull << 28;
    const uint64_t mx = n < CH ? n : CH;
    struct HostBuf { uint8_t *p = nullptr; ~HostBuf() { free(p); } } hold;   // released to the caller on success only (FMD_HIP_TRY returns early)
    uint8_t *&h = hold.p; uint64_t h_n = 0, h_cap = 0;
    DevPtr sym, len, nruns, tmp, nb, start, out, t2;
    DALLOC(sym, mx); DALLOC(len, mx * 4); DALLOC(nruns, 8); DALLOC(nb, mx * 8); DALLOC(start, mx * 8); DALLOC(out, mx);
    size_t tb = 0, b2 = 0;
    FMD_HIP_TRY(fmd_run_length_encode(nullptr, tb, d_bwt, (unsigned)mx, (uint8_t *)sym.p, (uint32_t *)len.p, (uint64_t *)nruns.p));
    FMD_HIP_TRY(fmd_exclusive_sum(nullptr, b2, (uint64_t *)nb.p, (uint64_t *)start.p, (size_t)mx));
    DALLOC(tmp, tb); DALLOC(t2, b2);
    for (uint64_t o = 0; o < n; o += CH) {
        const uint64_t m = n - o < CH ? n - o : CH;
        size_t tb1 = tb, b21 = b2;
        FMD_HIP_TRY(fmd_run_length_encode(tmp.p, tb1, d_bwt + o, (unsigned)m, (uint8_t *)sym.p, (uint32_t *)len.p, (uint64_t *)nruns.p));
        uint64_t n_runs = 0;
        FMD_HIP_TRY(hipMemcpy(&n_runs, nruns.p, 8, hipMemcpyDeviceToHost));
        if (n_runs == 0 || n_runs > m) return FMD_E_HIP;
        k_run_bytes<<<nblk(n_runs, 256), 256>>>((uint32_t *)len.p, n_runs, (uint64_t *)nb.p);
        FMD_HIP_TRY(fmd_exclusive_sum(t2.p, b21, (uint64_t *)nb.p, (uint64_t *)start.p, (size_t)n_runs));
        uint64_t a = 0, b = 0;
        FMD_HIP_TRY(hipMemcpy(&a, (uint64_t *)start.p + n_runs - 1, 8, hipMemcpyDeviceToHost));
        FMD_HIP_TRY(hipMemcpy(&b, (uint64_t *)nb.p + n_runs - 1, 8, hipMemcpyDeviceToHost));
        const uint64_t total = a + b;
        if (total > m) return FMD_E_HIP;                 // cannot happen: a run of l symbols is ceil(l / 31) <= l bytes
        k_run_emit<<<nblk(n_runs, 256), 256>>>((uint8_t *)sym.p, (uint32_t *)len.p, n_runs, (uint64_t *)start.p, (uint8_t *)out.p);
        if (h_n + total > h_cap) {   // the first chunk's ratio for the whole stream + a fifth, then doubling
            const uint64_t est = o == 0 ? (uint64_t)((double)total * ((double)n / (double)m) * 1.2) + 64 : (h_n + total) * 2 + 64;
            h_cap = est > h_n + total ? est : h_n + total + 64;
            uint8_t *nh = (uint8_t *)realloc(h, h_cap);
            if (!nh) return FMD_E_NOMEM;
            h = nh;
        }
        hipError_t e = hipMemcpy(h + h_n, out.p, total, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { fmd_set_hip_error(e, "copy rle6"); return FMD_E_HIP; }
        h_n += total;
    }
    *h_rle6 = h; *n_bytes = h_n;
    h = nullptr;
    return FMD_OK;
}
extern "C" void fmd_host_free(void *p) { free(p); }
