// tools/probe_sweep.hip -- how many random lines per second does this memory system give, and does the answer depend on the SHAPE of the asking?
//
// DESIGN section 4 prices every kernel of the library against ONE probe (fmd_probe.hip: LDS-DMA, 16 instructions in flight per wave, a drain to vmcnt(0), ten
// waves per CU, line numbers from a 64-bit modulo).  This sweep varies everything that probe fixes: the load path (global_load_lds into LDS / global_load_dwordx4
// into VGPRs), the wait (drain / rolling vmcnt(N)), the number of instructions in flight, resident waves per CU, the cache-policy bits, the line size (64 / 128 B,
// and 16-byte pieces of 64 different lines per instruction = requests without bytes), and the working set (inside one L2, inside the Infinity Cache, 1 .. 64 GiB).
// Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/probe_sweep.hip -o build/probe_sweep ; build/probe_sweep [quick] > profiles/r6_probe/sweep.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// line number in [0, n_lines): multiply-high instead of a 64-bit modulo (which is ~150 VALU instructions on this ISA)
__device__ __forceinline__ uint64_t pick(uint64_t ctr, uint64_t n_lines, int use_mod)
{
    const uint64_t h = mix64(ctr);
    return use_mod ? h % n_lines : __umul64hi(h, n_lines);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int AUX> __device__ __forceinline__ uint4 ldv(const uint4 *src)
{
    if (AUX == 2) { const u32x4 t = __builtin_nontemporal_load((const u32x4 *)src); return make_uint4(t.x, t.y, t.z, t.w); }
    return *src;
}
// MODE 0: LDS-DMA, B instructions, drain.  MODE 1: LDS-DMA, B in flight, rolling vmcnt(B - 1).  MODE 2: VGPR loads, B, drain.  MODE 3: VGPR loads, rolling (the
// compiler's own vmcnt(B - 1) in an unrolled ring).  LPL = lanes per line (4: 64 B, 8: 128 B, 1: every lane its own line, 16 bytes of it).
template <int MODE, int B, int LPL, int AUX>
__global__ __launch_bounds__(64) void k_sweep(const uint4 *__restrict__ ws, uint64_t n_lines, uint64_t iters, uint32_t *__restrict__ sink, int use_mod)
{
    extern __shared__ uint4 lds[];
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPL, grp = lane / LPL;
    constexpr int LPI = 64 / LPL;                    // lines per instruction
    const uint64_t line_u4 = LPL == 1 ? 4 : LPL;     // uint4 per line (LPL == 1: 64-byte lines, one piece each)
    uint32_t acc = 0;
    uint64_t ctr = (uint64_t)blockIdx.x * iters * B * LPI + grp;
    if (MODE == 0) {
        for (uint64_t it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                const uint4 *src = ws + pick(ctr, n_lines, use_mod) * line_u4 + sub; ctr += LPI;
                __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(lds + r * 64), 16, 0, AUX);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        acc += lds[lane].x;
    } else if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < B; ++r) {
            const uint4 *src = ws + pick(ctr, n_lines, use_mod) * line_u4 + sub; ctr += LPI;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(lds + r * 64), 16, 0, AUX);
        }
        for (uint64_t it = 1; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                const uint4 *src = ws + pick(ctr, n_lines, use_mod) * line_u4 + sub; ctr += LPI;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B - 1) : "memory");     // the oldest one has landed: its slot is free
                __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(lds + r * 64), 16, 0, AUX);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += lds[lane].x;
    } else if (MODE == 2) {
        for (uint64_t it = 0; it < iters; ++it) {
            uint4 v[B];
#pragma unroll
            for (int r = 0; r < B; ++r) {
                const uint4 *src = ws + pick(ctr, n_lines, use_mod) * line_u4 + sub; ctr += LPI;
                v[r] = ldv<AUX>(src);
            }
#pragma unroll
            for (int r = 0; r < B; ++r) acc += v[r].x ^ v[r].w;
        }
    } else {
        uint4 v[B];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            const uint4 *src = ws + pick(ctr, n_lines, use_mod) * line_u4 + sub; ctr += LPI;
            v[r] = ldv<AUX>(src);
        }
        for (uint64_t it = 1; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                const uint4 *src = ws + pick(ctr, n_lines, use_mod) * line_u4 + sub; ctr += LPI;
                acc += v[r].x ^ v[r].w;
                v[r] = ldv<AUX>(src);
            }
        }
#pragma unroll
        for (int r = 0; r < B; ++r) acc += v[r].x ^ v[r].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

struct Variant { const char *name; const void *fn; int mode, b, lpl, aux; };
#define V(M, Bv, L, A) {#M "/" #Bv "/" #L "/" #A, (const void *)k_sweep<M, Bv, L, A>, M, Bv, L, A}
static const Variant VARS[] = {
    V(0, 16, 4, 0), V(0, 8, 4, 0), V(0, 4, 4, 0), V(0, 32, 4, 0),                 // 0-3: the library's shape and its depth
    V(1, 16, 4, 0), V(1, 8, 4, 0), V(1, 4, 4, 0), V(1, 32, 4, 0),                 // 4-7: rolling
    V(2, 16, 4, 0), V(2, 8, 4, 0), V(3, 16, 4, 0), V(3, 8, 4, 0), V(3, 4, 4, 0),  // 8-12: through VGPRs
    V(0, 8, 4, 2), V(0, 8, 4, 1), V(0, 8, 4, 16), V(0, 8, 4, 17), V(0, 8, 4, 3), V(0, 8, 4, 18), V(3, 16, 4, 2), V(1, 8, 4, 2), V(0, 16, 4, 2),   // 13-21: cache-policy bits: sc0 = 1, nt = 2, sc1 = 16
    V(0, 8, 8, 0), V(1, 8, 8, 0), V(3, 16, 8, 0), V(0, 4, 8, 0), V(0, 8, 8, 2), V(3, 16, 8, 2),   // 22-27: 128-byte lines
    V(0, 4, 1, 0), V(3, 16, 1, 0), V(0, 4, 1, 2),                                 // 28-30: 64 lines per instruction, 16 bytes of each
};

static double run(const Variant &v, const uint4 *ws, uint64_t ws_bytes, int waves_per_cu, int n_cu, uint64_t n_access, uint32_t *sink, int use_mod, int reps, int *waves_got)
{
    const uint64_t line_bytes = v.lpl == 1 ? 64 : (uint64_t)v.lpl * 16;
    const uint64_t n_lines = ws_bytes / line_bytes;
    const int lpi = 64 / v.lpl;
    const size_t lds_need = v.mode < 2 ? (size_t)v.b * 1024 : 1024;
    size_t lds = 160 * 1024 / (size_t)waves_per_cu / 1280 * 1280;      // LDS comes in 1280-byte granules: as much as lets waves_per_cu fit
    if (lds < lds_need) lds = lds_need;                                 // (then the occupancy query below says whether they do)
    if (lds > 64 * 1024) { if (hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1; }
    int per = 0;
    TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, v.fn, 64, lds));
    *waves_got = per;
    if (per < waves_per_cu) return -1;                                      // (registers: the VGPR forms at 32 waves)
    const int grid = n_cu * waves_per_cu;
    uint64_t iters = n_access / ((uint64_t)grid * v.b * lpi);
    if (iters < 2) iters = 2;
    hipEvent_t e0, e1;
    TRY(hipEventCreate(&e0)); TRY(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i <= reps; ++i) {
        void *args[] = {(void *)&ws, (void *)&n_lines, (void *)&iters, (void *)&sink, (void *)&use_mod};
        TRY(hipEventRecord(e0, 0));
        TRY(hipLaunchKernel(v.fn, dim3(grid), dim3(64), args, lds, 0));
        TRY(hipEventRecord(e1, 0));
        TRY(hipEventSynchronize(e1));
        float t = 0;
        TRY(hipEventElapsedTime(&t, e0, e1));
        if (i > 0 && t < best) best = t;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)iters * grid * v.b * lpi / (best * 1e-3);               // lines per second
}

int main(int argc, char **argv)
{
    const int quick = argc > 1 && strcmp(argv[1], "quick") == 0;
    hipDeviceProp_t prop;
    TRY(hipSetDevice(0));
    TRY(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    size_t free_b = 0, total_b = 0;
    TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t ws_max = quick ? (4ull << 30) : (64ull << 30);
    while (ws_max + (8ull << 30) > free_b) ws_max >>= 1;
    uint4 *ws = nullptr; uint32_t *sink = nullptr;
    TRY(hipMalloc((void **)&ws, ws_max));
    TRY(hipMalloc((void **)&sink, 64));
    TRY(hipMemset(ws, 1, ws_max));
    TRY(hipDeviceSynchronize());
    printf("# %s, %d CUs, working set up to %.0f GiB; columns: variant = MODE/B/lanes-per-line/aux  (MODE 0 LDS-DMA drain, 1 LDS-DMA rolling vmcnt(B-1), 2 VGPR drain, 3 VGPR rolling;\n", prop.name, n_cu, ws_max / 1073741824.0);
    printf("#   aux: 1 = sc0, 2 = nt, 16 = sc1; lanes-per-line 4 = 64-byte lines, 8 = 128-byte lines, 1 = 64 lines per instruction, 16 bytes of each)\n");
    printf("%-14s %6s %8s %5s %12s %10s\n", "variant", "waves", "ws_GiB", "mod", "Glines/s", "TB/s");
    const uint64_t n_access = quick ? (1ull << 28) : (1ull << 31);
    const int reps = quick ? 1 : 2;
    auto line = [&](const Variant &v, int w, uint64_t wsb, int use_mod) {
        int got = 0;
        const double r = run(v, ws, wsb, w, n_cu, n_access, sink, use_mod, reps, &got);
        const double lb = v.lpl == 1 ? 16 : v.lpl * 16;
        if (r < 0) printf("%-14s %6d %8.3f %5d %12s %10s   (does not fit: %d waves per CU)\n", v.name, w, wsb / 1073741824.0, use_mod, "-", "-", got);
        else printf("%-14s %6d %8.3f %5d %12.2f %10.3f\n", v.name, w, wsb / 1073741824.0, use_mod, r * 1e-9, r * lb * 1e-12);
        fflush(stdout);
    };
    const uint64_t ws_main = ws_max < (16ull << 30) ? ws_max : (16ull << 30);
    // 1. the library's probe as it is (64-bit modulo for the line number; 16 KiB of LDS per wave: nine fit), then the same with a multiply-high
    line(VARS[0], 9, ws_main, 1);
    line(VARS[0], 9, ws_main, 0);
    line(VARS[1], 10, ws_main, 1);
    line(VARS[1], 10, ws_main, 0);
    line(VARS[22], 10, ws_main, 1);     // 128-byte lines, the modulo's cost per LINE doubles
    line(VARS[22], 10, ws_main, 0);
    // 2. every shape at 10 waves per CU (or what fits), 16 GiB
    for (size_t i = 0; i < sizeof(VARS) / sizeof(VARS[0]); ++i) {
        const int w = VARS[i].mode < 2 && VARS[i].b == 32 ? 4 : (VARS[i].mode < 2 && VARS[i].b == 16 ? 9 : 10);
        line(VARS[i], w, ws_main, 0);
    }
    // 3. resident waves per CU
    const int waves[] = {1, 2, 4, 8, 16, 32};
    for (int w : waves) {
        line(VARS[0], w, ws_main, 0);   // 0/16
        line(VARS[2], w, ws_main, 0);   // 0/4: up to 32
        line(VARS[6], w, ws_main, 0);   // rolling 4
        line(VARS[11], w, ws_main, 0);  // VGPR rolling 8
        line(VARS[25], w, ws_main, 0);  // 128-byte lines, 0/4
        line(VARS[19], w, ws_main, 0);  // VGPR rolling 16, nt
    }
    // 4. working set: one L2 (2 MiB), the Infinity Cache (64, 192 MiB), then HBM; TLB reach shows between 1 and 64 GiB
    const double sizes[] = {0.002, 0.0625, 0.1875, 0.5, 1, 2, 4, 8, 11, 16, 32, 64};
    for (double g : sizes) {
        const uint64_t b = (uint64_t)(g * 1073741824.0) & ~(uint64_t)4095;
        if (b > ws_max) continue;
        line(VARS[1], 10, b, 0);    // 64-byte lines
        line(VARS[13], 10, b, 0);   // 64-byte lines, nt
        line(VARS[22], 10, b, 0);   // 128-byte lines
        line(VARS[26], 10, b, 0);   // 128-byte lines, nt
        line(VARS[28], 10, b, 0);   // 16-byte pieces
    }
    (void)hipFree(ws); (void)hipFree(sink);
    return 0;
}
