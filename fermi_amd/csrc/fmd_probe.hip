// fmd_probe.hip -- random-gather microbenchmark: the practical HBM ceiling for the rank kernels.
// Same access machinery as fmd_wave.h (LDS-DMA, whole lines per lane group, one wave per
// workgroup, 16 KiB in flight per wave), but with pseudo-random line numbers and no arithmetic.
#include "fmd_internal.h"

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// LANES_PER_LINE = line_bytes / 16
template <int LANES_PER_LINE>
__global__ __launch_bounds__(64) void k_probe(const uint4 *__restrict__ ws, uint64_t n_lines, uint64_t iters_per_wave,
                                              uint32_t *__restrict__ sink)
{
    __shared__ uint4 lds[1024];
    const int lane = threadIdx.x & 63;
    const int sub = lane % LANES_PER_LINE, grp = lane / LANES_PER_LINE;
    constexpr int LINES_PER_INSTR = 64 / LANES_PER_LINE;
    uint32_t acc = 0;
    for (uint64_t it = 0; it < iters_per_wave; ++it) {
        const uint64_t base = ((uint64_t)blockIdx.x * iters_per_wave + it) * 16 * LINES_PER_INSTR;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint64_t line = mix64(base + (uint64_t)r * LINES_PER_INSTR + grp) % n_lines;
            const uint4 *src = ws + line * LANES_PER_LINE + sub;
            __builtin_amdgcn_global_load_lds((fmd_glb_void *)src, (fmd_lds_void *)(lds + r * 64), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += lds[lane * 16 + (it & 15)].x;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int fmd_probe_gather(int device, uint64_t ws_bytes, uint32_t line_bytes, uint64_t n_access, int iters, float *ms)
{
    if (!ms || (line_bytes != 64 && line_bytes != 128 && line_bytes != 256) || ws_bytes < (1u << 20)) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    FMD_HIP_TRY(hipGetDeviceProperties(&prop, device));
    uint4 *ws = nullptr; uint32_t *sink = nullptr;
    FMD_HIP_TRY(hipMalloc((void **)&ws, ws_bytes));
    FMD_HIP_TRY(hipMalloc((void **)&sink, 64));
    FMD_HIP_TRY(hipMemset(ws, 1, ws_bytes));
    const uint64_t n_lines = ws_bytes / line_bytes;
    const int lanes = (int)line_bytes / 16, lines_per_iter = 16 * (64 / lanes);
    const int grid = prop.multiProcessorCount * 10;
    uint64_t ipw = n_access / ((uint64_t)grid * lines_per_iter);
    if (ipw == 0) ipw = 1;
    hipEvent_t e0, e1;
    FMD_HIP_TRY(hipEventCreate(&e0)); FMD_HIP_TRY(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < iters + 1; ++i) {
        FMD_HIP_TRY(hipEventRecord(e0, 0));
        if (lanes == 4) k_probe<4><<<grid, 64>>>(ws, n_lines, ipw, sink);
        else if (lanes == 8) k_probe<8><<<grid, 64>>>(ws, n_lines, ipw, sink);
        else k_probe<16><<<grid, 64>>>(ws, n_lines, ipw, sink);
        FMD_HIP_TRY(hipEventRecord(e1, 0));
        FMD_HIP_TRY(hipEventSynchronize(e1));
        float t = 0;
        FMD_HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        if (i > 0 && t < best) best = t; // first pass = warm-up
    }
    // report time normalised to exactly n_access lines
    *ms = best * (float)((double)n_access / (double)(ipw * (uint64_t)grid * lines_per_iter));
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(ws); hipFree(sink);
    return FMD_OK;
}
