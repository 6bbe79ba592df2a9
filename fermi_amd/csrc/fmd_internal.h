// fmd_internal.h -- host-side internals shared by the .hip translation units of libfmdhip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/fmd_hip.h"
#include "fmd_wave.h"

#define FMD_OVLP_MAX_PARTS 8
struct fmd_dev {
    int device;
    int n_cu;                 // compute units on this GPU
    uint64_t n_blocks;        // rank blocks incl. one trailing pad block
    uint64_t bytes;           // HBM bytes held
    uint4 *blocks;            // device
    uint64_t cnt[7], mcnt[7];
    uint4 *ptab;              // device: intervals of all strings of ptab_d bases (FmdIndexView)
    int ptab_d;
    unsigned long long *tail;  // device: FmdIndexView::tail
    uint4 *pair;               // device: two-base blocks (fmd_pair.hip), built on first use by the sorted overlap job; nullptr = none
    unsigned long long *pair_tab;
    uint64_t pair_bytes;
    int pair_tried;
    uint32_t *queues;         // device ring of work-queue heads for the persistent kernels
    uint32_t queue_next;      // host-side ring cursor (atomic)
    unsigned long long *stat; // device: FMD_STAT_SLOTS x FMD_STAT_STRIDE line counters (written by the instrumented build only)
    // second stream + events of the pipelined overlap batch (fmd_ovlp_dev), created on first use;
    // aux_busy (atomic) lets one call at a time use them, a concurrent call takes the serial path
    // device buffers kept between host-form calls (fmd_scratch_*): hipFree + hipMalloc of tens of GB per call cost
    // 1-2 s (`unitig` on 10 M reads: 1.05 s of a 1.2 s table pass); released by fmd_dev_close
    struct { void *p; size_t bytes; int busy; } scratch[24];
    int scratch_lock;
    hipStream_t aux_stream;
    hipEvent_t aux_ev[FMD_OVLP_MAX_PARTS + 1];
    int aux_ready, aux_busy;
    // side stream of fm6_get_nei's lane-per-strand kernel (k_ovl_nei on the strands k_ovl_classify sets aside runs beside the group
    // kernels: a few long dependent chains, nothing to gain from having the GPU to itself); same ownership rule as aux_*
    hipStream_t slow_stream;
    hipEvent_t slow_ev[2];
    int slow_ready, slow_busy;
};
#define FMD_N_QUEUES 256

void fmd_set_hip_error(hipError_t e, const char *what);

#define FMD_HIP_TRY(expr)                                       \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) {                                \
            fmd_set_hip_error(e__, #expr);                      \
            return e__ == hipErrorOutOfMemory ? FMD_E_NOMEM : FMD_E_HIP; \
        }                                                       \
    } while (0)

static inline FmdIndexView fmd_view(const fmd_dev *h)
{
    FmdIndexView v;
    v.blocks = h->blocks;
    for (int i = 0; i < 7; ++i) v.cnt[i] = h->cnt[i];
    v.n_sym = h->mcnt[0];
    v.n_seq = h->mcnt[1];
    v.ptab = h->ptab; v.ptab_d = h->ptab_d; v.tail = h->tail;
    v.pair = h->pair; v.pair_tab = h->pair_tab;
    v.stat = h->stat;
    return v;
}

// A device buffer of at least `bytes` from the handle's cache (hipMalloc when none fits); give it back with
// fmd_scratch_release.  Thread-safe; nullptr when the device is out of memory.
void *fmd_scratch_acquire(fmd_dev *h, size_t bytes);
void fmd_scratch_release(fmd_dev *h, void *p);

// the in-place index builder's hooks into fmd_index.hip
int fmd_index_alloc(int device, uint64_t n_sym, fmd_dev **out);
int fmd_index_put_slice(fmd_dev *h, hipStream_t st, const uint8_t *d_slice, uint64_t first, uint64_t m);
int fmd_index_finish(fmd_dev *h);

// the two-base blocks (fmd_pair.hip): built when forced (fmd_dev_build_pairs) or FMD_PAIR asks, and they fit; FMD_OK either way (h->pair says)
int fmd_pairs_ensure(fmd_dev *h, int force);

// next zeroed work-queue head for a persistent launch on `stream`
uint32_t *fmd_next_queue(fmd_dev *h, hipStream_t stream);

// persistent-grid size: waves (= 64-thread workgroups) to launch for n items
int fmd_grid_for(const fmd_dev *h, size_t n_items);
// same for a kernel that uses lds_bytes of LDS per 64-thread workgroup (160 KiB per CU)
int fmd_grid_for_lds(const fmd_dev *h, size_t n_items, size_t lds_bytes);

// Resident 64-thread workgroups per CU of `kernel` with lds_bytes of static LDS, for kernels that deal
// their work statically (round-robin or by ranges): a workgroup beyond the resident set would start when
// the others are done and then work through a full share alone.  The runtime's occupancy query, bounded by
// the LDS granule (1280 bytes on gfx950: 160 KiB / 128) and by `cap`.
#include <stdlib.h>
template <typename K>
static inline int fmd_resident_per_cu(K kernel, size_t lds_bytes, int cap, const char *name)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 64, 0) != hipSuccess || nb < 1) nb = 1;
    const int by_lds = lds_bytes ? (int)((160 * 1024) / (((lds_bytes + 1279) / 1280) * 1280)) : nb;
    if (nb > by_lds) nb = by_lds;
    if (nb > cap) nb = cap;
    if (getenv("FMD_DEBUG_OCC")) fprintf(stderr, "[occupancy] %s: %d workgroups per CU\n", name, nb);
    return nb;
}
