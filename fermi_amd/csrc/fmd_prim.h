// fmd_prim.h -- the three device-wide primitives the kernels' host code needs (exclusive prefix sum, LSD radix sort of
// (key, value) pairs, run-length encoding), taken from rocPRIM directly: the ROCm-native library, no CUB-compatibility layer
// in between.  All take (temporary storage, its size) the rocPRIM way: a call with tmp == nullptr only sets tmp_bytes.
#pragma once
#include <cstring>
#include <iterator>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

template <class In, class Out>
static inline hipError_t fmd_exclusive_sum(void *tmp, size_t &tmp_bytes, In in, Out out, size_t n, hipStream_t st = 0)
{
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, T(0), n, rocprim::plus<T>(), st);
}

template <class K, class V>
static inline hipError_t fmd_sort_pairs(void *tmp, size_t &tmp_bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n,
                                        unsigned begin_bit, unsigned end_bit, hipStream_t st = 0)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st);
}

template <class In, class Sym, class Len, class Cnt>
static inline hipError_t fmd_run_length_encode(void *tmp, size_t &tmp_bytes, In in, unsigned n, Sym sym, Len len, Cnt n_runs, hipStream_t st = 0)
{
    return rocprim::run_length_encode(tmp, tmp_bytes, in, n, sym, len, n_runs, st);
}
