// k_pairs_sort.hip -- deterministic order of a device pair list.
// rsk_mu_filter_dev appends its survivors in no particular order (atomics); the reference walks pairs row-major
// (GetNextPairSelf runself.cpp:72-99, GetNextPairQuery runquery.cpp), so the host layer wants them by (A-side chain,
// B-side chain).  Millions of survivors through a host counting sort + per-chain std::sort were 0.15 s of an 11,211-chain
// search; here: one 64-bit radix sort on the device (hipcub), the host receives the two columns already ordered.
#include <algorithm>

#include <hipcub/hipcub.hpp>

#include "rsk_internal.h"

__global__ void k_pairs_pack(const uint32_t *major, const uint32_t *minor, size_t n, unsigned long long *keys)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) keys[k] = ((unsigned long long) major[k] << 32) | minor[k];
}

__global__ void k_pairs_unpack(const unsigned long long *keys, size_t n, uint32_t *major, uint32_t *minor)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { major[k] = (uint32_t) (keys[k] >> 32); minor[k] = (uint32_t) keys[k]; }
}

extern "C" int rsk_pairs_sort_dev(rsk_ctx *ctx, uint32_t *d_major, uint32_t *d_minor, size_t n, uint32_t major_bound)
{
    if (!ctx || (n && (!d_major || !d_minor))) { rsk_set_error("rsk_pairs_sort_dev: NULL argument"); return RSK_E_INVALID; }
    if (n == 0) return RSK_OK;
    if (n > 0x7FFFFFFFull) { rsk_set_error("rsk_pairs_sort_dev: more than 2^31 pairs"); return RSK_E_RANGE; }
    RSK_HIP(hipSetDevice(ctx->device));
    rsk_scratch ws(ctx);
    unsigned long long *d_in = nullptr, *d_out = nullptr;
    void *d_tmp = nullptr;
    int rc;
    if ((rc = ws.alloc(&d_in, n)) != RSK_OK || (rc = ws.alloc(&d_out, n)) != RSK_OK) return rc;
    int end_bit = 64;                                      // only the bits the major index can occupy
    if (major_bound) { end_bit = 32; while (end_bit < 64 && ((unsigned long long) (major_bound - 1) >> (end_bit - 32)) != 0) ++end_bit; }
    size_t tmp_bytes = 0;
    RSK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_in, d_out, (int) n, 0, end_bit, ctx->stream));
    if ((rc = ws.alloc((void **) &d_tmp, std::max<size_t>(tmp_bytes, 16))) != RSK_OK) return rc;
    const unsigned nb = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(k_pairs_pack, dim3(nb), dim3(256), 0, ctx->stream, d_major, d_minor, n, d_in);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_in, d_out, (int) n, 0, end_bit, ctx->stream));
    hipLaunchKernelGGL(k_pairs_unpack, dim3(nb), dim3(256), 0, ctx->stream, d_out, n, d_major, d_minor);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}

// (query, target, score) triples of rsk_mu_prefilter_dev (unordered) -> 64-bit keys query << 48 | target << 16 | score in
// ascending order = grouped by query, a query's triples by target: the order RankedScoresBag sees them in with -threads 1
// (targets ascending, muprefilter.cpp:21-60), so the host replay (rsk_rsb_select_keys) is one linear pass per query.
__global__ void k_triples_pack(const uint32_t *q, const uint32_t *t, const uint32_t *s, size_t n, unsigned long long *keys)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) keys[k] = ((unsigned long long) q[k] << 48) | ((unsigned long long) t[k] << 16) | (s[k] & 0xFFFFu);
}

extern "C" int rsk_triples_sort_dev(rsk_ctx *ctx, const uint32_t *d_q, const uint32_t *d_t, const uint32_t *d_score, size_t n, uint64_t *d_keys)
{
    if (!ctx || (n && (!d_q || !d_t || !d_score || !d_keys))) { rsk_set_error("rsk_triples_sort_dev: NULL argument"); return RSK_E_INVALID; }
    if (n == 0) return RSK_OK;
    RSK_HIP(hipSetDevice(ctx->device));
    rsk_scratch ws(ctx);
    unsigned long long *d_in = nullptr;
    void *d_tmp = nullptr;
    int rc;
    if ((rc = ws.alloc(&d_in, n)) != RSK_OK) return rc;
    size_t tmp_bytes = 0;
    RSK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_in, (unsigned long long *) d_keys, n, 0, 64, ctx->stream));
    if ((rc = ws.alloc((void **) &d_tmp, std::max<size_t>(tmp_bytes, 16))) != RSK_OK) return rc;
    hipLaunchKernelGGL(k_triples_pack, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, d_q, d_t, d_score, n, d_in);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_in, (unsigned long long *) d_keys, n, 0, 64, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}

int rsk_sort_pairs_u64_u32(rsk_ctx *ctx, const unsigned long long *d_keys_in, unsigned long long *d_keys_out, const uint32_t *d_vals_in,
                           uint32_t *d_vals_out, size_t n, int end_bit)
{
    if (n == 0) return RSK_OK;
    rsk_scratch ws(ctx);
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    RSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, 0, end_bit, ctx->stream));
    const int rc = ws.alloc(&d_tmp, std::max<size_t>(tmp_bytes, 16));
    if (rc != RSK_OK) return rc;
    RSK_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, 0, end_bit, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));            // the temporary goes back to the pool with `ws`
    return RSK_OK;
}
