// rsk_dev_tables.h -- per-translation-unit device copies of the constant tables (no -fgpu-rdc:
// each .hip file owns its __constant__ symbols and uploads them once per device).
#pragma once
#include "rsk_internal.h"
#include "rsk_tables_data.h"

static __device__ __constant__ signed char c_mu_int[36 * 36];   // IntScoreMx_Mu (mumx_data.cpp:42)

static int rsk_upload_mu_tables(rsk_ctx *ctx)
{
    static std::atomic<int> done[64];          // one per translation unit that includes this header (each has its own c_mu_int)
    return rsk_once_per_device(done, ctx->device, [&]() -> int {
        RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_mu_int), rsk_mu_int, sizeof(rsk_mu_int)));
        return RSK_OK;
    });
}
