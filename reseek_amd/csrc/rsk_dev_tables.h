// rsk_dev_tables.h -- per-translation-unit device copies of the constant tables (no -fgpu-rdc:
// each .hip file owns its __constant__ symbols and uploads them once per device).
#pragma once
#include "rsk_internal.h"
#include "rsk_tables_data.h"

static __device__ __constant__ signed char c_mu_int[36 * 36];   // IntScoreMx_Mu (mumx_data.cpp:42)

static int rsk_upload_mu_tables(rsk_ctx *ctx)
{
    static bool done[64] = { false };
    if (ctx->device < 64 && done[ctx->device]) return RSK_OK;
    RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_mu_int), rsk_mu_int, sizeof(rsk_mu_int)));
    if (ctx->device < 64) done[ctx->device] = true;
    return RSK_OK;
}
