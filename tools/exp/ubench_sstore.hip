// micro-benchmark: can comparison masks leave a wave through SCALAR stores (s_store_dwordx4) at a useful rate on gfx950?
// each wave: ITER steps of 16 v_cmp (-> 16 x 64-bit masks) + 8 s_store_dwordx4 (512 B per step) vs. the same masks folded
// into VGPRs with v_addc (the k_sw_qp way) and written with vector stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_sstore(const float *in, unsigned long long *out, int iters)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float x = in[lane + 64 * (wave & 7)];
    unsigned long long *o0 = out + (size_t) wave * iters * 16;
    // wave-uniform pointer in SGPRs
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (unsigned long long) o0);
    const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) ((unsigned long long) o0 >> 32));
    unsigned long long *o = (unsigned long long *) (((unsigned long long) hi << 32) | lo);
    for (int it = 0; it < iters; ++it) {
        unsigned long long m[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            unsigned long long mm;
            asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(mm) : "v"(x), "v"((float) (k + it & 31)));
            m[k] = mm;
            x += 0.37f;
            if (x > 40.f) x -= 40.f;
        }
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            unsigned long long *p = o + (size_t) it * 16 + k;
            asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(__uint128_t(m[k]) | (__uint128_t(m[k + 1]) << 64)), "s"(p) : "memory");
        }
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256) void k_vstore(const float *in, unsigned *out, int iters)
{
    const int gt = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = gt >> 6;
    float x = in[lane + 64 * (wave & 7)];
    unsigned *o = out + (size_t) gt * iters;
    for (int it = 0; it < iters; ++it) {
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x), "v"((float) (k + it & 31)) : "vcc");
            x += 0.37f;
            if (x > 40.f) x -= 40.f;
        }
        o[it] = w;
    }
}

int main()
{
    const int iters = 256, blocks = 256 * 8, threads = 256;
    const size_t waves = (size_t) blocks * threads / 64;
    float *d_in; unsigned long long *d_out; unsigned *d_out2;
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float) ((i * 7919) % 40);
    CHECK(hipMalloc(&d_in, 2048));
    CHECK(hipMemcpy(d_in, h.data(), 2048, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, waves * iters * 16 * 8));
    CHECK(hipMalloc(&d_out2, (size_t) blocks * threads * iters * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_sstore, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("scalar stores: %.3f ms  (%.1f GB/s of masks, %.2f T cmp lane-ops/s)\n", ms, waves * iters * 128.0 / ms / 1e6, waves * iters * 16.0 * 64 / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_vstore, dim3(blocks), dim3(threads), 0, 0, d_in, d_out2, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("v_addc + vector stores: %.3f ms (%.2f T cmp lane-ops/s)\n", ms, waves * iters * 16.0 * 64 / ms / 1e9);
    }
    CHECK(hipDeviceSynchronize());
    // check: bit `lane` of mask k of (wave, it) == bit (15 - k) of the folded word of that lane
    std::vector<unsigned long long> m(iters * 16); std::vector<unsigned> w((size_t) 64 * iters);
    CHECK(hipMemcpy(m.data(), d_out + (size_t) 5 * iters * 16, m.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        std::vector<unsigned> ww(iters);
        CHECK(hipMemcpy(ww.data(), d_out2 + ((size_t) 5 * 64 + lane) * iters, iters * 4, hipMemcpyDeviceToHost));
        for (int it = 0; it < iters; ++it)
            for (int k = 0; k < 16; ++k) bad += ((m[it * 16 + k] >> lane) & 1) != ((ww[it] >> (15 - k)) & 1);
    }
    printf("mismatching bits: %zu\n", bad);
    return 0;
}
