// ubench_sw.hip -- the Mu SW cell recurrence in isolation (registers only), to separate VALU issue
// limits from LDS/branch effects in k_mu_sw.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
#define NCOL 2048
template <int MODE, int R> __global__ __launch_bounds__(256) void k(int *out, int seed, int open, int ext)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int H[R], E[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { H[r] = 0; E[r] = 0; }
    int S[R];
#pragma unroll
    for (int r = 0; r < R; ++r) S[r] = ((seed + r * 7 + threadIdx.x) % 12) - 7;
    int best = 0, diag_in = 0, F0 = 0;
    const char *p = smem + (threadIdx.x & 63) * 16;
    for (int col = 0; col < NCOL; ++col) {
        if (MODE == 1) {          // LDS: R/4 b128 reads (int32 profile)
#pragma unroll
            for (int k = 0; k < R / 4; ++k) {
                v4i v = *(const volatile v4i *) (p + ((col + k) & 15) * 1024);
                S[4 * k] = v.x; S[4 * k + 1] = v.y; S[4 * k + 2] = v.z; S[4 * k + 3] = v.w;
            }
        }
        int diag = diag_in, F = F0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = diag + S[r];
            h = max(h, 0);
            h = max(h, E[r]);
            h = max(h, F);
            diag = H[r];
            H[r] = h;
            best = max(best, h);
            const int ho = h - open;
            E[r] = max(E[r] - ext, ho);
            F = max(F - ext, ho);
        }
        diag_in = H[R - 1] & 0xFF;
        F0 = F & 0x7F;
        asm volatile("" : "+v"(S[0]));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}
template <int MODE, int R> void run(const char *name, int blocks_per_cu)
{
    int *d; (void) hipMalloc(&d, 256 * 256 * 8 * 4);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    dim3 grid(256 * blocks_per_cu), blk(256);
    hipLaunchKernelGGL((k<MODE, R>), grid, blk, 32768, 0, d, 3, 2, 1);
    (void) hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, R>), grid, blk, 32768, 0, d, 3, 2, 1);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double cells = (double) grid.x * 256 * NCOL * R;
    printf("%-40s %8.3f ms  %7.2f Tcells/s   %.1f SIMD-cycles per wave-cell @2.4GHz\n", name, ms, cells / ms / 1e9,
           ms * 1e-3 * 2.4e9 * 1024 / (cells / 64));
    (void) hipFree(d);
}
int main()
{
    run<0, 32>("regs only R=32, 1 wave/SIMD", 1);
    run<0, 32>("regs only R=32, 2 waves/SIMD", 2);
    run<0, 32>("regs only R=32, 4 waves/SIMD", 4);
    run<0, 16>("regs only R=16, 4 waves/SIMD", 4);
    run<0, 16>("regs only R=16, 8 waves/SIMD", 8);
    run<1, 32>("b128 LDS profile R=32, 4 waves/SIMD", 4);
    run<1, 16>("b128 LDS profile R=16, 8 waves/SIMD", 8);
    return 0;
}
