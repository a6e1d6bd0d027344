// ubench_f64max.hip -- issue rates behind two choices in k_sw_qp: v_max_f64 next to the 32-bit ops of the best-cell tracking
// (is a 64-bit maximum on {~column, score} pairs cheaper than v_cmp_gt_f32 + 2 v_cndmask_b32?), and v_mad_u32_u16 against
// the other address forms (measured: all of them issue at the full rate -- an experiment that replaced k_sw_qp's eight
// v_mad_u32_u16 per step by four shifts + eight SDWA adds changed nothing: 18.7 vs 18.5 ms).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITERS 4096
template <int MODE> __global__ __launch_bounds__(256) void k(int *out, int seed)
{
    double a[8];
    float x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; x[i] = seed * 0.5f + i; y[i] = threadIdx.x + i; }
    double s = seed * 1.25 + threadIdx.x;
    float f = seed * 0.75f;
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 1) asm volatile("v_add_f32 %1, %1, %2\n\tv_max_f64 %0, %0, %3" : "+v"(a[i]), "+v"(x[i]) : "v"(f), "v"(s));
            if (MODE == 2) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(x[i]), "+v"(y[i]) : "v"(f), "v"(seed) : "vcc");
            if (MODE == 3) asm volatile("v_add_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %0" : "+v"(x[i]), "+v"(y[i]) : "v"(f));
            if (MODE == 4) asm volatile("v_add_f32 %1, %1, %2\n\tv_max_f64 %0, %0, %3\n\tv_mov_b32 %4, %1" : "+v"(a[i]), "+v"(x[i]) : "v"(f), "v"(s), "v"(y[i]));
            // address arithmetic of the float-SW kernels: a 16-bit half of a packed word times a stride plus a base
            if (MODE == 5) asm volatile("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(x[i]) : "v"(y[i]), "s"(seed));
            if (MODE == 6) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "s"(seed));
            if (MODE == 7) asm volatile("v_lshrrev_b32 %1, 16, %2\n\tv_mad_u32_u24 %0, %1, %3, %0" : "+v"(x[i]), "+v"(y[i]) : "v"(f), "s"(seed));
            if (MODE == 8) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x[i]) : "v"(y[i]));
        }
        asm volatile("" : "+v"(s), "+v"(f));
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int) r;
}
template <int MODE> void run(const char *name, double ops_per_iter)
{
    int *d; hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * 4), blk(256);
    hipLaunchKernelGGL(k<MODE>, grid, blk, 0, 0, d, 3);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, grid, blk, 0, 0, d, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double insts = (double) grid.x * 256 * N_ITERS * ops_per_iter;
    printf("%-44s %8.3f ms  %8.2f T lane-instructions/s\n", name, ms, insts / ms / 1e9);
    hipFree(d);
}
int main()
{
    run<0>("v_max_f64 (8 chains)", 8);
    run<1>("v_add_f32 + v_max_f64 (2 ops)", 16);
    run<2>("v_cmp_gt_f32 + 2 v_cndmask_b32 (3 ops)", 24);
    run<3>("v_add_f32 + v_max_f32 (2 ops)", 16);
    run<4>("v_add_f32 + v_max_f64 + v_mov_b32 (3 ops)", 24);
    run<5>("v_mad_u32_u16 op_sel (1 op)", 8);
    run<6>("v_mad_u32_u24 (1 op)", 8);
    run<7>("v_lshrrev_b32 + v_mad_u32_u24 (2 ops)", 16);
    run<8>("v_add_u32_sdwa src0_sel:WORD_1 (1 op)", 8);
    return 0;
}
