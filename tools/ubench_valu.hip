// ubench_valu.hip -- measures issue rates of the instructions the gapless ring kernel is made of
// (packed int16 add/max, DPP move, ds_read_b128) on the current GPU.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v2s __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define N_ITERS 4096
template <int MODE> __global__ __launch_bounds__(256) void k(int *out, int seed)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
    int s = seed | 1;
    const char *p = smem + (threadIdx.x & 63) * 16;
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = __builtin_bit_cast(int, __builtin_elementwise_add_sat(__builtin_bit_cast(v2s, a[i]), __builtin_bit_cast(v2s, s)));
            if (MODE == 1) a[i] = __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(v2s, a[i]), __builtin_bit_cast(v2s, s)));
            if (MODE == 2) a[i] = a[i] + s;                       // v_add_u32
            if (MODE == 3) a[i] = max(a[i], s);                   // v_max_i32
            if (MODE == 4) a[i] = __builtin_amdgcn_update_dpp(a[i], a[i], 0x13C, 0xF, 0xF, false);
            if (MODE == 7) asm volatile("v_max3_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(seed));
            if (MODE == 8) asm volatile("v_max3_i16 %0, %0, %1, %2 op_sel:[1,1,1,1]" : "+v"(a[i]) : "v"(s), "v"(seed));
            if (MODE == 9) asm volatile("v_add_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(s));
            if (MODE == 10) asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(s));
            if (MODE == 11) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(seed));
            if (MODE == 12) asm volatile("v_pk_mad_i16 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(s), "v"(seed));
            if (MODE == 13) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(seed));
            if (MODE == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
            if (MODE == 15) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 16) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 17) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 18) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(s));
            if (MODE == 19) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(s));
            if (MODE == 20) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 21) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 22) asm volatile("v_bfe_i32 %0, %1, 8, 8" : "+v"(a[i]) : "v"(s));
            if (MODE == 23) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
            if (MODE == 24) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 25) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 26) asm volatile("v_max_i32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(s));
            if (MODE == 27) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 28) asm volatile("v_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 29) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(s));
            if (MODE == 30) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i & 1]) : "v"(s));
            if (MODE == 31) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i & 3]) : "v"(s));
            if (MODE == 40) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(s));
            if (MODE == 41) { if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s)); else asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(s)); }
            if (MODE == 42) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            if (MODE == 43) asm volatile("v_max_i32 %0, 0, %0" : "+v"(a[i]));
            if (MODE == 44) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s)); }
            if (MODE == 45) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            if (MODE == 46) asm volatile("v_max_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            if (MODE == 47) asm volatile("v_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %2\n\tv_subrev_u32 %0, %3, %0" : "+v"(a[i]) : "v"(s), "v"(a[(i + 3) & 15]), "s"(seed));
            if (MODE == 5) asm volatile("v_pk_add_i16 %0, %0, %1 clamp\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 50) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(seed));
            if (MODE == 51) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 52) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 53) asm volatile("v_pk_maximum3_f16 %0, %0, %1, 0" : "+v"(a[i]) : "v"(s));
            if (MODE == 54) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(0xFFFF), "v"(s));
            if (MODE == 55) asm volatile("v_pk_add_f16 %0, %0, %1\n\tv_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(seed));
        }
        if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v4i v = *(const volatile v4i *) (p + ((it + i) & 15) * 1024);
                a[i * 4] += v.x; a[i * 4 + 1] += v.y; a[i * 4 + 2] += v.z; a[i * 4 + 3] += v.w;
            }
        }
        if (MODE != 6) asm volatile("" : "+v"(s));
    }
    int r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, double ops_per_iter, int blocks_per_cu)
{
    int *d; hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * blocks_per_cu), blk(256);
    hipLaunchKernelGGL(k<MODE>, grid, blk, 32768, 0, d, 3);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, grid, blk, 32768, 0, d, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double lane_ops = (double) grid.x * 256 * N_ITERS * ops_per_iter;
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s  (%.1f lane-ops/clk/CU @2.4GHz)\n", name, ms, lane_ops / ms / 1e9,
           lane_ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(d);
}
int main()
{
    run<0>("v_pk_add_i16 clamp", 16, 8);
    run<1>("v_pk_max_i16", 16, 8);
    run<2>("v_add_u32", 16, 8);
    run<3>("v_max_i32", 16, 8);
    run<4>("v_mov_b32_dpp wave_ror:1", 16, 8);
    run<5>("pk_add+pk_max dep pair", 32, 8);
    run<50>("v_pk_maximum3_f16", 16, 8);
    run<51>("v_pk_add_f16", 16, 8);
    run<52>("v_pk_max_f16", 16, 8);
    run<53>("v_pk_maximum3_f16 x, y, 0", 16, 8);
    run<54>("v_bfi_b32", 16, 8);
    run<55>("pk_add_f16+pk_maximum3_f16 dep pair", 32, 8);
    run<6>("ds_read_b128 (B per lane=16)", 4, 4);
    run<7>("v_max3_i16", 16, 8);
    run<8>("v_max3_i16 op_sel hi", 16, 8);
    run<9>("v_add_i16 clamp (vop3)", 16, 8);
    run<10>("v_add_i32 clamp", 16, 8);
    run<11>("v_max3_i32", 16, 8);
    run<29>("v_add_u32 1 chain, 8 w/SIMD", 16, 8);
    run<29>("v_add_u32 1 chain, 4 w/SIMD", 16, 4);
    run<29>("v_add_u32 1 chain, 1 w/SIMD", 16, 1);
    run<30>("v_add_u32 2 chains, 4 w/SIMD", 16, 4);
    run<30>("v_add_u32 2 chains, 1 w/SIMD", 16, 1);
    run<31>("v_add_u32 4 chains, 4 w/SIMD", 16, 4);
    run<31>("v_add_u32 4 chains, 1 w/SIMD", 16, 1);
    run<2>("v_add_u32 16 chains, 4 w/SIMD", 16, 4);
    run<2>("v_add_u32 16 chains, 1 w/SIMD", 16, 1);
    run<40>("v_add_u32 d=a[i+1]+s (not in place)", 16, 4);
    run<41>("alternate add/max in place", 16, 4);
    run<42>("v_subrev_u32 a, sgpr, a", 16, 4);
    run<43>("v_max_i32 a, 0, a", 16, 4);
    run<45>("v_add_u32 d = x + y (3 regs)", 16, 4);
    run<46>("v_max_i32 d = max(x, y) (3 regs)", 16, 4);
    run<47>("add;max;subrev triple (3 ops)", 48, 4);
    run<19>("v_add_u32_sdwa sext byte", 16, 8);
    run<20>("v_add_f32", 16, 8);
    run<21>("v_max_f32", 16, 8);
    run<22>("v_bfe_i32", 16, 8);
    run<23>("v_cmp_gt_f32+v_addc (2 ops)", 32, 8);
    run<24>("v_sub_u32", 16, 8);
    run<25>("v_max_u32", 16, 8);
    run<26>("v_max_i32_sdwa sext word", 16, 8);
    run<27>("v_sub_u16", 16, 8);
    run<28>("v_max_u16", 16, 8);
    run<12>("v_pk_mad_i16 clamp", 16, 8);
    run<13>("v_alignbit_b32", 16, 8);
    run<14>("v_cndmask_b32", 16, 8);
    run<15>("v_mov_b32_dpp row_shr:1", 16, 8);
    run<16>("v_max_i16 (vop2)", 16, 8);
    run<17>("v_pk_add_u16", 16, 8);
    run<18>("v_pk_sub_u16 clamp", 16, 8);
    return 0;
}
