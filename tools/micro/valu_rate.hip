// Microbenchmark (diagnostic): issue rate of plain and packed fp32 VALU ops on gfx950 at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, unsigned long long *cyc)
{
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0000001f, c = 1e-9f;
    const float2v pm = {m, m}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent v_mul_f32
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            }
        } else if (MODE == 1) {  // 8 independent v_add_f32
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            }
        } else if (MODE == 2) {  // 4 independent v_pk_mul_f32 (8 values)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
            }
        } else if (MODE == 3) {  // v_pk_add_f32
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
            }
        } else if (MODE == 4) {  // v_fma_f32
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (MODE == 5) {  // v_pk_fma_f32
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            }
        } else if (MODE == 7) {  // v_mov_b32_dpp wave_shr:1
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 8) {  // v_sub_f32_dpp with a wave_shl:1 operand
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("s_nop 1\n v_sub_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %1, %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                             "v_sub_f32_dpp %2, %3, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %3, %4, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                             "v_sub_f32_dpp %4, %5, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %5, %6, %7 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                             "v_sub_f32_dpp %6, %7, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %7, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 9) {  // ds_bpermute-free LDS comparison: ds_read_b64 + ds_write_b64 pairs are measured elsewhere
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else {  // v_mov_b32
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                             "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads, float *d_out, unsigned long long *d_cyc)
{
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 2000, blocks = 256;  // ITERS=100000: kernels long enough for the clocks to ramp
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, d_cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
    const double instr_per_wave = (double)iters * 32;  // 32 wave-instructions per iteration in every mode
    const int waves_per_simd = threads / 256;
    printf("%-14s waves/SIMD=%d  cycles/instr/wave=%.2f  -> SIMD issue interval %.2f cycles/instr  (%.3f ms)\n", name,
           waves_per_simd ? waves_per_simd : 1, avg / instr_per_wave, avg / instr_per_wave / (waves_per_simd ? waves_per_simd : 1), ms);
}

int main()
{
    float *d_out; unsigned long long *d_cyc;
    hipMalloc(&d_out, 256 * 1024 * sizeof(float)); hipMalloc(&d_cyc, 256 * sizeof(unsigned long long));
    for (int threads : {512, 1024}) {
        run<0>("v_mul_f32", threads, d_out, d_cyc);
        run<1>("v_add_f32", threads, d_out, d_cyc);
        run<4>("v_fma_f32", threads, d_out, d_cyc);
        run<2>("v_pk_mul_f32", threads, d_out, d_cyc);
        run<3>("v_pk_add_f32", threads, d_out, d_cyc);
        run<5>("v_pk_fma_f32", threads, d_out, d_cyc);
        run<6>("v_mov_b32", threads, d_out, d_cyc);
        run<7>("mov_dpp wave_shr", threads, d_out, d_cyc);
        run<8>("sub_dpp wave_shl", threads, d_out, d_cyc);
        run<9>("mov_dpp row_shr", threads, d_out, d_cyc);
    }
    return 0;
}
