// micro-benchmark: issue cost of wave64 VALU instructions on gfx950 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) { // 4 independent v_pk_fma_f32 (8 flop-lanes)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                         "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));
        } else if (MODE == 2) { // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) { // 8 dependent v_fma_f32 (latency)
            asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                         "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                         : "+v"(a0) : "v"(s));
        } else if (MODE == 4) { // 8 independent v_mul_f32 + v_add (e32 encodings)
            asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 5) { // 8 independent dpp adds
            asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 6) { // 8 v_cndmask
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd; // 256 CUs x (4 waves per block = 1 wave per SIMD) x waves_per_simd
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 100, 1.0001f);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)iters * 8 * waves_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 indep", w); run<1>("v_pk_fma_f32 indep", w); run<2>("v_exp_f32 indep", w); run<3>("v_fma_f32 dependent", w);
        run<4>("v_mul/v_add e32", w); run<5>("v_add_f32_dpp", w); run<6>("v_cndmask", w);
    }
    return 0;
}
