// micro-benchmark 2: v_cndmask / v_cmp / min / max / sub / dpp costs (cycles per wave64 instruction per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) {
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                         "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "s20", "s21");
        } else if (MODE == 2) {
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) { // cmp + cndmask pairs (4 pairs)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
        } else if (MODE == 4) { // min/max
            asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 5) { // v_rcp
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 6) { // fma with multiply-by-mask style select: v = fma(m, x-v, v)
            asm volatile("v_fma_f32 %0, %8, %1, %0\n v_fma_f32 %1, %8, %2, %1\n v_fma_f32 %2, %8, %3, %2\n v_fma_f32 %3, %8, %4, %3\n"
                         "v_fma_f32 %4, %8, %5, %4\n v_fma_f32 %5, %8, %6, %5\n v_fma_f32 %6, %8, %7, %6\n v_fma_f32 %7, %8, %0, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char* name, int w)
{
    int blocks = 256 * w; float* out; hipMalloc(&out, (size_t)blocks * 256 * 4); int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 100, 1.0001f); hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, iters, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s waves/SIMD=%d  %.3f ms -> %.2f cyc/instr/SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * w));
    hipFree(out);
}
int main()
{
    for (int w : {4, 8}) {
        run<0>("v_cndmask_b32 e32 (vcc)", w); run<1>("v_cndmask_b32 e64 (sgpr pair)", w); run<2>("v_mov_b32", w);
        run<3>("v_cmp + v_cndmask pairs", w); run<4>("v_min/v_max", w); run<5>("v_rcp_f32", w); run<6>("v_fma_f32 (3 vgpr src)", w);
    }
}
