// micro-benchmark 4 (round 3): (a) packed f32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against v_fma_f32 -- does a packed
// instruction cost one issue slot (2 flops for 1) or two on gfx950?  (b) how many waves per SIMD a DEPENDENT fma chain needs to
// reach the issue rate (K7 runs at 5 waves / SIMD with long dependent chains per pair evaluation).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate4.hip -o tools/ubench/valu_rate4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}, ps = {s, s};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n"
                         "v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) { // 8 independent v_pk_fma_f32 (16 flops-pairs)
            asm volatile("v_pk_fma_f32 %0, %8, %0, %0\n v_pk_fma_f32 %1, %8, %1, %1\n v_pk_fma_f32 %2, %8, %2, %2\n v_pk_fma_f32 %3, %8, %3, %3\n"
                         "v_pk_fma_f32 %4, %8, %4, %4\n v_pk_fma_f32 %5, %8, %5, %5\n v_pk_fma_f32 %6, %8, %6, %6\n v_pk_fma_f32 %7, %8, %7, %7\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps));
        } else if (MODE == 2) { // v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n"
                         "v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps));
        } else if (MODE == 3) { // v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n"
                         "v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps));
        } else if (MODE == 4) { // ONE dependent chain of 8 v_fma_f32
            asm volatile("v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                         "v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                         : "+v"(a0) : "v"(s));
        } else if (MODE == 5) { // TWO interleaved dependent chains
            asm volatile("v_fma_f32 %0, %2, %0, %0\n v_fma_f32 %1, %2, %1, %1\n v_fma_f32 %0, %2, %0, %0\n v_fma_f32 %1, %2, %1, %1\n"
                         "v_fma_f32 %0, %2, %0, %0\n v_fma_f32 %1, %2, %1, %1\n v_fma_f32 %0, %2, %0, %0\n v_fma_f32 %1, %2, %1, %1\n"
                         : "+v"(a0), "+v"(a1) : "v"(s));
        } else if (MODE == 6) { // v_exp_f32, independent
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE> void run(const char* name, int w)
{
    int blocks = 1024 * w; float* out; hipMalloc(&out, (size_t)blocks * 64 * 4); int iters = 20000;   // 1024 SIMDs x w one-wave workgroups
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 64>>>(out, 100, 1.0001f); hipEventRecord(a); k<MODE><<<blocks, 64>>>(out, iters, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-40s waves/SIMD=%d  %.3f ms -> %.2f cyc/instr/SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * w));
    hipFree(out);
}
int main()
{
    for (int w : {4, 8}) { run<0>("v_fma_f32 x8 independent", w); run<1>("v_pk_fma_f32 x8 independent", w); run<2>("v_pk_mul_f32", w); run<3>("v_pk_add_f32", w); run<6>("v_exp_f32", w); }
    for (int w : {1, 2, 3, 4, 5, 6, 8}) { run<4>("v_fma_f32 one dependent chain", w); run<5>("v_fma_f32 two dependent chains", w); }
}
