// micro-benchmark 3: packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 results per lane per
// instruction on 64-bit register pairs) against the scalar forms -- cycles per wave64 instruction per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s)
{
    f2 a0 = {(float)threadIdx.x, 1.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    f2 m = {s, s};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile("v_pk_fma_f32 %0, %8, %1, %0\n v_pk_fma_f32 %1, %8, %2, %1\n v_pk_fma_f32 %2, %8, %3, %2\n v_pk_fma_f32 %3, %8, %4, %3\n"
                         "v_pk_fma_f32 %4, %8, %5, %4\n v_pk_fma_f32 %5, %8, %6, %5\n v_pk_fma_f32 %6, %8, %7, %6\n v_pk_fma_f32 %7, %8, %0, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 1) {
            asm volatile("v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n"
                         "v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 2) {
            asm volatile("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n"
                         "v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 3) { // scalar fma on the same 8 registers (reference point)
            float b0 = a0.x, b1 = a1.x, b2 = a2.x, b3 = a3.x, b4 = a4.x, b5 = a5.x, b6 = a6.x, b7 = a7.x;
            asm volatile("v_fma_f32 %0, %8, %1, %0\n v_fma_f32 %1, %8, %2, %1\n v_fma_f32 %2, %8, %3, %2\n v_fma_f32 %3, %8, %4, %3\n"
                         "v_fma_f32 %4, %8, %5, %4\n v_fma_f32 %5, %8, %6, %5\n v_fma_f32 %6, %8, %7, %6\n v_fma_f32 %7, %8, %0, %7\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(s));
            a0.x = b0; a1.x = b1; a2.x = b2; a3.x = b3; a4.x = b4; a5.x = b5; a6.x = b6; a7.x = b7;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a1.y + a2.y + a3.y + a4.y + a5.y + a6.y + a7.y;
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
    for (int w : {4, 8}) { run<0>("v_pk_fma_f32", w); run<1>("v_pk_mul_f32", w); run<2>("v_pk_add_f32", w); run<3>("v_fma_f32 (scalar reference)", w); }
}
