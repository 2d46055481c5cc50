// micro-benchmark 3: does gfx950 skip the inactive 32-lane half of a wave64 VALU instruction?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned lane = threadIdx.x & 63;
    bool act = MODE == 0 ? true : MODE == 1 ? (lane < 32) : MODE == 2 ? (lane < 16) : MODE == 3 ? ((lane & 1) == 0) : (lane >= 32);
    if (act) {
        for (int i = 0; i < iters; i++) {
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
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
        run<0>("fma, all 64 lanes active", w); run<1>("fma, lanes 0-31 active", w); run<2>("fma, lanes 0-15 active", w);
        run<3>("fma, even lanes active", w); run<4>("fma, lanes 32-63 active", w);
    }
}
