// micro-benchmark 5 (round 6): issue cost of the instructions of the Q24.40 row sums (lg_wq_rowsum): v_cvt_f64_f32, v_add_f64, v_lshl_add_u64,
// v_add_u32 (+ dpp), v_cvt_u32_f32, v_fract_f32, v_rndne_f32, against v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate5.hip -o tools/ubench/valu_rate5
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float s)
{
    float a[8]; double d[8]; unsigned long long u[8]; unsigned int w[8];
    for (int q = 0; q < 8; q++) { a[q] = threadIdx.x + q; d[q] = a[q]; u[q] = threadIdx.x * 77 + q; w[q] = threadIdx.x + q; }
    const double ds = s;
    for (int i = 0; i < iters; i++) {
#define FMA(q) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[q]) : "v"(s));
#define CVT(q) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[q]) : "v"(a[q]));
#define ADD64(q) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[q]) : "v"(ds));
#define LSHLADD(q) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(u[q]) : "v"(u[(q + 1) & 7]));
#define ADDU(q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[q]) : "v"(w[(q + 1) & 7]));
#define ADDDPP(q) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(w[q]));
#define CVTU(q) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(w[q]) : "v"(a[q]));
#define FRACT(q) asm volatile("v_fract_f32 %0, %0" : "+v"(a[q]));
#define RNDNE(q) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[q]));
#define CVTF32(q) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[q]) : "v"(d[q]));
#define FMA64(q) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(d[q]) : "v"(ds));
        if (MODE == 0) { R8(FMA) }
        if (MODE == 1) { R8(CVT) }
        if (MODE == 2) { R8(ADD64) }
        if (MODE == 3) { R8(LSHLADD) }
        if (MODE == 4) { R8(ADDU) }
        if (MODE == 5) { R8(ADDDPP) }
        if (MODE == 6) { R8(CVTU) }
        if (MODE == 7) { R8(FRACT) }
        if (MODE == 8) { R8(RNDNE) }
        if (MODE == 9) { R8(CVTF32) }
        if (MODE == 10) { R8(FMA64) }
    }
    float r = 0;
    for (int q = 0; q < 8; q++) r += a[q] + (float)d[q] + (float)u[q] + (float)w[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, int w)
{
    int blocks = 1024 * w; float* out; hipMalloc(&out, (size_t)blocks * 64 * 4); int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 64>>>(out, 100, 1.0001f); hipEventRecord(a); k<MODE><<<blocks, 64>>>(out, iters, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s waves/SIMD=%d  %.3f ms -> %.2f cyc/instr/SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * w));
    hipFree(out);
}
int main()
{
    for (int w : {4, 8}) {
        run<0>("v_fma_f32", w); run<1>("v_cvt_f64_f32", w); run<2>("v_add_f64", w); run<10>("v_fma_f64", w); run<3>("v_lshl_add_u64", w); run<4>("v_add_u32", w); run<5>("v_add_u32_dpp", w);
        run<6>("v_cvt_u32_f32", w); run<7>("v_fract_f32", w); run<8>("v_rndne_f32", w); run<9>("v_cvt_f32_f64", w);
    }
}
