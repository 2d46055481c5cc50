// lg_loss.h -- fused photometric loss of the training step: L1 and SSIM (11x11 Gaussian window, sigma 1.5, zero padding)
// forward and backward in ONE pass over the image each.  Reference: utils/loss_utils.py:18-19 (l1_loss), :26-43 (window),
// :46-85 (ssim/_ssim), combined at prune_finetune.py:161-164 / distill_train.py:142-145 / train_densify_prune.py:135-138.
// The reference evaluates SSIM as five grouped 11x11 convolutions plus ~20 elementwise kernels (and their autograd
// mirrors); here the five windowed moments are formed by a separable 11+11 tap filter in one pass over both images, and the
// three per-pixel partial derivatives needed by the backward are written next to the partial sums.  The backward filters
// those three maps with the same (symmetric) window.
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

// Decomposition (round 4; the 32x32-tile kernels of rounds 1-3 ran 3x off their own instruction count -- 16-way LDS bank conflicts
// on the horizontally filtered planes, a third of the threads idle in the horizontal pass, 41 KB of LDS per workgroup):
// ONE WAVE streams a strip of 64 columns down LG_LOSS_TH output rows (+ 5 rows of halo at either end).  Per input row: the 74
// pixels the strip's window reaches go through an LDS row buffer (lane l reads its eleven neighbours l .. l + 10: consecutive
// lanes, consecutive words -- no bank conflict), the horizontal filter runs on them, and the vertical filter is a ring of eleven
// accumulators per quantity IN REGISTERS: input row i adds w[t] * h(i) to output row i - t, the row that received its eleventh
// term is finished and leaves.  No workgroup barrier, no LDS planes, every global access a coalesced row segment, loads of
// row i + 2 in flight while row i is filtered.  The quantities pair up -- (x, y), (xx, yy) -- so two thirds of the
// multiply-adds are v_pk_fma_f32.  Each output still sums its taps in ascending order from fma(w[0], v, 0), horizontally then
// vertically, exactly as the tile kernels did (the windowed moments are bit-identical to rounds 1-3; the three quotients of
// the SSIM map are v_rcp_f32 + a Newton step now, see lg_loss_rcp).
#ifndef LG_LOSS_TH
#define LG_LOSS_TH 34                    // output rows per wave: 44 input rows = four turns of the 11-row ring
#endif
#ifndef LG_LOSS_PF
#define LG_LOSS_PF 2                     // rows whose loads are in flight ahead of the row being filtered
#endif
#define LG_LOSS_HALO 5
#define LG_LOSS_STRIP 64
#define LG_LOSS_ROWBUF (LG_LOSS_STRIP + 2 * LG_LOSS_HALO + 6)   // 80 words per quantity

typedef float lg_v2f __attribute__((ext_vector_type(2)));

// float32 values of utils/loss_utils.py:26-33 gaussian(11, 1.5) (torch.Tensor of exp(..) / its sum); the 2-D window of
// :36-43 is their outer product, so the separable filter uses exactly the reference's weights
// (tests/test_golden_reference_python.py pins these eleven numbers against the reference function).
__device__ __constant__ float LG_SSIM_W[11] = {
    0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.10656p-2f,
    0x1.b43c3ep-3f,  0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d956cp-10f};

#define LG_SSIM_C1 (0.01f * 0.01f)
#define LG_SSIM_C2 (0.03f * 0.03f)

struct LossView {
    float* dmu1;      // [C*H*W] d ssim / d mu1 (total: through sigma1_sq and sigma12 as well)
    float* dsig1;     // [C*H*W] d ssim / d sigma1_sq
    float* dsig12;    // [C*H*W] d ssim / d sigma12
    float2* partials; // [waves] {sum |x-y|, sum ssim} of each forward wave (or L1 block)
    size_t total;
};
static inline int lg_loss_strips(int W) { return (W + LG_LOSS_STRIP - 1) / LG_LOSS_STRIP; }
static inline int lg_loss_segs(int H) { return (H + LG_LOSS_TH - 1) / LG_LOSS_TH; }
static LossView carve_loss(void* base, int C, int H, int W)
{
    LossView v; size_t off = 0; char* p = (char*)base; const size_t P = (size_t)C * H * W;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    const size_t waves = (size_t)lg_loss_strips(W) * lg_loss_segs(H) * C;
    v.dmu1 = (float*)take(P * 4);
    v.dsig1 = (float*)take(P * 4);
    v.dsig12 = (float*)take(P * 4);
    v.partials = (float2*)take(waves * 8);
    v.total = off;
    return v;
}

__device__ __forceinline__ lg_v2f lg_pk_fma(float w, lg_v2f a, lg_v2f c)
{
    const lg_v2f ww = {w, w};
    return __builtin_elementwise_fma(ww, a, c);
}
#define LG_LOSS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Loads of the row stream are UNCONDITIONAL (row and column clamped into the image, the value replaced by zero afterwards):
// a load inside a branch makes hipcc wait for every outstanding load (s_waitcnt vmcnt(0)) before the first use of any of them,
// which would put the full memory latency in front of every row; straight-line loads are counted, and rows i + 1 and i + 2 stay
// in flight while row i is filtered.  Lanes without a second pixel (lane >= 10) re-read their first one (same cache line).
struct LgLossCols {
    uint32_t colA, colB;     // clamped columns of this lane's two pixels of a row
    bool inA, inB;           // ... and whether they exist (conv2d(padding=5): zeros outside the image)
};
__device__ __forceinline__ LgLossCols lg_loss_cols(int cx0, int lane, int W)
{
    LgLossCols c;
    const int a = cx0 - LG_LOSS_HALO + lane, b = cx0 + LG_LOSS_STRIP - LG_LOSS_HALO + lane;
    c.inA = a >= 0 && a < W;
    c.inB = lane < 2 * LG_LOSS_HALO && b < W;
    c.colA = (uint32_t)min(max(a, 0), W - 1);
    c.colB = c.inB ? (uint32_t)b : c.colA;
    return c;
}

// 1 / b for b > 0 to about 1 ulp: v_rcp_f32 and one Newton step (the correctly rounded division of this library's build flags
// costs 12 instructions; the SSIM map needs three quotients per pixel, and its contract is 1e-4, not the last bit)
__device__ __forceinline__ float lg_loss_rcp(float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r, 1.0f), r, r);
}

// forward: grid (strips, segments, C), one wave per workgroup
__global__ void __launch_bounds__(LG_LOSS_STRIP)
lg_loss_fwd(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, float* __restrict__ dmu1,
            float* __restrict__ dsig1, float* __restrict__ dsig12, float2* __restrict__ partials)
{
    __shared__ lg_v2f rb[LG_LOSS_ROWBUF];          // {x, y} of the row being filtered: columns cx0 - 5 ... cx0 + 68
    const int lane = threadIdx.x;
    const int cx0 = blockIdx.x * LG_LOSS_STRIP, y0 = blockIdx.y * LG_LOSS_TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int nout = min(LG_LOSS_TH, H - y0);      // output rows of this wave
    const int nrows = nout + 2 * LG_LOSS_HALO;     // input rows y0 - 5 ... y0 + nout + 4
    const LgLossCols cc = lg_loss_cols(cx0, lane, W);
    const uint32_t ocol = (uint32_t)(cx0 + lane);
    const bool ocol_in = ocol < (uint32_t)W;
    float w[11];
#pragma unroll
    for (int t = 0; t < 11; t++) w[t] = LG_SSIM_W[t];

    // input row i of the segment (image row y0 - 5 + i): this lane's {x, y} at colA and -- lanes 0..9 -- at colB, as loaded (row
    // clamped); zeroed when the row is USED, two iterations later -- a select next to the load would wait for it on the spot
    auto load_row = [&](int i, lg_v2f& a, lg_v2f& b) {
        const int gy = y0 - LG_LOSS_HALO + i;
        const size_t r = plane + (size_t)min(max(gy, 0), H - 1) * W;
        const float* px = img + r;
        const float* py = gt + r;
        a.x = px[cc.colA]; a.y = py[cc.colA]; b.x = px[cc.colB]; b.y = py[cc.colB];
    };
    lg_v2f pa[LG_LOSS_PF + 1], pb[LG_LOSS_PF + 1];   // rows i ... i + PF of the stream
#pragma unroll
    for (int d = 0; d < LG_LOSS_PF; d++) load_row(d, pa[d], pb[d]);
    lg_v2f acc01[11], acc23[11];                   // ring of the vertical filter: {mu1, mu2}, {xx, yy} ...
    float acc4[11];                                //                              ... and xy
#pragma unroll
    for (int s = 0; s < 11; s++) { acc01[s] = lg_v2f{0.0f, 0.0f}; acc23[s] = lg_v2f{0.0f, 0.0f}; acc4[s] = 0.0f; }
    float l1 = 0.0f, ss = 0.0f;
    float* o1 = dmu1 + plane + (size_t)y0 * W;     // output row pointers (wave-uniform), advanced as rows leave
    float* o2 = dsig1 + plane + (size_t)y0 * W;
    float* o3 = dsig12 + plane + (size_t)y0 * W;
    for (int ib = 0; ib < nrows; ib += 11) {
#pragma unroll
        for (int j = 0; j < 11; j++) {
            const int i = ib + j;
            // (the last turn of the ring may run past nrows: those rows are clamped loads and arithmetic nobody reads -- a `break`
            //  here would keep hipcc from unrolling, and the ring slots must be compile-time register names)
            load_row(i + LG_LOSS_PF, pa[LG_LOSS_PF], pb[LG_LOSS_PF]);
            const lg_v2f a0 = pa[0], b0 = pb[0];
            LG_LOSS_SYNC();                        // (the previous row's reads are done)
            {
                const int gy = y0 - LG_LOSS_HALO + i;
                const bool rin = gy >= 0 && gy < H;                    // wave-uniform
                const bool ka = rin && cc.inA, kb = rin && cc.inB;
                rb[lane] = lg_v2f{ka ? a0.x : 0.0f, ka ? a0.y : 0.0f};
                if (lane < 2 * LG_LOSS_HALO) rb[LG_LOSS_STRIP + lane] = lg_v2f{kb ? b0.x : 0.0f, kb ? b0.y : 0.0f};
            }
            LG_LOSS_SYNC();
            // horizontal filter: taps in ascending order from fma(w[0], v, 0)
            lg_v2f h01 = {0.0f, 0.0f}, h23 = {0.0f, 0.0f};
            float h4 = 0.0f;
#pragma unroll
            for (int t = 0; t < 11; t++) {
                const lg_v2f p = rb[lane + t];
                const lg_v2f sq = p * p;
                const float xy = p.x * p.y;
                h01 = lg_pk_fma(w[t], p, h01);
                h23 = lg_pk_fma(w[t], sq, h23);
                h4 = fmaf(w[t], xy, h4);
                if (t == LG_LOSS_HALO) {           // the centre tap is this lane's own pixel
                    const float ad = fabsf(p.x - p.y);
                    l1 += (i >= LG_LOSS_HALO && i < LG_LOSS_HALO + nout && ocol_in) ? ad : 0.0f;
                }
            }
            // vertical filter: this row is tap t of output row i - t (ring slot (i - t) mod 11; i = j mod 11 here)
#pragma unroll
            for (int t = 0; t < 11; t++) {
                const int s = (j - t + 11) % 11;
                if (t == 0) { acc01[s] = lg_pk_fma(w[0], h01, lg_v2f{0.0f, 0.0f}); acc23[s] = lg_pk_fma(w[0], h23, lg_v2f{0.0f, 0.0f}); acc4[s] = fmaf(w[0], h4, 0.0f); }
                else { acc01[s] = lg_pk_fma(w[t], h01, acc01[s]); acc23[s] = lg_pk_fma(w[t], h23, acc23[s]); acc4[s] = fmaf(w[t], h4, acc4[s]); }
            }
            // output row i - 10 has all eleven terms
            if (i >= 2 * LG_LOSS_HALO && i < nrows) {   // wave-uniform
                const int s = (j + 1) % 11;
                const float mu1 = acc01[s].x, mu2 = acc01[s].y;
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
                const float sigma1_sq = acc23[s].x - mu1_sq, sigma2_sq = acc23[s].y - mu2_sq, sigma12 = acc4[s] - mu1_mu2;
                const float A1 = 2.0f * mu1_mu2 + LG_SSIM_C1, A2 = 2.0f * sigma12 + LG_SSIM_C2;
                const float B1 = mu1_sq + mu2_sq + LG_SSIM_C1, B2 = sigma1_sq + sigma2_sq + LG_SSIM_C2;   // >= C1, >= C2 (up to rounding): never 0
                const float iB1 = lg_loss_rcp(B1), iB2 = lg_loss_rcp(B2);
                const float inv = iB1 * iB2;
                const float S = A1 * A2 * inv;
                // partial derivatives of S
                const float dS_dsig12 = 2.0f * A1 * inv;
                const float dS_dsig1 = -S * iB2;
                const float dS_dmu1_explicit = 2.0f * mu2 * A2 * inv - 2.0f * mu1 * S * iB1;
                const float dS_dmu1 = dS_dmu1_explicit - 2.0f * mu1 * dS_dsig1 - mu2 * dS_dsig12;
                if (ocol_in) { o1[ocol] = dS_dmu1; o2[ocol] = dS_dsig1; o3[ocol] = dS_dsig12; ss += S; }
                o1 += W; o2 += W; o3 += W;
            }
#pragma unroll
            for (int d = 0; d < LG_LOSS_PF; d++) { pa[d] = pa[d + 1]; pb[d] = pb[d + 1]; }
        }
    }
    // wave sums in a fixed order (deterministic)
    l1 = wave_sum_to_lane63(l1);
    ss = wave_sum_to_lane63(ss);
    if (lane == 63) partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float2(l1, ss);
}

// out[0] = mean |x-y|, out[1] = mean ssim_map; one block, fixed summation order, double accumulation
__global__ void __launch_bounds__(256)
lg_loss_finalize(int nblocks, double inv_count, const float2* __restrict__ partials, float* __restrict__ out)
{
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { const float2 p = partials[i]; a += (double)p.x; b += (double)p.y; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s0[threadIdx.x] += s0[threadIdx.x + s]; s1[threadIdx.x] += s1[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)(s0[0] * inv_count); out[1] = (float)(s1[0] * inv_count); }
}

// backward: dL/dimg = g_l1 * sign(x - y) / n  +  g_ssim / n * [ G*dmu1 + 2 x G*dsig1 + y G*dsig12 ]
// (G symmetric, maps are zero outside the image).  g_l1 = scale_l1 * *dL_dl1, g_ssim = scale_ssim * *dL_dssim.
// Same decomposition as the forward: a wave per (strip of 64 columns, LG_LOSS_TH rows), the three maps through the LDS row
// buffer, the vertical filter in a register ring; x and y of an output row are requested two rows before the row leaves.
__global__ void __launch_bounds__(LG_LOSS_STRIP)
lg_loss_bwd(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ dmu1,
            const float* __restrict__ dsig1, const float* __restrict__ dsig12, const float* __restrict__ dL_dl1, float scale_l1,
            const float* __restrict__ dL_dssim, float scale_ssim, float inv_count, float* __restrict__ dL_dimg)
{
    __shared__ float rb[3][LG_LOSS_ROWBUF];
    const int lane = threadIdx.x;
    const int cx0 = blockIdx.x * LG_LOSS_STRIP, y0 = blockIdx.y * LG_LOSS_TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int nout = min(LG_LOSS_TH, H - y0);
    const int nrows = nout + 2 * LG_LOSS_HALO;
    const LgLossCols cc = lg_loss_cols(cx0, lane, W);
    const bool ocol_in = (uint32_t)(cx0 + lane) < (uint32_t)W;
    const uint32_t ocol = (uint32_t)min(cx0 + lane, W - 1);            // clamped: the loads of x, y below are unconditional
    const float g_l1 = (dL_dl1 ? dL_dl1[0] * scale_l1 : 0.0f) * inv_count;
    const float g_ss = (dL_dssim ? dL_dssim[0] * scale_ssim : 0.0f) * inv_count;
    float w[11];
#pragma unroll
    for (int t = 0; t < 11; t++) w[t] = LG_SSIM_W[t];

    struct Row { float a[3], b[3], x, y; };        // the three maps at colA / colB of input row i; x, y of output row i - 10
    auto load_row = [&](int i, Row& r) {           // as loaded (row clamped); zeroed when the row is used (see lg_loss_fwd)
        const int gy = y0 - LG_LOSS_HALO + i;
        const size_t p = plane + (size_t)min(max(gy, 0), H - 1) * W;
        const float* m0 = dmu1 + p; const float* m1 = dsig1 + p; const float* m2 = dsig12 + p;
        r.a[0] = m0[cc.colA]; r.a[1] = m1[cc.colA]; r.a[2] = m2[cc.colA];
        r.b[0] = m0[cc.colB]; r.b[1] = m1[cc.colB]; r.b[2] = m2[cc.colB];
        // x, y of the output row that leaves when input row i is filtered (row y0 + i - 10; clamped while there is none)
        const size_t po = plane + (size_t)min(max(y0 + i - 2 * LG_LOSS_HALO, 0), H - 1) * W;
        r.x = (img + po)[ocol]; r.y = (gt + po)[ocol];
    };
    Row pr[LG_LOSS_PF + 1];
#pragma unroll
    for (int d = 0; d < LG_LOSS_PF; d++) load_row(d, pr[d]);
    float acc[3][11];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int s = 0; s < 11; s++) acc[q][s] = 0.0f;
    float* out = dL_dimg + plane + (size_t)y0 * W;
    for (int ib = 0; ib < nrows; ib += 11) {
#pragma unroll
        for (int j = 0; j < 11; j++) {
            const int i = ib + j;
            load_row(i + LG_LOSS_PF, pr[LG_LOSS_PF]);   // (rows past nrows in the last turn of the ring: see lg_loss_fwd)
            const Row r0 = pr[0];
            LG_LOSS_SYNC();
            {
                const int gy = y0 - LG_LOSS_HALO + i;
                const bool rin = gy >= 0 && gy < H;                    // wave-uniform
                const bool ka = rin && cc.inA, kb = rin && cc.inB;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    rb[q][lane] = ka ? r0.a[q] : 0.0f;
                    if (lane < 2 * LG_LOSS_HALO) rb[q][LG_LOSS_STRIP + lane] = kb ? r0.b[q] : 0.0f;
                }
            }
            LG_LOSS_SYNC();
            float h[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 11; t++)
#pragma unroll
                for (int q = 0; q < 3; q++) h[q] = fmaf(w[t], rb[q][lane + t], h[q]);
#pragma unroll
            for (int t = 0; t < 11; t++) {
                const int s = (j - t + 11) % 11;
#pragma unroll
                for (int q = 0; q < 3; q++) acc[q][s] = (t == 0) ? fmaf(w[0], h[q], 0.0f) : fmaf(w[t], h[q], acc[q][s]);
            }
            if (i >= 2 * LG_LOSS_HALO && i < nrows) {   // wave-uniform: output row i - 10 has all eleven terms
                const int s = (j + 1) % 11;
                const float x = r0.x, y = r0.y;
                const float d = x - y;
                const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); // torch: grad of abs at 0 is 0
                if (ocol_in) out[ocol] = fmaf(g_l1, sgn, g_ss * (acc[0][s] + 2.0f * x * acc[1][s] + y * acc[2][s]));
                out += W;
            }
#pragma unroll
            for (int d = 0; d < LG_LOSS_PF; d++) pr[d] = pr[d + 1];
        }
    }
}

// L1 alone (LG_FLAG_L1_ONLY): mean |x - y| without the windowed moments, for callers that do not use SSIM.
// Same partials / finalize as the fused kernel (ssim partial = 0); 4 elements per thread, float4 when aligned.
__global__ void __launch_bounds__(256)
lg_l1_fwd(size_t n, const float* __restrict__ img, const float* __restrict__ gt, float2* __restrict__ partials)
{
    __shared__ float wsum[4];
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += fabsf(img[i] - gt[i]);
    acc = wave_sum_to_lane63(acc);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = make_float2((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]), 0.0f);
}
__global__ void __launch_bounds__(256)
lg_l1_bwd(size_t n, const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ dL_dl1, float scale,
          float* __restrict__ dL_dimg)
{
    const float g = dL_dl1 ? dL_dl1[0] * scale : 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = img[i] - gt[i];
        dL_dimg[i] = d > 0.0f ? g : (d < 0.0f ? -g : 0.0f);
    }
}
