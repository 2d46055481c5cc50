// lg_loss.h -- fused photometric loss of the training step: L1 and SSIM (11x11 Gaussian window, sigma 1.5, zero padding)
// forward and backward in ONE pass over the image each.  Reference: utils/loss_utils.py:18-19 (l1_loss), :26-43 (window),
// :46-85 (ssim/_ssim), combined at prune_finetune.py:161-164 / distill_train.py:142-145 / train_densify_prune.py:135-138.
// The reference evaluates SSIM as five grouped 11x11 convolutions plus ~20 elementwise kernels (and their autograd
// mirrors); here a 32x32 pixel tile (+5 halo) of both images is staged in LDS once, the five windowed moments are formed
// by a separable 11+11 tap filter, and the three per-pixel partial derivatives needed by the backward are written next to
// the two block partial sums.  The backward filters those three maps with the same (symmetric) window.
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

#define LG_LOSS_TILE 32
#define LG_LOSS_HALO 5
#define LG_LOSS_EXT (LG_LOSS_TILE + 2 * LG_LOSS_HALO) // 42
#define LG_LOSS_PITCH (LG_LOSS_EXT + 1)

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
    float2* partials; // [blocks] {sum |x-y|, sum ssim} of each forward block
    size_t total;
};
static LossView carve_loss(void* base, int C, int H, int W)
{
    LossView v; size_t off = 0; char* p = (char*)base; const size_t P = (size_t)C * H * W;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    const size_t blocks = (size_t)((W + LG_LOSS_TILE - 1) / LG_LOSS_TILE) * ((H + LG_LOSS_TILE - 1) / LG_LOSS_TILE) * C;
    v.dmu1 = (float*)take(P * 4);
    v.dsig1 = (float*)take(P * 4);
    v.dsig12 = (float*)take(P * 4);
    v.partials = (float2*)take(blocks * 8);
    v.total = off;
    return v;
}

// Vertical 11-tap filter of NQ quantities for 4 consecutive rows of one column, from the horizontally filtered LDS
// planes h[q][EXT rows][TILE cols].
template <int NQ>
__device__ __forceinline__ void vfilter4(const float (*h)[LG_LOSS_EXT][LG_LOSS_TILE], int col, int row0, float out[NQ][4])
{
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; k++) v[k] = h[q][row0 + k][col];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < 11; t++) acc = fmaf(LG_SSIM_W[t], v[o + t], acc);
            out[q][o] = acc;
        }
    }
}

// forward: grid (ceil(W/32), ceil(H/32), C), block 256
__global__ void __launch_bounds__(256)
lg_loss_fwd(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, float* __restrict__ dmu1,
            float* __restrict__ dsig1, float* __restrict__ dsig12, float2* __restrict__ partials)
{
    __shared__ float sx[LG_LOSS_EXT][LG_LOSS_PITCH], sy[LG_LOSS_EXT][LG_LOSS_PITCH];
    __shared__ float h[5][LG_LOSS_EXT][LG_LOSS_TILE];
    __shared__ float2 wsum[4];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LG_LOSS_TILE, y0 = blockIdx.y * LG_LOSS_TILE;
    const size_t plane = (size_t)blockIdx.z * H * W;

    for (int i = tid; i < LG_LOSS_EXT * LG_LOSS_EXT; i += 256) {
        const int r = i / LG_LOSS_EXT, c = i - r * LG_LOSS_EXT;
        const int gy = y0 + r - LG_LOSS_HALO, gx = x0 + c - LG_LOSS_HALO;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H; // conv2d(padding=5): zeros outside
        const size_t a = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
        sx[r][c] = in ? img[a] : 0.0f;
        sy[r][c] = in ? gt[a] : 0.0f;
    }
    __syncthreads();

    // horizontal pass: 42 rows x 4 segments of 8 outputs; x, y, xx, yy, xy
    for (int item = tid; item < LG_LOSS_EXT * 4; item += 256) {
        const int r = item >> 2, c0 = (item & 3) * 8;
        float a[5][8];
#pragma unroll
        for (int q = 0; q < 5; q++)
#pragma unroll
            for (int o = 0; o < 8; o++) a[q][o] = 0.0f;
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const float x = sx[r][c0 + k], y = sy[r][c0 + k];
            const float xx = x * x, yy = y * y, xy = x * y;
#pragma unroll
            for (int o = 0; o < 8; o++) {
                const int t = k - o;
                if (t >= 0 && t < 11) {
                    const float w = LG_SSIM_W[t];
                    a[0][o] = fmaf(w, x, a[0][o]); a[1][o] = fmaf(w, y, a[1][o]); a[2][o] = fmaf(w, xx, a[2][o]);
                    a[3][o] = fmaf(w, yy, a[3][o]); a[4][o] = fmaf(w, xy, a[4][o]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 5; q++)
#pragma unroll
            for (int o = 0; o < 8; o++) h[q][r][c0 + o] = a[q][o];
    }
    __syncthreads();

    // vertical pass + SSIM: thread = (column, 4 consecutive rows)
    const int col = tid & 31, row0 = (tid >> 5) * 4;
    float m[5][4];
    vfilter4<5>(h, col, row0, m);
    float l1 = 0.0f, ss = 0.0f;
    const int gx = x0 + col;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int gy = y0 + row0 + o;
        if (gx < W && gy < H) {
            const float mu1 = m[0][o], mu2 = m[1][o];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
            const float sigma1_sq = m[2][o] - mu1_sq, sigma2_sq = m[3][o] - mu2_sq, sigma12 = m[4][o] - mu1_mu2;
            const float A1 = 2.0f * mu1_mu2 + LG_SSIM_C1, A2 = 2.0f * sigma12 + LG_SSIM_C2;
            const float B1 = mu1_sq + mu2_sq + LG_SSIM_C1, B2 = sigma1_sq + sigma2_sq + LG_SSIM_C2;
            const float inv = 1.0f / (B1 * B2);
            const float S = A1 * A2 * inv;
            // partial derivatives of S
            const float dS_dsig12 = 2.0f * A1 * inv;
            const float dS_dsig1 = -S / B2;
            const float dS_dmu1_explicit = 2.0f * mu2 * A2 * inv - 2.0f * mu1 * S / B1;
            const float dS_dmu1 = dS_dmu1_explicit - 2.0f * mu1 * dS_dsig1 - mu2 * dS_dsig12;
            const size_t a = plane + (size_t)gy * W + gx;
            dmu1[a] = dS_dmu1; dsig1[a] = dS_dsig1; dsig12[a] = dS_dsig12;
            ss += S;
            l1 += fabsf(sx[row0 + o + LG_LOSS_HALO][col + LG_LOSS_HALO] - sy[row0 + o + LG_LOSS_HALO][col + LG_LOSS_HALO]);
        }
    }
    // block sums in a fixed order (deterministic): wave reduce, then 4 waves
    l1 = wave_sum_to_lane63(l1);
    ss = wave_sum_to_lane63(ss);
    if ((tid & 63) == 63) wsum[tid >> 6] = make_float2(l1, ss);
    __syncthreads();
    if (tid == 0) {
        const float2 p0 = wsum[0], p1 = wsum[1], p2 = wsum[2], p3 = wsum[3];
        partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
            make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
    }
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
__global__ void __launch_bounds__(256)
lg_loss_bwd(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ dmu1,
            const float* __restrict__ dsig1, const float* __restrict__ dsig12, const float* __restrict__ dL_dl1, float scale_l1,
            const float* __restrict__ dL_dssim, float scale_ssim, float inv_count, float* __restrict__ dL_dimg)
{
    __shared__ float sm[3][LG_LOSS_EXT][LG_LOSS_PITCH];
    __shared__ float h[3][LG_LOSS_EXT][LG_LOSS_TILE];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LG_LOSS_TILE, y0 = blockIdx.y * LG_LOSS_TILE;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float g_l1 = (dL_dl1 ? dL_dl1[0] * scale_l1 : 0.0f) * inv_count;
    const float g_ss = (dL_dssim ? dL_dssim[0] * scale_ssim : 0.0f) * inv_count;

    for (int i = tid; i < LG_LOSS_EXT * LG_LOSS_EXT; i += 256) {
        const int r = i / LG_LOSS_EXT, c = i - r * LG_LOSS_EXT;
        const int gy = y0 + r - LG_LOSS_HALO, gx = x0 + c - LG_LOSS_HALO;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t a = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
        sm[0][r][c] = in ? dmu1[a] : 0.0f;
        sm[1][r][c] = in ? dsig1[a] : 0.0f;
        sm[2][r][c] = in ? dsig12[a] : 0.0f;
    }
    __syncthreads();
    for (int item = tid; item < LG_LOSS_EXT * 4; item += 256) {
        const int r = item >> 2, c0 = (item & 3) * 8;
        float a[3][8];
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int o = 0; o < 8; o++) a[q][o] = 0.0f;
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const float v0 = sm[0][r][c0 + k], v1 = sm[1][r][c0 + k], v2 = sm[2][r][c0 + k];
#pragma unroll
            for (int o = 0; o < 8; o++) {
                const int t = k - o;
                if (t >= 0 && t < 11) {
                    const float w = LG_SSIM_W[t];
                    a[0][o] = fmaf(w, v0, a[0][o]); a[1][o] = fmaf(w, v1, a[1][o]); a[2][o] = fmaf(w, v2, a[2][o]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int o = 0; o < 8; o++) h[q][r][c0 + o] = a[q][o];
    }
    __syncthreads();
    const int col = tid & 31, row0 = (tid >> 5) * 4;
    float m[3][4];
    vfilter4<3>(h, col, row0, m);
    const int gx = x0 + col;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int gy = y0 + row0 + o;
        if (gx < W && gy < H) {
            const size_t a = plane + (size_t)gy * W + gx;
            const float x = img[a], y = gt[a];
            const float d = x - y;
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); // torch: grad of abs at 0 is 0
            dL_dimg[a] = fmaf(g_l1, sgn, g_ss * (m[0][o] + 2.0f * x * m[1][o] + y * m[2][o]));
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
