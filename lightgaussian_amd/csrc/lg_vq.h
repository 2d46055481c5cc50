// lg_vq.h -- nearest-code search of the VecTree vector quantiser (second half) on the matrix cores.
// Replaces vectree/vq.py:262-266 (EuclideanCodebook.forward: dist = -torch.cdist(flatten, embed, p=2);
// embed_ind = dist.argmax(-1), i.e. gumbel_sample at temperature 0), driven by vectree/vectree.py:87-101 in chunks of 8192
// feature rows against the 8192 x 27 (SH degree 2) or 8192 x 48 (degree 3) codebook -- the one GEMM-shaped piece of the
// reference and the consumer of this path's imp_score.npz.
//
//   argmin_c |x - c|^2 = argmin_c (|c|^2 - 2 x.c): the code side is augmented once per codebook (lg_vq_prepare) to
//   A'[c] = (-2 c_0 .. -2 c_{d-1}, |c|^2, 0..) and the point side to B'[p] = (x_0 .. x_{d-1}, 1, 0..), so one f32 MFMA chain
//   v_mfma_f32_32x32x2_f32 over the d+1 augmented dimensions leaves the score of 32 codes x 32 points in the accumulators
//   (exact f32: bitwise a k-ordered fmaf chain, MI355X guide -- there is no reduced-precision shortcut to take and none is
//   wanted: indices must match).  Codes are the ROWS of the MFMA tile: a lane then holds 16 codes for ONE point
//   (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) and keeps a running (min, index) in two registers --
//   no cross-lane work until the single lane <-> lane + 32 merge at the end.  Ties go to the lowest code index.
//   A workgroup = 4 waves x 32 points; the augmented codebook streams through LDS in chunks of 128 codes, double-buffered
//   through registers; rows are padded to an odd stride so the per-lane ds_read_b32 of the A operand is conflict-free.
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

#define LG_VQ_CHUNK 128               // codes per LDS chunk (4 MFMA row tiles)
typedef float lg_f16v __attribute__((ext_vector_type(16)));

// smallest supported number of k-pairs >= the needed one (augmented dimension d + 1, two k per MFMA); 0 = unsupported
static inline int lg_vq_dk2(int d)
{
    const int need = (d + 2) / 2;
    const int sizes[] = {2, 4, 7, 8, 14, 16, 25, 32};
    for (int s : sizes) if (s >= need) return s;
    return 0;
}
static inline int lg_vq_kpad(int K) { return (K + LG_VQ_CHUNK - 1) / LG_VQ_CHUNK * LG_VQ_CHUNK; }

// A'[c][0..d) = -2 c, A'[c][d] = |c|^2 (sequential fmaf chain), rest 0; rows K..Kpad never win (|c|^2 = FLT_MAX)
__global__ void __launch_bounds__(256)
lg_vq_prepare(int K, int Kpad, int d, int dk, const float* __restrict__ cb, float* __restrict__ cbA)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Kpad) return;
    float nrm = 0.0f;
    for (int k = 0; k < dk; k++) {
        float v = 0.0f;
        if (c < K && k < d) {
            const float w = cb[(size_t)c * d + k];
            nrm = fmaf(w, w, nrm);
            v = -2.0f * w;
        }
        cbA[(size_t)c * dk + k] = v;
    }
    cbA[(size_t)c * dk + d] = c < K ? nrm : 3.402823466e38f;
}

template <int DK2>
__global__ void __launch_bounds__(256)
lg_vq_nearest_kernel(int n, int d, int Kpad, const float* __restrict__ x, const float* __restrict__ cbA, int32_t* __restrict__ out)
{
    constexpr int DK = 2 * DK2, S = DK + 1;                    // LDS row stride: odd => bank-conflict-free column reads
    __shared__ float buf[2][LG_VQ_CHUNK * S];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int p = blockIdx.x * 128 + (int)wave * 32 + (int)(lane & 31u);   // this lane's point (column of the MFMA tile)
    const int kh = (int)(lane >> 5);                            // which of the two k of an MFMA step this lane feeds
    float xb[DK2];
#pragma unroll
    for (int kk = 0; kk < DK2; kk++) {
        const int k = 2 * kk + kh;
        xb[kk] = (p < n && k < d) ? x[(size_t)p * d + k] : (k == d ? 1.0f : 0.0f);
    }
    float stage[DK2];                                           // LG_VQ_CHUNK * DK / 256 = DK2 floats per thread
    auto fetch = [&](int chunk) {
        const float* src = cbA + (size_t)chunk * LG_VQ_CHUNK * DK;
#pragma unroll
        for (int j = 0; j < DK2; j++) stage[j] = src[j * 256 + (int)tid];
    };
    auto park = [&](float* dst) {
#pragma unroll
        for (int j = 0; j < DK2; j++) {
            const int idx = j * 256 + (int)tid;
            dst[(idx / DK) * S + (idx % DK)] = stage[j];
        }
    };
    const int nchunks = Kpad / LG_VQ_CHUNK;
    fetch(0);
    park(buf[0]);
    __syncthreads();
    float best = 3.402823466e38f;
    int32_t bidx = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        if (ch + 1 < nchunks) fetch(ch + 1);                    // global loads fly while the matrix cores work
        const float* cur = buf[ch & 1];
#pragma unroll 1
        for (int tile = 0; tile < LG_VQ_CHUNK / 32; tile++) {
            lg_f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const float* arow = cur + (tile * 32 + (int)(lane & 31u)) * S + kh;
#pragma unroll
            for (int kk = 0; kk < DK2; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * kk], xb[kk], acc, 0, 0, 0);
            const int code0 = ch * LG_VQ_CHUNK + tile * 32 + 4 * kh;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = acc[r];
                const bool lt = v < best;                       // strict: rows are visited in increasing code order
                best = lt ? v : best;
                bidx = lt ? code0 + (r & 3) + 8 * (r >> 2) : bidx;
            }
        }
        if (ch + 1 < nchunks) park(buf[(ch + 1) & 1]);
        __syncthreads();
    }
    const float ov = __shfl_xor(best, 32, 64);
    const int32_t oi = __shfl_xor(bidx, 32, 64);
    if (ov < best || (ov == best && oi < bidx)) bidx = oi;
    if (lane < 32u && p < n) out[p] = bidx;
}
