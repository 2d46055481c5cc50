// lg_prune.h -- device-resident prune epilogue: the step right after the significance pass.
//   v_list = (volume / kth_volume)^v_pow * imp_list        prune.py:112-128 (calculate_v_imp_score)
//   mask   = v_list <= sorted(v_list)[int(p * (N - 1))]    scene/gaussian_model.py:776-782 (prune_gaussians)
// The reference does two full sorts of N floats and reads two elements back through host indexing.  Both order
// statistics are radix SELECTS here: four 8-bit histogram passes over the keys, every pass multi-block, the winning
// bucket of the earlier passes re-derived by every block from the finished histograms (a 256-bin scan by one wave) --
// so there is no host round trip and no single-block serial tail anywhere.  Exact: a select returns the same element a
// sort would put at that index.
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

// float -> unsigned key with the same total order (negative floats reversed, sign flipped); -0 < +0, NaNs last/first by sign
__device__ __forceinline__ uint32_t lg_order_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float lg_order_key_inv(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Select state of ONE order statistic: hist[pass][256], zeroed before the first pass.
struct LgSelect { uint32_t hist[4][256]; };

// Resolve the first `npass` digits of the rank-th smallest key (0-based) from the finished histograms.
// Called by all threads of a block; thread-uniform result via LDS.  Returns the key prefix (digits in the top bits).
__device__ __forceinline__ uint32_t lg_select_resolve(const LgSelect* __restrict__ st, uint32_t rank, int npass, uint32_t* lds /*[2]*/)
{
    if (threadIdx.x < 64) {
        uint32_t prefix = 0, r = rank;
        for (int p = 0; p < npass; p++) {
            // lane l owns bins 4l..4l+3
            const uint32_t l = threadIdx.x;
            const uint4 h = *reinterpret_cast<const uint4*>(&st->hist[p][4 * l]);
            const uint32_t mine = h.x + h.y + h.z + h.w;
            // inclusive scan across the wave (DPP-free, 6 steps of shuffles: runs once per block per pass)
            uint32_t inc = mine;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if ((int)l >= d) inc += o;
            }
            const uint32_t exc = inc - mine;
            const bool owner = r >= exc && r < inc;     // exactly one lane (total count > rank by construction)
            uint32_t digit = 0, below = 0;
            if (owner) {
                uint32_t c = exc;
                if (r < c + h.x) { digit = 4 * l; below = c; }
                else if (r < (c += h.x) + h.y) { digit = 4 * l + 1; below = c; }
                else if (r < (c += h.y) + h.z) { digit = 4 * l + 2; below = c; }
                else { c += h.z; digit = 4 * l + 3; below = c; }
            }
            const unsigned long long who = __ballot(owner);
            const int src = who ? (int)__builtin_ctzll(who) : 0;
            digit = __shfl(digit, src, 64);
            below = __shfl(below, src, 64);
            prefix |= digit << (24 - 8 * p);
            r -= below;
        }
        if (threadIdx.x == 0) { lds[0] = prefix; lds[1] = r; }
    }
    __syncthreads();
    return lds[0];
}

// volume of a Gaussian from its ACTIVATED scaling row: torch.prod(get_scaling, dim=1) = (s0 * s1) * s2
__device__ __forceinline__ float lg_volume(const float* __restrict__ scaling, int i)
{
    return (scaling[3 * (size_t)i] * scaling[3 * (size_t)i + 1]) * scaling[3 * (size_t)i + 2];
}

// One histogram pass of a select.  SRC 0: keys = volumes computed from scaling; SRC 1: keys = values[].
// Elements whose already-resolved digits differ from the prefix are ignored.
template <int SRC>
__global__ void __launch_bounds__(256)
lg_select_pass(int N, int pass, uint32_t rank, const float* __restrict__ src, LgSelect* __restrict__ st)
{
    __shared__ uint32_t lh[256];
    __shared__ uint32_t res[2];
    lh[threadIdx.x] = 0;
    const uint32_t prefix = lg_select_resolve(st, rank, pass, res); // syncs
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (32 - 8 * pass));
    const int shift = 24 - 8 * pass;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const uint32_t k = lg_order_key(SRC == 0 ? lg_volume(src, i) : src[i]);
        if ((k & mask) == prefix) atomicAdd(&lh[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    const uint32_t c = lh[threadIdx.x];
    if (c) atomicAdd(&st->hist[pass][threadIdx.x], c);
}

// v_list = pow(volume / kth, v_pow) * imp, plus the first histogram pass of the second select (fused)
__global__ void __launch_bounds__(256)
lg_v_imp_score_kernel(int N, uint32_t rank_volume, const float* __restrict__ scaling, const float* __restrict__ imp, float v_pow,
                      const LgSelect* __restrict__ st_volume, float* __restrict__ v_list, LgSelect* __restrict__ st_score,
                      float* __restrict__ thresholds)
{
    __shared__ uint32_t lh[256];
    __shared__ uint32_t res[2];
    lh[threadIdx.x] = 0;
    const float kth = lg_order_key_inv(lg_select_resolve(st_volume, rank_volume, 4, res));
    if (blockIdx.x == 0 && threadIdx.x == 0) thresholds[0] = kth;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const float v = powf(lg_volume(scaling, i) / kth, v_pow) * imp[i];
        v_list[i] = v;
        atomicAdd(&lh[lg_order_key(v) >> 24], 1u);
    }
    __syncthreads();
    const uint32_t c = lh[threadIdx.x];
    if (c) atomicAdd(&st_score->hist[0][threadIdx.x], c);
}

// mask[i] = v_list[i] <= threshold  (ties at the threshold are pruned, scene/gaussian_model.py:778-781)
__global__ void __launch_bounds__(256)
lg_prune_mask_kernel(int N, uint32_t rank_score, const float* __restrict__ v_list, const LgSelect* __restrict__ st_score,
                     uint8_t* __restrict__ mask, float* __restrict__ thresholds)
{
    __shared__ uint32_t res[2];
    const float thr = lg_order_key_inv(lg_select_resolve(st_score, rank_score, 4, res));
    if (blockIdx.x == 0 && threadIdx.x == 0) thresholds[1] = thr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) mask[i] = v_list[i] <= thr ? 1 : 0;
}

// stand-alone select: out_value[0] = the rank-th smallest value; mask[i] = values[i] <= it (mask may be NULL)
__global__ void __launch_bounds__(256)
lg_select_finish_kernel(int N, uint32_t rank, const float* __restrict__ values, const LgSelect* __restrict__ st, uint8_t* __restrict__ mask,
                        float* __restrict__ out_value)
{
    __shared__ uint32_t res[2];
    const float thr = lg_order_key_inv(lg_select_resolve(st, rank, 4, res));
    if (blockIdx.x == 0 && threadIdx.x == 0) out_value[0] = thr;
    if (!mask) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) mask[i] = values[i] <= thr ? 1 : 0;
}

// out[j] = ((rows[0][j] + rows[1][j]) + rows[2][j]) + ... : the reference's sequential in-place float adds over the views
// (prune.py:144-155), one launch instead of V - 1 elementwise kernels.  Thread per column, rows streamed coalesced.
__global__ void __launch_bounds__(256)
lg_ordered_sum_kernel(int V, size_t n, const float* __restrict__ rows, size_t row_stride, float* __restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float acc = rows[j];
    for (int v = 1; v < V; v++) acc += rows[(size_t)v * row_stride + j];
    out[j] = acc;
}
