// lg_compact.h -- row compaction of the Gaussian tensors after a prune (second half).
// Replaces GaussianModel._prune_optimizer / prune_points (scene/gaussian_model.py:564-600): the reference indexes every
// parameter, both Adam moments of every parameter and three bookkeeping tensors with the boolean keep-mask -- 21 boolean
// index kernels, each with its own nonzero() + host sync.  Here: ONE prefix scan of the mask (lg_compact_plan: per-1024-row
// counts -> single-workgroup scan -> destination row per kept row, total on the device) and ONE launch that moves the rows of
// all tensors (lg_compact_rows, grid.y = tensor).  Rows keep their order, so the result equals tensor[mask] bit for bit.
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

#define LG_COMPACT_ROWS 1024          // rows per workgroup of the plan kernels (256 threads x 4)
#define LG_COMPACT_MAX_TENSORS 32

// per-workgroup number of kept rows
__global__ void __launch_bounds__(256)
lg_compact_count(int N, const uint8_t* __restrict__ keep, uint32_t* __restrict__ blk_sum)
{
    __shared__ uint32_t ws[4];
    const int r0 = blockIdx.x * LG_COMPACT_ROWS + (int)threadIdx.x * 4;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) c += (r0 + k < N && keep[r0 + k]) ? 1u : 0u;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) c += (uint32_t)__shfl_xor((int)c, sh);
    if ((threadIdx.x & 63u) == 0u) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of n words by one workgroup (n = N / 1024: a few thousand), total to *total
__global__ void __launch_bounds__(1024)
lg_scan_words(int n, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int32_t* __restrict__ total)
{
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + (int)tid;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = __shfl_up(x, s, 64);
            if ((int)lane >= s) x += o;
        }
        if (lane == 63u) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t t = wsum[w];
            woff += (w < (int)wave) ? t : 0u;
            tot += t;
        }
        if (i < n) out[i] = carry + woff + x - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) *total = (int32_t)carry;
}

// dest[row] = index of the row in the compacted tensors, or -1 for pruned rows
__global__ void __launch_bounds__(256)
lg_compact_dest(int N, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ blk_off, int32_t* __restrict__ dest)
{
    __shared__ uint32_t ws[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * LG_COMPACT_ROWS + (int)threadIdx.x * 4;
    bool k4[4];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { k4[k] = r0 + k < N && keep[r0 + k]; c += k4[k] ? 1u : 0u; }
    uint32_t inc = c;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t o = __shfl_up(inc, s, 64);
        if ((int)lane >= s) inc += o;
    }
    if (lane == 63u) ws[wave] = inc;
    __syncthreads();
    uint32_t pos = blk_off[blockIdx.x] + inc - c;
    for (uint32_t w = 0; w < wave; w++) pos += ws[w];
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (r0 + k < N) { dest[r0 + k] = k4[k] ? (int32_t)pos : -1; pos += k4[k] ? 1u : 0u; }
}

// rows of every tensor to their compacted position.  Tensor t: N rows of words[t] 32-bit words, contiguous.
struct LgCompactArgs {
    const uint32_t* src[LG_COMPACT_MAX_TENSORS];
    uint32_t* dst[LG_COMPACT_MAX_TENSORS];
    uint32_t words[LG_COMPACT_MAX_TENSORS];
};
__global__ void __launch_bounds__(256)
lg_compact_move(int N, const int32_t* __restrict__ dest, LgCompactArgs a)
{
    const uint32_t t = blockIdx.y;
    const uint32_t wpr = a.words[t];
    const uint32_t* __restrict__ src = a.src[t];
    uint32_t* __restrict__ dst = a.dst[t];
    const size_t total = (size_t)N * wpr;
    for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < total; w += (size_t)gridDim.x * 256) {
        const uint32_t row = (uint32_t)(w / wpr), col = (uint32_t)(w - (size_t)row * wpr);
        const int32_t d = dest[row];
        if (d >= 0) dst[(size_t)d * wpr + col] = src[w];
    }
}
