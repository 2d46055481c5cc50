// lg_binning.h -- binning kernels: depth-maximum reduction, K3 lg_duplicate (packed / pair keys), K5 lg_tile_ranges, lg_tile_order
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

// ------------------------------------------------------------------------------------------------
// Gathers what the host reads back after the scan into counters[0..3] (every word written: no memset of the counters):
// [1] = any prefiltered violation (bit 31 of the per-workgroup words), [2] = max over the per-workgroup depth maxima,
// [3] = instance count of the view (last element of the inclusive scan).
__global__ void __launch_bounds__(1024)
lg_reduce_dmax(int nblk, const uint32_t* __restrict__ blk_dmax, const uint32_t* __restrict__ last_offset, uint32_t* __restrict__ counters)
{
    __shared__ uint32_t wmax[16], wflag[16];
    uint32_t m = 0, f = 0;
    // this single block sits on the critical path of the forward's read-back: 16-byte loads (the array is 256-byte aligned)
    const int nvec = nblk >> 2;
    for (int i = threadIdx.x; i < nvec; i += 1024) {
        const uint4 v = reinterpret_cast<const uint4*>(blk_dmax)[i];
        m = max(max(m, v.x & 0x7FFFFFFFu), max(v.y & 0x7FFFFFFFu, max(v.z & 0x7FFFFFFFu, v.w & 0x7FFFFFFFu)));
        f |= (v.x | v.y | v.z | v.w) >> 31;
    }
    for (int i = (nvec << 2) + threadIdx.x; i < nblk; i += 1024) { const uint32_t v = blk_dmax[i]; m = max(m, v & 0x7FFFFFFFu); f |= v >> 31; }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) { m = max(m, (uint32_t)__shfl_xor((int)m, sh)); f |= (uint32_t)__shfl_xor((int)f, sh); }
    if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = m; wflag[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) { m = max(m, wmax[w]); f |= wflag[w]; }
        counters[0] = 0;
        counters[1] = f;
        counters[2] = m;
        counters[3] = last_offset[0];
    }
}

// K3: duplicate with keys
// Key formats.  PACKED: tile | (depth bits - bias) | Gaussian id in one u64, sorted keys-only on the tile+depth
// bits: the stable radix sort keeps the emission (= id) order among equal depths, and the id rides along for free
// (5 passes x 16 B instead of 6 x 24 B).  PAIRS (fallback when the fields do not fit 64 bits): tile<<32 | depth
// with the Gaussian id as value.
#define LG_DEPTH_BIAS (124u << 23) // bit pattern of 0.125f < the 0.2 near plane

template <bool PACKED>
__global__ void __launch_bounds__(256)
lg_duplicate(int N, int gx, int depth_bits, int gid_bits, const uint32_t* __restrict__ touched, const uint32_t* __restrict__ offsets,
             uint4* __restrict__ tinfo, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int ntiles, uint2* __restrict__ ranges)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // empty tiles keep {0, 0}: cleared here (this kernel runs before the sort) instead of by a memset
    for (int t = i; t < ntiles; t += gridDim.x * blockDim.x) ranges[t] = make_uint2(0u, 0u);
    if (i >= N) return;
    const uint32_t t = touched[i];
    if (t == 0) return;
    uint32_t off = offsets[i] - t;
    const uint4 r = tinfo[i];
    const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
    tinfo[i].w = off; // slot base of this Gaussian's instances (lg_slot_of: row address in the backward)
    if (PACKED) {
        const uint64_t low = ((uint64_t)(r.z - LG_DEPTH_BIAS) << gid_bits) | (uint32_t)i;
        const int sh = depth_bits + gid_bits;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) keys[off++] = ((uint64_t)(uint32_t)(y * gx + x) << sh) | low;
    } else {
        const uint64_t d = (uint64_t)r.z;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | d;
                vals[off] = (uint32_t)i;
                off++;
            }
    }
}

// Pre-sort slot of the instance of Gaussian `gid` in tile (tx, ty): lg_duplicate emits a Gaussian's instances row by row
// over its tile rectangle starting at tinfo.w, so the slot is a closed form of the rectangle -- no slot array is stored.
__device__ __forceinline__ uint32_t lg_slot_of(const uint4 r, int tx, int ty)
{
    const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF;
    return r.w + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
}

// K5: tile ranges from the sorted keys (one pass over 8 B per instance; nothing else is materialised in the packed
// format -- the blend kernels read the sorted keys themselves).  Pairs format: also writes entries[i] = Gaussian id.
//
// DROP > 0 (packed format only): the radix sort skipped the lowest `drop` depth bits to save a whole 8-bit pass
// (39 -> 32 sorted bits at C3: 5 -> 4 passes).  Entries that agree on the sorted bits form short runs (depth agrees to
// 2^-16 relative within one tile: about one pair per tile) which are still in emission order; the first thread of each
// run finishes the job with a stable insertion sort (merge sort beyond 32 entries) on the full tile | depth field.  The result is exactly the
// order a sort over all bits gives (stable => id order among equal depths).  Other threads may read a key of the run
// while it moves: they only look at its tile field, which all members share.
template <bool PACKED>
__global__ void __launch_bounds__(256)
lg_tile_ranges(uint32_t R, int tile_shift, int gid_bits, int drop, const uint64_t* keys /* == entries in the packed format */,
               const uint32_t* __restrict__ vals_sorted, uint64_t* entries, uint64_t* scratch, uint2* __restrict__ ranges)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const uint64_t key = keys[i];
    const uint32_t t = (uint32_t)(key >> tile_shift);
    if (!PACKED) entries[i] = (uint64_t)vals_sorted[i];
    uint64_t prev = 0;
    if (i == 0) ranges[t].x = 0;
    else {
        prev = keys[i - 1];
        const uint32_t tp = (uint32_t)(prev >> tile_shift);
        if (t != tp) { ranges[tp].y = i; ranges[t].x = i; }
    }
    if (i == R - 1) ranges[t].y = R;
    if (PACKED && drop > 0) {
        const int fs = gid_bits + drop;
        if (i == 0 || (prev >> fs) != (key >> fs)) {            // first entry of a run of equal sorted bits
            uint32_t e = i + 1;
            while (e < R && (keys[e] >> fs) == (key >> fs)) e++;
            if (e - i <= 32u) {
                for (uint32_t a = i + 1; a < e; a++) {           // stable insertion sort of [i, e) on tile | depth
                    const uint64_t k = entries[a];
                    uint32_t b = a;
                    while (b > i && (entries[b - 1] >> gid_bits) > (k >> gid_bits)) { entries[b] = entries[b - 1]; b--; }
                    if (b != a) entries[b] = k;
                }
            } else {
                // a long run (a slab of Gaussians within 2^-16 of one depth): bottom-up stable merge sort, O(n log n),
                // ping-ponging with the same range of the (now free) radix-sort input buffer
                uint64_t* src = entries; uint64_t* dst = scratch;
                for (uint32_t w = 1; w < e - i; w <<= 1) {
                    for (uint32_t lo = i; lo < e; lo += 2 * w) {
                        const uint32_t mid = min(lo + w, e), hi = min(lo + 2 * w, e);
                        uint32_t a = lo, b = mid, o = lo;
                        while (a < mid && b < hi) {
                            const uint64_t ka = src[a], kb = src[b];
                            if ((kb >> gid_bits) < (ka >> gid_bits)) { dst[o++] = kb; b++; } else { dst[o++] = ka; a++; }
                        }
                        while (a < mid) dst[o++] = src[a++];
                        while (b < hi) dst[o++] = src[b++];
                    }
                    uint64_t* t = src; src = dst; dst = t;
                }
                if (src != entries)
                    for (uint32_t a = i; a < e; a++) entries[a] = src[a];
            }
        }
    }
}


// Longest-processing-time-first dispatch order of the per-tile kernels.  The backward runs ONE wave per tile and only
// ~1.6 tiles per wave slot, so which tiles share a slot decides the makespan: handing out the long lists first lets
// the short ones fill the gaps.  Counting sort of the tiles by list length (256 buckets of 16 entries, longest first);
// the order inside a bucket is arbitrary, which is harmless because tiles are independent.
__global__ void __launch_bounds__(1024)
lg_tile_order(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ order)
{
    __shared__ uint32_t hist[256], base[256];
    const uint32_t tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int t = (int)tid; t < T; t += 1024) {
        const uint2 r = ranges[t];
        atomicAdd(&hist[255u - min((r.y - r.x) >> 4, 255u)], 1u); // bucket 0 = longest
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int b = 0; b < 256; b++) { base[b] = acc; acc += hist[b]; }
    }
    __syncthreads();
    for (int t = (int)tid; t < T; t += 1024) {
        const uint2 r = ranges[t];
        order[atomicAdd(&base[255u - min((r.y - r.x) >> 4, 255u)], 1u)] = (uint32_t)t;
    }
}
