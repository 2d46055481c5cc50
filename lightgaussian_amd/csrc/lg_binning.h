// lg_binning.h -- binning kernels: K2 lg_scan_blocks (+ depth maximum), K3 lg_duplicate (packed keys, fused digit histograms), lg_tile_ranges (one-stage cross-check), lg_tile_sort (second sort stage), lg_work_order
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

// ------------------------------------------------------------------------------------------------
// K2: exclusive scan of the per-workgroup instance counts of K1 (one word per 64 Gaussians), fused with the reduction of the
// per-workgroup depth maxima / prefiltered flags.  counters[0..3] (every word written: no memset):
//   [0] abort flags of the view: bit 0 = more instances than `capacity`, bit 1 = a depth beyond the `depth_bits` the keys
//       were laid out for (both can only fire in the capacity-bounded forward, whose host side knows neither number);
//       every later kernel of the view returns at once when [0] != 0
//   [1] any prefiltered violation (bit 31 of the per-workgroup words)   [2] largest depth bit pattern   [3] R
// Two levels in ONE launch: workgroup p scans words [1024 p, 1024 p + 1024) (blk_off = exclusive prefix INSIDE the part) and
// publishes the part's total / depth maximum; the workgroup that arrives last (one agent-scope atomic per workgroup, 46 at
// C3) scans the part totals into part_prefix and writes the counters.  Consumers add part_prefix[b >> 10] to blk_off[b].
// (Two single-workgroup versions of this kernel -- a load per iteration, then all loads in flight from registers -- both
// measured 50 us at C3: a lone workgroup on an otherwise idle device is slow whatever it does; so are the 10 us of
// lg_work_order's 8160 tiles.  46 workgroups finish the same work in a few microseconds.)
#define LG_DEPTH_BIAS (124u << 23) // bit pattern of 0.125f < the 0.2 near plane
#define LG_PART 1024               // words per part = threads per workgroup

__global__ void __launch_bounds__(LG_PART)
lg_scan_blocks(int nblk, const uint32_t* __restrict__ blk_sum, const uint32_t* __restrict__ blk_dmax, uint32_t* __restrict__ blk_off,
               uint32_t* part_sum, uint32_t* part_dmax, uint32_t* __restrict__ part_prefix, uint32_t* done, uint32_t capacity,
               int depth_bits, uint32_t* __restrict__ counters, uint32_t* __restrict__ status)
{
    __shared__ uint32_t wsum[16], wmax[16], wflag[16];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int nparts = (int)gridDim.x;
    {
        const int i = (int)blockIdx.x * LG_PART + (int)tid;
        const uint32_t v = i < nblk ? blk_sum[i] : 0u, d = i < nblk ? blk_dmax[i] : 0u;
        uint32_t inc = v, m = d & 0x7FFFFFFFu, f = d >> 31;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = __shfl_up(inc, s, 64);
            if ((int)lane >= s) inc += o;
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { m = max(m, (uint32_t)__shfl_xor((int)m, sh)); f |= (uint32_t)__shfl_xor((int)f, sh); }
        if (lane == 63u) { wsum[wave] = inc; wmax[wave] = m; wflag[wave] = f; }
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t t = wsum[w];
            woff += (w < (int)wave) ? t : 0u;
            total += t;
            m = max(m, wmax[w]); f |= wflag[w];
        }
        if (i < nblk) blk_off[i] = woff + inc - v;
        if (tid == 0) {
            __hip_atomic_store(&part_sum[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&part_dmax[blockIdx.x], m | (f << 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();                                        // the two words above before the arrival below
            s_last = (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)nparts - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) return;
    }
    // ---- last workgroup to arrive: scan the part totals (64-bit running sum: an overflow past 2^32 must be seen) ----
    __threadfence();
    uint32_t m = 0, f = 0;
    uint64_t carry = 0;
    for (int base = 0; base < nparts; base += LG_PART) {
        const int j = base + (int)tid;
        const uint32_t v = j < nparts ? __hip_atomic_load(&part_sum[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t d = j < nparts ? __hip_atomic_load(&part_dmax[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        m = max(m, d & 0x7FFFFFFFu); f |= d >> 31;
        uint32_t inc = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = __shfl_up(inc, s, 64);
            if ((int)lane >= s) inc += o;
        }
        __syncthreads();                                            // (wsum of the previous round / of the part scan is free)
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        uint64_t total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t t = wsum[w];
            woff += (w < (int)wave) ? t : 0u;
            total += t;
        }
        if (j < nparts) part_prefix[j] = (uint32_t)carry + woff + inc - v;
        carry += total;
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) { m = max(m, (uint32_t)__shfl_xor((int)m, sh)); f |= (uint32_t)__shfl_xor((int)f, sh); }
    __syncthreads();
    if (lane == 0u) { wmax[wave] = m; wflag[wave] = f; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) { m = max(m, wmax[w]); f |= wflag[w]; }
        const bool overflow = carry > 0xFFFFFFFFull;
        const uint32_t dspan = m > LG_DEPTH_BIAS ? m - LG_DEPTH_BIAS : 0u;
        uint32_t abort = (carry > (uint64_t)capacity) ? 1u : 0u;
        if (depth_bits < 32 && (dspan >> depth_bits) != 0u) abort |= 2u;
        counters[0] = abort;
        counters[1] = f;
        counters[2] = m;
        counters[3] = overflow ? 0xFFFFFFFFu : (uint32_t)carry;
        if (status) {
            // the caller's copy of the four words (lg_forward_bounded).  `status` may be pinned HOST memory that the host polls
            // while the rest of the view runs (word 0 is the "arrived" flag) -- words 1..3 first, then word 0,
            // each made visible system-wide
            status[1] = f; status[2] = m; status[3] = overflow ? 0xFFFFFFFFu : (uint32_t)carry;
            __threadfence_system();
            __hip_atomic_store(&status[0], abort, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        *done = 0u;                                                 // ready for the next view that reuses this buffer
    }
}

// K3: duplicate with keys
// Key format: tile | (depth bits - bias) >> store_drop | Gaussian id in one u64, sorted keys-only on the tile + depth bits: the
// stable radix sort keeps the emission (= id) order among equal depths, and the id rides along for free.  store_drop > 0 only
// when tile + depth + id exceed 64 bits (6 M Gaussians at 4K, 20 M at 1080p, ...): the lowest depth bits are then not STORED
// at all and the tile sort takes the full depth from tinfo (no pair format, no library sort: the former
// (tile << 32 | depth, id) fallback through hipCUB is gone).
//
// Wave-cooperative expansion: a wave takes the 64 Gaussians of one K1 workgroup, scans their instance counts, and then
// LANE l WRITES INSTANCE p = 64 c + l of the wave (c = 0, 1, ...), finding its Gaussian by a 6-step binary search over
// the 64 exclusive offsets in LDS -- consecutive lanes write consecutive 8-byte keys (one 512-byte store per wave
// instruction; the thread-per-Gaussian version wrote one scattered 8-byte store per lane: WRITE_SIZE 2.3x the key
// bytes).  The keys are in registers here, so the digit histograms of all radix passes are accumulated on the spot (LDS
// atomics, flushed once per workgroup): the sort needs no counting pass over the keys.  Persistent grid: a workgroup's 4
// waves walk groups of four K1 workgroups g = waves * blockIdx + wave, + waves * gridDim, ...
// 1024-thread workgroups, at most one per CU: the digit histograms are flushed with one global atomic per (workgroup, bin),
// and atomics on one address serialise (~12 ns each): 1024 workgroups of 256 threads cost ~1000 atomics per bin, about as long
// as the rest of the kernel; 256 workgroups of 1024 threads a quarter of that for the same number of waves.
#define LG_DUP_THREADS 1024
#ifndef LG_DUP_GRID
#define LG_DUP_GRID 512        // workgroups (two per CU; 256: 0.050 ms, 512: 0.044 ms at C3)
#endif
#define LG_DUP_WAVES (LG_DUP_THREADS / 64)
__global__ void __launch_bounds__(LG_DUP_THREADS)
lg_duplicate(int N, int nblk, int gx, int stored_depth_bits, int store_drop, int gid_bits, int sort_begin, int sort_end, uint32_t capacity,
             const uint32_t* __restrict__ touched, const uint32_t* __restrict__ blk_off, const uint32_t* __restrict__ part_prefix,
             const uint32_t* __restrict__ counters,
             uint32_t* __restrict__ offsets, uint4* __restrict__ tinfo, uint64_t* __restrict__ keys,
             int ntiles, uint2* __restrict__ ranges, uint32_t* __restrict__ hist, uint32_t range_init, uint32_t* __restrict__ long_tiles)
{
    __shared__ uint32_t lh[LG_SORT_MAX_PASSES * 256];
    __shared__ uint32_t s_exc[LG_DUP_WAVES][64], s_xy[LG_DUP_WAVES][64], s_w[LG_DUP_WAVES][64], s_hi[LG_DUP_WAVES][64], s_lo[LG_DUP_WAVES][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // tile ranges initialised here (this kernel runs before the sort) instead of by a memset: {0, 0} when lg_tile_ranges fills
    // them in (empty tiles keep it), {0xFFFFFFFF, 0} when the last radix pass does with atomicMin / atomicMax (two-stage sort;
    // lg_tile_sort turns what is left of it -- empty tiles, every tile of an aborted view -- into {0, 0})
    for (int t = blockIdx.x * LG_DUP_THREADS + (int)tid; t < ntiles; t += gridDim.x * LG_DUP_THREADS) ranges[t] = make_uint2(range_init, 0u);
    if (blockIdx.x == 0 && tid == 0) long_tiles[0] = 0u;             // list of the tiles lg_tile_sort leaves to lg_tile_sort_long
    const int passes = (sort_end - sort_begin + 7) / 8;
    for (int i = (int)tid; i < passes * 256; i += LG_DUP_THREADS) lh[i] = 0;
    __syncthreads();
    if (counters[0] != 0u) return;                 // view aborted by lg_scan_blocks (capacity-bounded forward)
    const int sh = stored_depth_bits + gid_bits;
    // a wave takes FOUR consecutive K1 workgroups per iteration and issues all of their loads (instance counts, base offsets,
    // tile rectangles) before it touches any of them: one exposed memory round trip per 256 Gaussians instead of two per 64
    // (the first version, one K1 workgroup per iteration with dependent loads, was latency-bound: 61 us at C3)
    const int ngroups = (nblk + 3) / 4;
    for (int g = blockIdx.x * LG_DUP_WAVES + (int)wave; g < ngroups; g += gridDim.x * LG_DUP_WAVES) {
        uint32_t t4[4], base4[4];
        uint4 r4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = g * 4 + u, i = b * 64 + (int)lane;
            t4[u] = (b < nblk && i < N) ? touched[i] : 0u;
            base4[u] = b < nblk ? blk_off[b] + part_prefix[b / LG_PART] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = (g * 4 + u) * 64 + (int)lane;
            r4[u] = make_uint4(0, 0, 0, 0);
            if (t4[u]) r4[u] = tinfo[i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = g * 4 + u, i = b * 64 + (int)lane;
            const uint32_t t = t4[u], base = base4[u];
            const uint4 r = r4[u];
            uint32_t inc = t;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const uint32_t o = __shfl_up(inc, s, 64);
                if ((int)lane >= s) inc += o;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (b < nblk && i < N) offsets[i] = base + inc;   // inclusive scan, as K9 expects (slot base = offsets - touched)
            if (total != 0u) {                                 // wave-uniform
                const uint32_t exc = inc - t;
                if (t) tinfo[i].w = base + exc;                // slot base of this Gaussian's instances (lg_slot_of: row address in the backward)
                s_exc[wave][lane] = exc;
                s_xy[wave][lane] = r.x;
                s_w[wave][lane] = (r.y & 0xFFFFu) - (r.x & 0xFFFFu);
                const uint64_t low = ((uint64_t)((r.z - LG_DEPTH_BIAS) >> store_drop) << gid_bits) | (uint32_t)i;
                s_lo[wave][lane] = (uint32_t)low;
                s_hi[wave][lane] = (uint32_t)(low >> 32);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint32_t p = lane; p < total; p += 64u) {
                    uint32_t j = 0;                            // largest j with exc[j] <= p: the Gaussian that owns instance p
#pragma unroll
                    for (uint32_t step = 32; step > 0; step >>= 1)
                        if (s_exc[wave][j + step] <= p) j += step;
                    const uint32_t k = p - s_exc[wave][j], w = s_w[wave][j], xy = s_xy[wave][j];
                    const uint32_t ky = k / w, kx = k - ky * w;
                    const uint32_t tile = ((xy >> 16) + ky) * (uint32_t)gx + (xy & 0xFFFFu) + kx;
                    const uint64_t pos = (uint64_t)base + p;
                    const uint64_t key = ((uint64_t)tile << sh) | ((uint64_t)s_hi[wave][j] << 32) | s_lo[wave][j];
                    if (pos < capacity) keys[pos] = key;
                    for (int q = 0; q < passes; q++) {
                        const int bit = sort_begin + 8 * q, nb = min(8, sort_end - bit);
                        atomicAdd(&lh[q * 256 + (uint32_t)((key >> bit) & ((1u << nb) - 1u))], 1u);
                    }
                }
                __builtin_amdgcn_wave_barrier();               // the next K1 workgroup overwrites this wave's LDS rows
            }
        }
    }
    __syncthreads();
    for (int i = (int)tid; i < passes * 256; i += LG_DUP_THREADS)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// Pre-sort slot of the instance of Gaussian `gid` in tile (tx, ty): lg_duplicate emits a Gaussian's instances row by row
// over its tile rectangle starting at tinfo.w, so the slot is a closed form of the rectangle -- no slot array is stored.
__device__ __forceinline__ uint32_t lg_slot_of(const uint4 r, int tx, int ty)
{
    const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF;
    return r.w + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
}

// K5: tile ranges from the sorted keys (one pass over 8 B per instance; nothing else is materialised -- the blend kernels read
// the sorted keys themselves).  NOT launched in the default TWO-STAGE sort (round 3): there the last radix pass leaves the ranges
// itself (lg_onesweep_pass) and lg_tile_sort (below) orders every list by depth.  With LG_FLAG_SORT_ALL_BITS the radix passes covered every STORED depth bit too (the round-2 scheme, kept as the independent
// cross-check of the tests) and this kernel completes the order on the bits a key beyond 64 bits does not store:
//   store_drop  low depth bits that are not in the key at all (tile + depth + id beyond 64 bits): read from tinfo[id].z
// Entries that agree on the stored bits form runs that are still in emission (= id) order; ordering a run stably on its
// finish_bits = store_drop low depth bits gives exactly the order a sort over all bits gives.  Short runs (<= 32 entries)
// are finished by their first thread with a stable
// insertion sort.  LONG runs -- a slab of coplanar splats puts thousands of entries at one depth; a narrow key (store_drop)
// makes every run longer -- are finished by the WHOLE WAVE of the thread that found them: a stable LSD counting sort over the
// low bits, 8 bits per pass (LDS digit counters, ballot-matched ranks, ping-pong with the free radix-sort input buffer), the
// wave serving its long runs one at a time.  (r2 used a single-thread merge sort there: 14.8 ms when the runs got long.)
// Other threads may read a key of a run while it moves: they only look at its tile field, which all members share.
#define LG_RUN_SHORT 32u
__device__ __forceinline__ uint32_t lg_low_bits(uint64_t key, int gid_bits, uint32_t gid_mask, int store_drop, uint32_t low_mask,
                                                const uint4* __restrict__ tinfo)
{
    if (store_drop == 0) return (uint32_t)(key >> gid_bits) & low_mask;          // wave-uniform branch
    return (tinfo[(uint32_t)key & gid_mask].z - LG_DEPTH_BIAS) & low_mask;        // the full depth pattern, from the binning record
}

// stable LSD counting sort of entries[i, e) on the low_bits low depth bits, by one wave (all lanes with i < R active)
__device__ __forceinline__ void lg_wave_sort_run(uint32_t i, uint32_t e, int low_bits, int gid_bits, uint32_t gid_mask, int store_drop,
                                                 uint64_t* entries, uint64_t* scratch, const uint4* __restrict__ tinfo, uint32_t* cnt /* LDS [256] */,
                                                 uint32_t lane)
{
    const uint32_t low_mask = low_bits >= 32 ? 0xFFFFFFFFu : ((1u << low_bits) - 1u);
    uint64_t* src = entries;
    uint64_t* dst = scratch;
    for (int shift = 0; shift < low_bits; shift += 8) {
        const uint32_t dmask = (1u << min(8, low_bits - shift)) - 1u;
        for (uint32_t c = lane; c < 256u; c += 64u) cnt[c] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t a = i + lane; a < e; a += 64u)
            atomicAdd(&cnt[(lg_low_bits(src[a], gid_bits, gid_mask, store_drop, low_mask, tinfo) >> shift) & dmask], 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {   // exclusive scan of the 256 counters: lane l owns digits 4 l .. 4 l + 3
            const uint32_t c0 = cnt[4u * lane], c1 = cnt[4u * lane + 1u], c2 = cnt[4u * lane + 2u], c3 = cnt[4u * lane + 3u];
            uint32_t inc = c0 + c1 + c2 + c3;
            const uint32_t own = inc;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const uint32_t o = __shfl_up(inc, s, 64);
                if ((int)lane >= s) inc += o;
            }
            const uint32_t ex = inc - own;
            cnt[4u * lane] = ex; cnt[4u * lane + 1u] = ex + c0; cnt[4u * lane + 2u] = ex + c0 + c1; cnt[4u * lane + 3u] = ex + c0 + c1 + c2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t a0 = i; a0 < e; a0 += 64u) {              // in list order: lanes of one digit keep their relative order
            const uint32_t a = a0 + lane;
            const bool valid = a < e;
            const uint64_t k = valid ? src[a] : 0ull;
            const uint32_t d = valid ? (lg_low_bits(k, gid_bits, gid_mask, store_drop, low_mask, tinfo) >> shift) & dmask : 0u;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (d >> b) & 1u;
                const uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t below = prefix_popc(peers);
            uint32_t pos = 0;
            if (valid) pos = cnt[d] + below;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (valid && below == 0u) cnt[d] += (uint32_t)__popcll(peers);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (valid) dst[i + pos] = k;
        }
        __threadfence();                                        // the next pass (other lanes) reads what this one wrote
        uint64_t* t = src; src = dst; dst = t;
    }
    if (src != entries) {
        for (uint32_t a = i + lane; a < e; a += 64u) entries[a] = src[a];
        __threadfence();
    }
}

__global__ void __launch_bounds__(256)
lg_tile_ranges(const uint32_t* __restrict__ counters, int tile_shift, int gid_bits, uint32_t gid_mask, int finish_bits, int store_drop,
               uint64_t* entries /* the sorted keys */, uint64_t* scratch, const uint4* __restrict__ tinfo, uint2* __restrict__ ranges,
               uint32_t* status /* lg_forward_bounded's status words, or NULL */)
{
    __shared__ uint32_t s_cnt[4][256];
    if (counters[0] != 0u) {                       // view aborted (capacity-bounded forward, or the sort's look-back gave up)
        // K2 handed the caller its copy of the abort word BEFORE the sort ran: an abort raised by the sort is added here
        if (status && blockIdx.x == 0 && threadIdx.x == 0 && (counters[0] & LG_ABORT_SORT)) {
            __hip_atomic_store(&status[0], counters[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const uint32_t R = counters[3];                // the grid is sized for the capacity; the instance count lives on the device
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int low_bits = finish_bits;              // 0 in the default two-stage sort: lg_tile_sort orders the lists
    bool mylong = false;
    uint32_t run_end = 0;
    if (i < R) {
        const uint64_t key = entries[i];
        const uint32_t t = (uint32_t)(key >> tile_shift);
        uint64_t prev = 0;
        if (i == 0) ranges[t].x = 0;
        else {
            prev = entries[i - 1];
            const uint32_t tp = (uint32_t)(prev >> tile_shift);
            if (t != tp) { ranges[tp].y = i; ranges[t].x = i; }
        }
        if (i == R - 1) ranges[t].y = R;
        if (low_bits > 0) {
            const int fs = gid_bits;
            if (i == 0 || (prev >> fs) != (key >> fs)) {            // first entry of a run of equal stored bits
                uint32_t e = i + 1;
                while (e < R && (entries[e] >> fs) == (key >> fs)) e++;
                if (e - i > LG_RUN_SHORT) { mylong = true; run_end = e; }
                else if (e - i > 1u) {
                    const uint32_t low_mask = low_bits >= 32 ? 0xFFFFFFFFu : ((1u << low_bits) - 1u);
                    for (uint32_t a = i + 1; a < e; a++) {           // stable insertion sort of [i, e) on the low depth bits
                        const uint64_t k = entries[a];
                        const uint32_t lk = lg_low_bits(k, gid_bits, gid_mask, store_drop, low_mask, tinfo);
                        uint32_t b = a;
                        while (b > i && lg_low_bits(entries[b - 1], gid_bits, gid_mask, store_drop, low_mask, tinfo) > lk) { entries[b] = entries[b - 1]; b--; }
                        if (b != a) entries[b] = k;
                    }
                }
            }
        }
    }
    // long runs: one at a time, by the whole wave (lanes beyond R idle along: every wave-level operation below is executed by
    // all 64 lanes)
    uint64_t big = __ballot(mylong);
    while (big) {
        const int srcl = (int)__builtin_ctzll(big);
        big &= big - 1;
        const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)i, srcl), re = (uint32_t)__builtin_amdgcn_readlane((int)run_end, srcl);
        lg_wave_sort_run(ri, re, low_bits, gid_bits, gid_mask, store_drop, entries, scratch, tinfo, s_cnt[wave], lane);
    }
}

// ------------------------------------------------------------------------------------------------
// Second stage of the sort (round 3): every tile's list ordered by depth INSIDE the tile.
// The global radix passes are bound by scattered 8-byte traffic and their own latency chain (0.035 ms per pass at C3 whatever
// the digit), and a key needs tile + depth = 13 + 27 bits sorted: five passes, four with the low bits left to K5 (round 2).
// But depth only has to be ordered among the ~500 entries of ONE tile, and those fit LDS: the global passes now cover the TILE
// bits only (two passes) and the lists are finished tile by tile -- a stable LSD counting sort over ALL depth bits, 9 bits per
// pass (three passes for 27 bits), entirely in LDS: keys in registers, per-wave digit counters, scan over waves and digits,
// scatter into LDS, read back in list order.  The global passes are stable and lg_duplicate emits in id order, so a list arrives
// in id order and a STABLE sort on depth gives exactly (depth, id) -- the order a sort over every key bit gives.  Depth bits a key
// beyond 64 bits does not store (store_drop) come from the binning record (tinfo[id].z) in the passes that touch them;
// coplanar slabs are just lists of equal digits.
//   n <= 1024         lg_tile_sort       one WAVE per tile, no workgroup barrier (9 KB LDS: 16 tiles in flight per CU)
//   n <= 4096         (same launch)      one 256-thread workgroup per tile (extra workgroups of the grid: those of other tiles return at once)
//   longer            lg_tile_sort_long  1024-thread workgroups over the few tiles the workgroup path lists: the same counting sort in
//                                        chunks of 8192 entries, ping-pong between the list and the (free) radix-sort input buffer
// Ranking without ballots: every lane ORs its lane bit into a 64-bit LDS word of its digit (ds_or_b64: commutative, so the result
// does not depend on the order the hardware serves the lanes in) and reads the word back -- that IS the set of lanes of this
// 64-key item with the same digit, which nine ballots + per-lane selects (~55 VALU instructions per item) would build.  The
// lowest lane of each set bumps the wave's counter of the digit and clears the word.  In the two LDS-resident kernels the words
// share LDS with the staging buffer (keys are in registers while they are ranked).
#define LG_TS_DIGIT 9
#define LG_TS_BINS (1 << LG_TS_DIGIT)
#define LG_TS_RUN 8u                                       // longest run of equal upper bits the wave path finishes by insertion
#define LG_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// bits [shift, shift + popc(dmask)) of the FULL depth pattern (minus bias) of a list entry; an all-ones key (padding) has all-ones digits
__device__ __forceinline__ uint32_t lg_ts_digit(uint64_t key, int shift, uint32_t dmask, int gid_bits, uint32_t gid_mask, int store_drop,
                                                const uint4* __restrict__ tinfo)
{
    // (shift + width never exceeds depth_bits, so the tile bits above the stored depth field are never reached)
    if (shift >= store_drop) return (uint32_t)(key >> (gid_bits + shift - store_drop)) & dmask;                        // wave-uniform branch
    if (key == ~0ull) return dmask;
    return ((tinfo[(uint32_t)key & gid_mask].z - LG_DEPTH_BIAS) >> shift) & dmask;
}

// rank of a key among the keys of its wave's earlier items and lower lanes with the same digit.  mask = 512 zeroed 64-bit words
// of this wave (zero again on return), wc = this wave's 16-bit digit counters.  Every lane takes part (padding keys included).
__device__ __forceinline__ uint32_t lg_ts_rank(uint32_t d, unsigned long long* mask, unsigned short* wc, unsigned long long mybit)
{
    atomicOr(&mask[d], mybit);
    LG_WAVE_SYNC();
    const unsigned long long peers = mask[d];
    const uint32_t prev = wc[d];
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
    LG_WAVE_SYNC();
    if (below == 0u) { wc[d] = (unsigned short)(prev + (uint32_t)__popcll(peers)); mask[d] = 0ull; }   // the lowest lane of the set
    LG_WAVE_SYNC();
    return prev + below;
}

// pass plan shared by the three kernels: P = ceil(depth_bits / 9) passes of ceil(depth_bits / P) bits
__device__ __forceinline__ int lg_ts_width(int depth_bits)
{
    const int passes = (depth_bits + LG_TS_DIGIT - 1) / LG_TS_DIGIT;
    return (depth_bits + passes - 1) / passes;
}

#define LG_TW_ITEMS 16
#define LG_TW_CAP (64 * LG_TW_ITEMS)
// one wave, one list of up to LG_TW_CAP entries.  stage = LG_TW_CAP keys of LDS (its first 4 KB double as the digit lane masks),
// cnt = 512 16-bit counters -- this wave's own: no workgroup barrier anywhere in here
__device__ __forceinline__ void lg_tile_sort_wave(uint32_t tile, uint32_t lane, uint64_t* stage, unsigned short* cnt, const uint2* __restrict__ ranges,
                                                  uint64_t* entries, int gid_bits, uint32_t gid_mask, int store_drop, int depth_bits,
                                                  const uint4* __restrict__ tinfo)
{
    const uint2 r = ranges[tile];
    const uint32_t n = r.y - r.x;
    if (n < 2u || n > (uint32_t)LG_TW_CAP) return; // longer lists: the workgroup / lg_tile_sort_long paths
    uint64_t* list = entries + r.x;
    const uint32_t items = (n + 63u) >> 6;                                    // 1 .. LG_TW_ITEMS, wave-uniform
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(stage);
    const unsigned long long mybit = 1ull << lane;
    uint64_t key[LG_TW_ITEMS];
    uint32_t rnk[LG_TW_ITEMS];
#pragma unroll
    for (int k = 0; k < LG_TW_ITEMS; k++) {
        const uint32_t idx = (uint32_t)k * 64u + lane;
        key[k] = ((uint32_t)k < items && idx < n) ? list[idx] : ~0ull;        // padding sorts behind everything (and stays there: stable)
    }
    const int width = lg_ts_width(depth_bits);
    // The LOWEST digit is not radix-sorted: after the passes over the upper digits the list is ordered on depth >> width, entries
    // that agree there form runs (one pair per tile on a scene of random depths: the upper digits resolve 2^-14 relative), and a
    // run of up to LG_TS_RUN entries is put in order by its first lane with a few compare-exchanges in LDS -- a third of the
    // kernel for the price of two LDS reads per entry.  A longer run (coplanar splats) sends the wave through all digits
    // instead: stable passes over what the first attempt left still end in (depth, id).  Needs every depth bit in the key.
    const bool finish = store_drop == 0 && depth_bits > width;
    for (int attempt = 0; attempt < 2; attempt++) {
    const int first = (finish && attempt == 0) ? width : 0;
    for (int shift = first; shift < depth_bits; shift += width) {
        const uint32_t dmask = (1u << min(width, depth_bits - shift)) - 1u;
        {   // clear the lane masks (4 KB) and the counters (1 KB): 16-byte stores
            uint4* z = reinterpret_cast<uint4*>(stage);
#pragma unroll
            for (int i = 0; i < (LG_TS_BINS * 8) / (64 * 16); i++) z[(uint32_t)i * 64u + lane] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(cnt)[lane] = make_uint4(0, 0, 0, 0);
        }
        LG_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < LG_TW_ITEMS; k++) {
            if ((uint32_t)k < items) {                                        // wave-uniform
                const uint32_t d = lg_ts_digit(key[k], shift, dmask, gid_bits, gid_mask, store_drop, tinfo);
                rnk[k] = lg_ts_rank(d, mask, cnt, mybit) | (d << 16);
            }
        }
        {   // exclusive scan of the 512 counters: lane l owns digits 8 l .. 8 l + 7 (one 16-byte read, one 16-byte write)
            const uint4 c = reinterpret_cast<const uint4*>(cnt)[lane];
            const uint32_t c0 = c.x & 0xFFFFu, c1 = c.x >> 16, c2 = c.y & 0xFFFFu, c3 = c.y >> 16, c4 = c.z & 0xFFFFu, c5 = c.z >> 16, c6 = c.w & 0xFFFFu,
                           c7 = c.w >> 16;
            const uint32_t own = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
            uint32_t inc = own;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const uint32_t o = __shfl_up(inc, s, 64);
                if ((int)lane >= s) inc += o;
            }
            const uint32_t e0 = inc - own, e1 = e0 + c0, e2 = e1 + c1, e3 = e2 + c2, e4 = e3 + c3, e5 = e4 + c4, e6 = e5 + c5, e7 = e6 + c6;
            reinterpret_cast<uint4*>(cnt)[lane] = make_uint4(e0 | (e1 << 16), e2 | (e3 << 16), e4 | (e5 << 16), e6 | (e7 << 16));
        }
        LG_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < LG_TW_ITEMS; k++)
            if ((uint32_t)k < items) stage[(uint32_t)cnt[rnk[k] >> 16] + (rnk[k] & 0xFFFFu)] = key[k];
        LG_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < LG_TW_ITEMS; k++)
            if ((uint32_t)k < items) key[k] = stage[(uint32_t)k * 64u + lane];
        LG_WAVE_SYNC();
    }
    if (first == 0) break;                         // every digit went through the passes
    // ---- runs of equal upper bits: found from the neighbours in LDS, ordered by their first lane ----
    const int up = gid_bits + width;               // key >> up = tile | depth >> width
    uint32_t starts = 0;                           // bit k: my entry of item k opens a run of at least two
#pragma unroll
    for (int k = 0; k < LG_TW_ITEMS; k++) {
        const uint32_t idx = (uint32_t)k * 64u + lane;
        if ((uint32_t)k < items && idx + 1u < n) {
            const uint64_t me = key[k] >> up;
            const bool opens = idx == 0u || (stage[idx - 1u] >> up) != me;
            if (opens && (stage[idx + 1u] >> up) == me) starts |= 1u << k;
        }
    }
    LG_WAVE_SYNC();
    bool too_long = false;
    while (starts) {                               // (divergent: few lanes own a run at all)
        const uint32_t k = (uint32_t)__builtin_ctz(starts);
        starts &= starts - 1u;
        const uint32_t idx = k * 64u + lane;
        const uint64_t me = stage[idx] >> up;
        uint32_t e = idx + 2u;
        while (e < n && e - idx <= LG_TS_RUN && (stage[e] >> up) == me) e++;     // (a neighbouring run may be moving: its members all differ from `me`)
        if (e - idx > LG_TS_RUN) { too_long = true; continue; }
        for (uint32_t a = idx + 1u; a < e; a++) {   // insertion sort on the whole key: upper bits equal, so this is (low depth bits, id)
            const uint64_t v = stage[a];
            uint32_t b = a;
            while (b > idx && stage[b - 1u] > v) { stage[b] = stage[b - 1u]; b--; }
            if (b != a) stage[b] = v;
        }
    }
    LG_WAVE_SYNC();
    if (__ballot(too_long) == 0ull) break;
#pragma unroll
    for (int k = 0; k < LG_TW_ITEMS; k++)
        if ((uint32_t)k < items) key[k] = stage[(uint32_t)k * 64u + lane];
    LG_WAVE_SYNC();
    }
    // `stage` holds the list in its final order (the last pass scattered into it; the finishing step worked in place)
#pragma unroll
    for (int k = 0; k < LG_TW_ITEMS; k++) {
        const uint32_t idx = (uint32_t)k * 64u + lane;
        if ((uint32_t)k < items && idx < n) list[idx] = stage[idx];
    }
}

#define LG_TS_THREADS 256
#define LG_TS_WAVES (LG_TS_THREADS / 64)
#define LG_TS_ITEMS 16                                     // keys per thread: lists up to 4096 entries stay in LDS (32 KB)
#define LG_TS_CAP (LG_TS_THREADS * LG_TS_ITEMS)
static_assert(LG_TS_CAP < 65536, "16-bit LDS counters");
// one 256-thread workgroup, one list of LG_TW_CAP < n <= LG_TS_CAP entries.  stage = LG_TS_CAP keys of LDS (its first 16 KB double
// as the four waves' lane masks)
__device__ __forceinline__ void lg_tile_sort_wg(uint32_t tile, uint64_t* stage, unsigned short (*wcnt)[LG_TS_BINS], uint32_t* lbase, uint32_t* wtot,
                                                const uint2* __restrict__ ranges, uint64_t* entries, int gid_bits, uint32_t gid_mask, int store_drop,
                                                int depth_bits, const uint4* __restrict__ tinfo, uint32_t* long_tiles)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint2 r = ranges[tile];
    const uint32_t n = r.y - r.x;
    if (n <= (uint32_t)LG_TW_CAP) return;          // a wave's
    if (n > (uint32_t)LG_TS_CAP) {
        if (tid == 0) long_tiles[1u + atomicAdd(&long_tiles[0], 1u)] = tile;
        return;
    }
    uint64_t* list = entries + r.x;
    const uint32_t items = (n + LG_TS_THREADS - 1u) / LG_TS_THREADS;          // 5 .. LG_TS_ITEMS, workgroup-uniform
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(stage) + wave * LG_TS_BINS;
    const unsigned long long mybit = 1ull << lane;
    uint64_t key[LG_TS_ITEMS];
    uint32_t rnk[LG_TS_ITEMS];
#pragma unroll
    for (int k = 0; k < LG_TS_ITEMS; k++) {
        const uint32_t idx = (wave * items + (uint32_t)k) * 64u + lane;       // list order = (wave, item, lane)
        key[k] = ((uint32_t)k < items && idx < n) ? list[idx] : ~0ull;
    }
    const int width = lg_ts_width(depth_bits);
    // (every digit through the passes here: finishing the lowest one by run insertion as lg_tile_sort_wave does was measured on this
    //  path -- 0.107 -> 0.110 ms on the dense scene, and 0.045 -> 0.084 ms on the heavy-tailed one, whose mid-length lists sit in
    //  the pile and hold runs beyond LG_TS_RUN: two passes + three more)
    for (int shift = 0; shift < depth_bits; shift += width) {
        const uint32_t dmask = (1u << min(width, depth_bits - shift)) - 1u;
        {   // clear the lane masks (16 KB) and the counters (4 KB)
            uint4* z = reinterpret_cast<uint4*>(stage);
#pragma unroll
            for (int i = 0; i < (LG_TS_WAVES * LG_TS_BINS * 8) / (LG_TS_THREADS * 16); i++) z[(uint32_t)i * LG_TS_THREADS + tid] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(&wcnt[0][0])[tid] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LG_TS_ITEMS; k++) {
            if ((uint32_t)k < items) {                                        // workgroup-uniform
                const uint32_t d = lg_ts_digit(key[k], shift, dmask, gid_bits, gid_mask, store_drop, tinfo);
                rnk[k] = lg_ts_rank(d, mask, wcnt[wave], mybit) | (d << 16);  // (rank < 1024; the digit rides along for the scatter)
            }
        }
        __syncthreads();
        // thread t owns digits 2 t and 2 t + 1: exclusive scan over the waves (written back), then over the digits
        uint32_t run0 = 0, run1 = 0;
#pragma unroll
        for (int w = 0; w < LG_TS_WAVES; w++) {
            uint32_t* pw = reinterpret_cast<uint32_t*>(&wcnt[w][0]) + tid;
            const uint32_t pr = *pw;
            *pw = run0 | (run1 << 16);
            run0 += pr & 0xFFFFu; run1 += pr >> 16;
        }
        const uint32_t own = run0 + run1;
        uint32_t inc = own;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = __shfl_up(inc, s, 64);
            if ((int)lane >= s) inc += o;
        }
        if (lane == 63u) wtot[wave] = inc;
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t w = 0; w < wave; w++) carry += wtot[w];
        lbase[2u * tid] = carry + inc - own;
        lbase[2u * tid + 1u] = carry + inc - own + run0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LG_TS_ITEMS; k++) {
            if ((uint32_t)k < items) {
                const uint32_t d = rnk[k] >> 16;
                stage[lbase[d] + (uint32_t)wcnt[wave][d] + (rnk[k] & 0xFFFFu)] = key[k];
            }
        }
        __syncthreads();
        if (shift + width < depth_bits) {
#pragma unroll
            for (int k = 0; k < LG_TS_ITEMS; k++)
                if ((uint32_t)k < items) key[k] = stage[(wave * items + (uint32_t)k) * 64u + lane];
            __syncthreads();                       // every key read back before the masks are cleared over it
        }
    }
    for (uint32_t q = tid; q < n; q += LG_TS_THREADS) list[q] = stage[q];
}

// ONE launch for both: workgroups [0, ceil(tiles / 4)) are four independent waves with a tile each (wave-synchronous code only:
// no workgroup barrier), workgroups [ceil(tiles / 4), + tiles) take one tile each as a whole and return at once unless its list
// is of the middle class -- so the second class costs the uniform scene no launch of its own (an empty launch and its gap: ~9 us).
#define LG_TS_SMEM_STAGE (LG_TS_CAP * 8)
#define LG_TS_SMEM_WCNT (LG_TS_WAVES * LG_TS_BINS * 2)
#define LG_TS_SMEM (LG_TS_SMEM_STAGE + LG_TS_SMEM_WCNT + LG_TS_BINS * 4 + 64)
static_assert(LG_TS_WAVES * (LG_TW_CAP * 8 + LG_TS_BINS * 2) <= LG_TS_SMEM, "the four single-wave sorts fit the workgroup sort's LDS");
__global__ void __launch_bounds__(LG_TS_THREADS)
lg_tile_sort(int ntiles, const uint32_t* __restrict__ counters, uint2* ranges, uint64_t* entries, int gid_bits, uint32_t gid_mask,
             int store_drop, int depth_bits, const uint4* __restrict__ tinfo, uint32_t* long_tiles, uint32_t* status)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LG_TS_SMEM];
    const bool aborted = counters[0] != 0u;        // capacity-bounded forward gave the view up, or the sort's look-back did
    const uint32_t nquad = ((uint32_t)ntiles + 3u) / 4u;
    if (aborted && status && blockIdx.x == 0 && threadIdx.x == 0 && (counters[0] & LG_ABORT_SORT)) {
        // K2 handed the caller its copy of the abort word BEFORE the sort ran: an abort raised by the sort is added here
        __hip_atomic_store(&status[0], counters[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (blockIdx.x < nquad) {
        const uint32_t wave = threadIdx.x >> 6, tile = blockIdx.x * 4u + wave;
        if (tile >= (uint32_t)ntiles) return;
        // The last radix pass left {begin, end} of every tile that has entries (atomicMin / atomicMax over {0xFFFFFFFF, 0}); the
        // blend kernels expect {0, 0} for an empty tile -- and for EVERY tile of an aborted view, whose ranges may be half-made
        const uint2 r0 = ranges[tile];
        if (aborted || r0.x == 0xFFFFFFFFu) {
            if ((threadIdx.x & 63u) == 0u) ranges[tile] = make_uint2(0u, 0u);
            return;
        }
        unsigned char* mine = smem + wave * (LG_TW_CAP * 8 + LG_TS_BINS * 2);
        lg_tile_sort_wave(tile, threadIdx.x & 63u, reinterpret_cast<uint64_t*>(mine), reinterpret_cast<unsigned short*>(mine + LG_TW_CAP * 8), ranges, entries,
                          gid_bits, gid_mask, store_drop, depth_bits, tinfo);
    } else {
        if (aborted) return;
        lg_tile_sort_wg(blockIdx.x - nquad, reinterpret_cast<uint64_t*>(smem), reinterpret_cast<unsigned short (*)[LG_TS_BINS]>(smem + LG_TS_SMEM_STAGE),
                        reinterpret_cast<uint32_t*>(smem + LG_TS_SMEM_STAGE + LG_TS_SMEM_WCNT),
                        reinterpret_cast<uint32_t*>(smem + LG_TS_SMEM_STAGE + LG_TS_SMEM_WCNT + LG_TS_BINS * 4), ranges, entries, gid_bits, gid_mask, store_drop,
                        depth_bits, tinfo, long_tiles);
    }
}

// Lists beyond LG_TS_CAP: persistent grid over the tiles the workgroup path listed (none on most scenes: an empty launch).
#define LG_TL_THREADS 1024
#define LG_TL_WAVES (LG_TL_THREADS / 64)
#define LG_TL_ITEMS 8
#define LG_TL_CHUNK (LG_TL_THREADS * LG_TL_ITEMS)
#ifndef LG_TL_GRID
#define LG_TL_GRID 256
#endif
__global__ void __launch_bounds__(LG_TL_THREADS)
lg_tile_sort_long(const uint32_t* __restrict__ counters, const uint2* __restrict__ ranges, uint64_t* entries, uint64_t* scratch, int gid_bits,
                  uint32_t gid_mask, int store_drop, int depth_bits, const uint4* __restrict__ tinfo, const uint32_t* long_tiles)
{
    __shared__ __attribute__((aligned(16))) unsigned long long masks[LG_TL_WAVES][LG_TS_BINS];   // 64 KB
    __shared__ __attribute__((aligned(16))) unsigned short wcnt[LG_TL_WAVES][LG_TS_BINS];        // 16 KB
    __shared__ uint32_t gbase[LG_TS_BINS];       // position of the next key of digit d
    __shared__ uint32_t gnext[LG_TS_BINS];       // digit totals of the next pass
    __shared__ uint32_t wtot[4];
    if (counters[0] != 0u) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nlong = long_tiles[0];
    if (blockIdx.x >= nlong) return;
    const unsigned long long mybit = 1ull << lane;
    for (uint32_t i = tid; i < LG_TL_WAVES * LG_TS_BINS; i += LG_TL_THREADS) (&masks[0][0])[i] = 0ull;   // (zero again after every item)
    const int width = lg_ts_width(depth_bits);
    for (uint32_t item = blockIdx.x; item < nlong; item += gridDim.x) {
        const uint2 r = ranges[long_tiles[1u + item]];
        const uint32_t n = r.y - r.x;
        uint64_t* src = entries + r.x;
        uint64_t* dst = scratch + r.x;
        // digit totals of the first pass: one counting pass over the list (eight loads in flight per thread); the totals of every
        // later pass are counted while the pass before it scatters (the keys are in registers there)
        if (tid < LG_TS_BINS) gnext[tid] = 0u;
        __syncthreads();
        {
            const uint32_t dmask0 = (1u << min(width, depth_bits)) - 1u;
            for (uint32_t i0 = 0; i0 < n; i0 += LG_TL_CHUNK) {
                uint64_t kk[LG_TL_ITEMS];
#pragma unroll
                for (int k = 0; k < LG_TL_ITEMS; k++) {
                    const uint32_t idx = i0 + (uint32_t)k * LG_TL_THREADS + tid;
                    kk[k] = idx < n ? src[idx] : ~0ull;
                }
#pragma unroll
                for (int k = 0; k < LG_TL_ITEMS; k++)
                    if (i0 + (uint32_t)k * LG_TL_THREADS + tid < n) atomicAdd(&gnext[lg_ts_digit(kk[k], 0, dmask0, gid_bits, gid_mask, store_drop, tinfo)], 1u);
            }
        }
        __syncthreads();
        for (int shift = 0; shift < depth_bits; shift += width) {
            const uint32_t dmask = (1u << min(width, depth_bits - shift)) - 1u;
            const bool more = shift + width < depth_bits;
            const uint32_t nmask = more ? (1u << min(width, depth_bits - shift - width)) - 1u : 0u;
            // ---- exclusive scan of this pass's digit totals (gnext -> gbase; gnext restarts for the next pass) ----
            uint32_t t0 = 0, t1 = 0, inc = 0;
            if (tid < 256u) {
                t0 = gnext[2u * tid]; t1 = gnext[2u * tid + 1u];
                gnext[2u * tid] = 0u; gnext[2u * tid + 1u] = 0u;
                inc = t0 + t1;
#pragma unroll
                for (int s = 1; s < 64; s <<= 1) {
                    const uint32_t o = __shfl_up(inc, s, 64);
                    if ((int)lane >= s) inc += o;
                }
                if (lane == 63u) wtot[wave] = inc;
            }
            __syncthreads();
            if (tid < 256u) {
                uint32_t carry = 0;
                for (uint32_t w = 0; w < wave; w++) carry += wtot[w];
                gbase[2u * tid] = carry + inc - (t0 + t1);
                gbase[2u * tid + 1u] = carry + inc - t1;
            }
            // ---- phase B: chunk by chunk in list order ----
            for (uint32_t c0 = 0; c0 < n; c0 += LG_TL_CHUNK) {
                const uint32_t cn = min((uint32_t)LG_TL_CHUNK, n - c0);
                for (uint32_t i = tid; i < LG_TL_WAVES * LG_TS_BINS / 2; i += LG_TL_THREADS) reinterpret_cast<uint32_t*>(&wcnt[0][0])[i] = 0u;
                uint64_t key[LG_TL_ITEMS];
                uint32_t rnk[LG_TL_ITEMS];
#pragma unroll
                for (int k = 0; k < LG_TL_ITEMS; k++) {
                    const uint32_t idx = (wave * LG_TL_ITEMS + (uint32_t)k) * 64u + lane;
                    key[k] = idx < cn ? src[c0 + idx] : ~0ull;
                }
                __syncthreads();                   // counters cleared; gbase of the previous chunk / of phase A complete
#pragma unroll
                for (int k = 0; k < LG_TL_ITEMS; k++) {
                    if ((wave * LG_TL_ITEMS + (uint32_t)k) * 64u < cn) {                       // wave-uniform: items with any key
                        const uint32_t d = lg_ts_digit(key[k], shift, dmask, gid_bits, gid_mask, store_drop, tinfo);
                        rnk[k] = lg_ts_rank(d, masks[wave], wcnt[wave], mybit) | (d << 16);
                    }
                }
                __syncthreads();
                uint32_t run0 = 0, run1 = 0;
                if (tid < 256u) {
#pragma unroll
                    for (int w = 0; w < LG_TL_WAVES; w++) {
                        uint32_t* pw = reinterpret_cast<uint32_t*>(&wcnt[w][0]) + tid;
                        const uint32_t pr = *pw;
                        *pw = run0 | (run1 << 16);
                        run0 += pr & 0xFFFFu; run1 += pr >> 16;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < LG_TL_ITEMS; k++) {
                    const uint32_t idx = (wave * LG_TL_ITEMS + (uint32_t)k) * 64u + lane;
                    if (idx < cn) {
                        const uint32_t d = rnk[k] >> 16;
                        dst[gbase[d] + (uint32_t)wcnt[wave][d] + (rnk[k] & 0xFFFFu)] = key[k];
                        if (more) atomicAdd(&gnext[lg_ts_digit(key[k], shift + width, nmask, gid_bits, gid_mask, store_drop, tinfo)], 1u);
                    }
                }
                __syncthreads();                   // every position of this chunk taken before the bases move
                // (the padding keys of the last chunk were counted under the all-ones digit: nothing is placed behind them)
                if (tid < 256u) { gbase[2u * tid] += run0; gbase[2u * tid + 1u] += run1; }
            }
            // the next pass (other waves of THIS workgroup) reads what this one wrote: workgroup scope is enough -- the waves share
            // the CU's vector cache -- and far cheaper than an agent-scope release (an L2 write-back on a multi-XCD part)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint64_t* t = src; src = dst; dst = t;
        }
        if (src != entries + r.x) {
            for (uint32_t i = tid; i < n; i += LG_TL_THREADS) entries[r.x + i] = src[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// Work list of the backward blend: one item per (tile, segment of S list entries), longest first.  The backward runs one
// wave per item and only ~1.6 items per wave slot, so which items share a slot decides the makespan: handing out the long
// ones first lets the short ones fill the gaps (longest-processing-time-first; counting sort by length, 256 buckets of 16
// entries, order inside a bucket arbitrary -- items are independent).  Empty tiles produce no item.  meta[0] = item count.
// Runs as ONE EXTRA WORKGROUP of the forward blend (lg_blend_fwd, block index ntiles_pad): the ranges are final by then, and
// the 10 us a lone workgroup needs for 8160 tiles hide behind the blend instead of standing in front of the backward.
// It also lists the items of the tiles whose list goes through the parallel long-tile forward (par_work, meta[4] items;
// lg_blend_fwd_seg / _scan / _rewalk): the lists longer than lg_par_min() -- a pure function of THIS view's device-side
// numbers (long-tile mode of the call, S, instance count R, tile count), the same one the tile workgroups of lg_blend_fwd
// evaluate to leave those lists alone.
__device__ __forceinline__ uint32_t lg_par_min(int long_mode, int S, uint32_t R, int ntiles)
{
    // 0 = serial: every list is walked by lg_blend_fwd.  2 = every multi-segment list in parallel.  1 ("auto") = lists longer
    // than two segments AND four times the mean list of the view: such a list is the forward's critical path (its serial walk
    // outlasts everything else), shorter ones are not.  Measured (fwd+bwd views/s, heavy-tailed scene / 6 M Gaussians at
    // 1600x1060): every multi-segment list 466 / 310; > 2 S 459 / 338; > 3 S 439 / 356; > 4 S 411 / 352; > 8 S 380 / 347;
    // serial only 416 / 353.  With the mean-relative rule the dense scene (mean list 1200) sends nothing and the heavy-tailed
    // one (mean 545, lists to 24 000) everything above 2180.
    if (long_mode == 0) return 0u;
    if (long_mode == 2) return (uint32_t)S;
    const uint64_t mean4 = 4ull * (uint64_t)R / (uint64_t)max(ntiles, 1);
    return max(2u * (uint32_t)S, (uint32_t)min(mean4, (uint64_t)(1u << 30)));
}

__device__ __forceinline__ void lg_work_order_body(int T, int S, const uint2* __restrict__ ranges, uint2* __restrict__ work,
                                                   uint32_t* __restrict__ meta, uint32_t* hist /* LDS [256] */,
                                                   uint32_t* base /* LDS [258] */, uint32_t tid, uint32_t nthreads,
                                                   uint2* __restrict__ par_work, uint32_t par_min, uint32_t* __restrict__ par_arrived)
{
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) { base[256] = 0; base[257] = 0; }
    __syncthreads();
    uint32_t longest = 0;
    for (int t = (int)tid; t < T; t += (int)nthreads) {
        const uint2 r = ranges[t];
        longest = max(longest, r.y - r.x);
        for (uint32_t lo = 0, n = r.y - r.x; lo < n; lo += (uint32_t)S)
            atomicAdd(&hist[255u - min(min((uint32_t)S, n - lo) >> 4, 255u)], 1u); // bucket 0 = longest
    }
    if (longest != 0u) atomicMax(&base[256], longest);
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int b = 0; b < 256; b++) { base[b] = acc; acc += hist[b]; }
        meta[0] = acc;
        meta[1] = base[256];                                    // longest list of the view
        meta[2] = (uint32_t)S;                                  // checked by the backward (lg_blend_bwd, lg_preprocess_bwd)
        meta[3] = par_min;
        meta[5] = 0u;                                           // pixels resolved by lg_count_fixup (diagnostics, lg_debug_view_meta)
    }
    __syncthreads();
    for (int t = (int)tid; t < T; t += (int)nthreads) {
        const uint2 r = ranges[t];
        const uint32_t n = r.y - r.x;
        const bool par = par_min != 0u && n > par_min;
        if (par) par_arrived[t] = 0u;                             // arrival counter of the tile's segments (lg_blend_fwd_seg -> lg_scan_tile)
        uint32_t seg = 0;
        for (uint32_t lo = 0; lo < n; lo += (uint32_t)S, seg++) {
            work[atomicAdd(&base[255u - min(min((uint32_t)S, n - lo) >> 4, 255u)], 1u)] = make_uint2((uint32_t)t, seg);
            if (par) par_work[atomicAdd(&base[257], 1u)] = make_uint2((uint32_t)t, seg);
        }
    }
    __syncthreads();
    if (tid == 0) meta[4] = base[257];
}
