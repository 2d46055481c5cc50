// lg_sort.h -- K4: hand-written onesweep LSD radix sort of 64-bit keys (keys only, stable, ascending) for gfx950 / wave64.
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
//
// One launch per 8-bit digit ("chained scan with decoupled look-back", the onesweep scheme): a workgroup takes a tile of
// LG_SORT_BLOCK x LG_SORT_ITEMS keys in ticket order, ranks them by digit, publishes its 256 per-digit counts, resolves
// the counts of all earlier tiles by looking back through their published states, and scatters its keys -- reordered in
// LDS first so that each digit's run leaves as contiguous 8-byte stores.  The digit histograms of ALL passes are not
// computed by a pass over the keys: K3 (lg_duplicate) has every key in registers when it emits it and accumulates them
// there (lg_sort_hist is the stand-alone form for other callers).
//
//   ranking      per wave and item (64 consecutive keys): lanes with equal digits find each other through a 64-bit LDS
//                word per (wave, digit) they OR their lane bit into (round 3; 8 ballots + selects before), the lowest of
//                them bumps the wave's 16-bit LDS counter of that digit; rank = old counter + number of equal lanes
//                below.  Key order (wave, item, lane) = index order, so the sort is stable.
//   look-back    thread d owns digit d: state word = 2 flag bits | 30-bit count, one relaxed agent-scope store / load
//                per word (the word is its own payload: no fence).  Tiles are numbered by an atomic ticket, so every
//                predecessor a tile waits for is already running: no deadlock whatever the dispatch order.
//   LDS          keys 8 B x tile + 16-bit counters [waves][256] + three 256-word tables: 74 KB at 1024 x 8 (2 workgroups
//                per CU), 38 KB at 512 x 8.
// n < 2^30 keys (30-bit state payload); the device-side key count and abort flag come from `counters` (lg_host.h) so that
// the capacity-bounded forward (lg_forward_bounded) needs no host round trip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef LG_SORT_BLOCK
#define LG_SORT_BLOCK 1024
#endif
#ifndef LG_SORT_ITEMS
#define LG_SORT_ITEMS 8
#endif
#define LG_SORT_WAVES (LG_SORT_BLOCK / 64)
#define LG_SORT_TILE (LG_SORT_BLOCK * LG_SORT_ITEMS)
#define LG_SORT_MAX_PASSES 8 // 64 key bits / 8
#ifndef LG_SORT_WINDOW
#define LG_SORT_WINDOW 4     // predecessors examined per look-back round trip (16 cost 26 more VGPRs: one tile per CU instead of two)
#endif
#define LG_SORT_FLAG_AGG 1u
#define LG_SORT_FLAG_PREFIX 2u
#define LG_SORT_VALUE_MASK 0x3FFFFFFFu
#define LG_ABORT_SORT 4u     // bit of the view's abort word (counters[0]) / of the stand-alone sort's error word
#ifndef LG_SORT_POLL_BUDGET
#define LG_SORT_POLL_BUDGET (1u << 18)
#endif
static_assert(LG_SORT_BLOCK >= 256 && LG_SORT_BLOCK % 64 == 0, "one thread per digit needs >= 256 threads");
static_assert(LG_SORT_ITEMS * 64 < 65536 && LG_SORT_TILE < 65536, "16-bit LDS counters");
static_assert(LG_SORT_ITEMS * 64 >= 256, "the digit lane masks (waves x 256 x 8 B) live in the staging buffer (waves x items x 64 x 8 B)");

// temp-storage layout: [hist 8 x 256 u32][tickets 8 u32 (64 B)][states passes x tiles x 256 u32][keys_tmp n u64]
struct LgSortLayout {
    size_t hist_off, ticket_off, state_off, keys_tmp_off, total;
    unsigned tiles;
};
static inline LgSortLayout lg_sort_layout(size_t n)
{
    LgSortLayout L{};
    if (n == 0) n = 1;
    L.tiles = (unsigned)((n + LG_SORT_TILE - 1) / LG_SORT_TILE);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += (bytes + 255) / 256 * 256; return r; };
    L.hist_off = take((size_t)LG_SORT_MAX_PASSES * 256 * 4);
    L.ticket_off = take(64);
    L.state_off = take((size_t)LG_SORT_MAX_PASSES * L.tiles * 256 * 4);
    L.keys_tmp_off = take(n * 8);
    L.total = off;
    return L;
}
// bytes at the start of the temp storage that must be zero before a sort with `passes` passes (hist, tickets, states)
static inline size_t lg_sort_clear_bytes(const LgSortLayout& L, unsigned passes)
{
    return L.state_off + (size_t)passes * L.tiles * 256 * 4;
}

// Stand-alone digit histograms of all passes (callers that do not produce the keys themselves; the rasterizer's K3 does).
__global__ void __launch_bounds__(256)
lg_sort_hist(const uint64_t* __restrict__ keys, uint32_t n, int begin_bit, int end_bit, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t lh[LG_SORT_MAX_PASSES * 256];
    const int passes = (end_bit - begin_bit + 7) / 8;
    for (int i = threadIdx.x; i < passes * 256; i += 256) lh[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint64_t k = keys[i];
        for (int p = 0; p < passes; p++) {
            const int bit = begin_bit + 8 * p, nb = min(8, end_bit - bit);
            atomicAdd(&lh[p * 256 + (uint32_t)((k >> bit) & ((1u << nb) - 1u))], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * 256; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// State words cross XCDs (per-XCD L2s are not coherent with each other): they are published with an agent-scope atomic
// exchange and polled with an agent-scope atomic RMW (fetch_add 0) -- read-modify-writes execute at the device's coherence
// point by construction, whatever a plain or sc1 load would be served from.
__device__ __forceinline__ uint32_t lg_ld_state(uint32_t* p) { return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lg_st_state(uint32_t* p, uint32_t v) { (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One digit pass.  counters (may be NULL): [0] != 0 aborts the view (capacity overflow, lg_forward_bounded), [3] = key
// count; with counters == NULL the count is n_arg.  hist = this pass's 256 global digit counts, ticket / states = this
// pass's ticket word and state array (zeroed by the caller's one memset).
// (two 1024-thread tiles per CU need 8 waves per SIMD, i.e. at most 64 VGPRs: requested explicitly)
#ifdef LG_SORT_NO_FORCE
#define LG_SORT_OCC
#else
#define LG_SORT_OCC __attribute__((amdgpu_waves_per_eu(LG_SORT_BLOCK / 128, LG_SORT_BLOCK / 128)))
#endif
__global__ void __launch_bounds__(LG_SORT_BLOCK) LG_SORT_OCC
lg_onesweep_pass(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, const uint32_t* __restrict__ counters, uint32_t n_arg,
                 int shift, int nbits, const uint32_t* __restrict__ hist, uint32_t* ticket, uint32_t* states, uint32_t* err,
                 uint32_t poll_budget, uint2* ranges, int range_shift)
{
    __shared__ uint64_t stage[LG_SORT_TILE];
    __shared__ unsigned short wcnt[LG_SORT_WAVES][256];
    __shared__ uint32_t lbase[256];   // position of digit d's run inside the tile
    __shared__ uint32_t goff[256];    // global position of the run minus lbase (wrap-around arithmetic)
    __shared__ uint32_t wtot[8];      // cross-wave scan scratch: [0..3] tile counts, [4..7] global counts
    __shared__ uint32_t s_tile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (counters && counters[0] != 0u) return;
    const uint32_t n = counters ? counters[3] : n_arg;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (uint32_t i = tid; i < LG_SORT_WAVES * 256; i += LG_SORT_BLOCK) (&wcnt[0][0])[i] = 0;
    for (uint32_t i = tid; i < LG_SORT_WAVES * 256; i += LG_SORT_BLOCK) stage[i] = 0ull;   // the waves' digit lane masks (ranking, below)
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t tile_base = (uint64_t)tile * LG_SORT_TILE;
    if (tile_base >= n) return;                       // tiles beyond the (device-side) key count: nobody looks back at them
    const uint32_t tile_n = (uint32_t)min((uint64_t)LG_SORT_TILE, (uint64_t)n - tile_base);
    const uint32_t dmask = (1u << nbits) - 1u;

    // ---- load + rank (per wave, item by item) ----
    uint64_t key[LG_SORT_ITEMS];
    uint32_t rnk[LG_SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < LG_SORT_ITEMS; k++) {
        const uint32_t idx = (wave * LG_SORT_ITEMS + k) * 64u + lane;
        key[k] = idx < tile_n ? src[tile_base + idx] : ~0ull;
    }
    // Lanes of one item (64 consecutive keys) with the same digit find each other through LDS: every lane ORs its lane bit into
    // its wave's 64-bit word of that digit (ds_or_b64 -- commutative: the result does not depend on the order the hardware serves
    // the lanes in) and reads the word back.  Round 2 built the same set with 8 ballots + per-lane selects, ~55 VALU instructions
    // per item and almost half of the kernel's issue time.  The words live in `stage`, which is free until the reorder.
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(stage) + wave * 256u;
    const unsigned long long mybit = 1ull << lane;
#pragma unroll
    for (int k = 0; k < LG_SORT_ITEMS; k++) {
        const uint32_t idx = (wave * LG_SORT_ITEMS + k) * 64u + lane;
        const bool valid = idx < tile_n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
        if (valid) atomicOr(&mask[d], mybit);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long peers = 0;
        uint32_t prev = 0;
        if (valid) { peers = mask[d]; prev = wcnt[wave][d]; }               // every peer reads the set and the old count ...
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0u) {                                         // ... the lowest one bumps the count and clears the set
            wcnt[wave][d] = (unsigned short)(prev + (uint32_t)__popcll(peers));
            mask[d] = 0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rnk[k] = prev + below;
    }
    __syncthreads();

    // ---- per-digit: scan over the waves, look-back over the tiles, scans over the digits ----
    uint32_t run = 0, gh = 0;
    if (tid < 256) {
#pragma unroll
        for (int w0 = 0; w0 < LG_SORT_WAVES; w0 += 4) {                     // four reads in flight, then their running sum
            uint32_t cw[4];
#pragma unroll
            for (int w = 0; w < 4; w++) cw[w] = wcnt[w0 + w][tid];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                wcnt[w0 + w][tid] = (unsigned short)run;                   // exclusive over the waves of this tile
                run += cw[w];
            }
        }
        gh = hist[tid];
        // publish the tile's aggregate before looking back: nobody ever waits for more than this store
        if (tile > 0) lg_st_state(&states[(size_t)tile * 256 + tid], (LG_SORT_FLAG_AGG << 30) | run);
        // exclusive scans over the 256 digits of (run, gh): wave scan + cross-wave carry through LDS
        uint32_t ir = run, ig = gh;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t orr = __shfl_up(ir, s, 64), og = __shfl_up(ig, s, 64);
            if ((int)lane >= s) { ir += orr; ig += og; }
        }
        if (lane == 63u) { wtot[wave] = ir; wtot[4 + wave] = ig; }
        lbase[tid] = ir - run;                                              // wave-local exclusive; carry added below
        goff[tid] = ig - gh;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t cr = 0, cg = 0;
        for (uint32_t w = 0; w < wave; w++) { cr += wtot[w]; cg += wtot[4 + w]; }
        const uint32_t lb = lbase[tid] + cr;                                // tile-local start of digit tid
        const uint32_t gb = goff[tid] + cg;                                 // global start of digit tid (all tiles)
        uint32_t excl = 0;
        if (tile > 0) {
            // Windowed look-back.  All tiles of a pass are co-resident and publish their aggregates at about the same time, so
            // a tile that walks back ONE predecessor per memory round trip needs ~sqrt(2 * tiles) dependent round trips
            // (~30 at 500 tiles, ~1 us each: most of a pass).  Reading LG_SORT_WINDOW predecessors per round trip (independent
            // loads, all in flight) cuts that to ~sqrt(2 * tiles / window).  States are consumed nearest-first; a zero
            // (unpublished) state is re-polled alone.  (The TOTAL number of polls of a thread is bounded: a predecessor
            // always publishes -- ticket order -- so the bound is never reached; if it ever is, the would-be hang of the device
            // becomes a REPORTED failure: bit LG_ABORT_SORT of *err (the view's abort word counters[0]: lg_tile_sort / lg_tile_ranges and the
            // blend kernels then leave the view empty, the status words / LG_FLAG_DEBUG / lg_view_status() surface LG_ERR_DEVICE).)
            uint32_t budget = poll_budget;
            int64_t b = (int64_t)tile - 1;
            bool found = false;
            while (b >= 0 && !found) {
                uint32_t win[LG_SORT_WINDOW];
#pragma unroll
                for (int k = 0; k < LG_SORT_WINDOW; k++)
                    win[k] = (b - k >= 0) ? lg_ld_state(&states[(size_t)(b - k) * 256 + tid]) : (LG_SORT_FLAG_PREFIX << 30);
#pragma unroll
                for (int k = 0; k < LG_SORT_WINDOW; k++) {
                    if (!found) {
                        uint32_t sv = win[k];
                        while ((sv >> 30) == 0u && budget > 0u) {
                            budget--;
                            __builtin_amdgcn_s_sleep(2);
                            sv = lg_ld_state(&states[(size_t)(b - k) * 256 + tid]);
                        }
                        excl += sv & LG_SORT_VALUE_MASK;
                        if ((sv >> 30) == 0u) atomicOr(err, LG_ABORT_SORT);       // budget exhausted: this sort's result is void
                        found = (sv >> 30) == LG_SORT_FLAG_PREFIX || budget == 0u;
                    }
                }
                b -= LG_SORT_WINDOW;
            }
        }
        lg_st_state(&states[(size_t)tile * 256 + tid], (LG_SORT_FLAG_PREFIX << 30) | ((excl + run) & LG_SORT_VALUE_MASK));
        lbase[tid] = lb;
        goff[tid] = gb + excl - lb;
    }
    __syncthreads();

    // ---- reorder inside the tile (LDS), then store each digit's run contiguously ----
#pragma unroll
    for (int k = 0; k < LG_SORT_ITEMS; k++) {
        const uint32_t idx = (wave * LG_SORT_ITEMS + k) * 64u + lane;
        if (idx < tile_n) {
            const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
            stage[lbase[d] + (uint32_t)wcnt[wave][d] + rnk[k]] = key[k];
        }
    }
    __syncthreads();
    for (uint32_t q = tid; q < tile_n; q += LG_SORT_BLOCK) {
        const uint64_t kq = stage[q];
        const uint32_t d = (uint32_t)(kq >> shift) & dmask;
        const uint32_t pos = goff[d] + q;
        dst[(size_t)pos] = kq;
        if (ranges) {
            // LAST pass of a sort whose top field (key >> range_shift) is the rasterizer's tile id: the keys of one id are
            // contiguous in `stage` (digit = its top bits, the passes before ordered the rest) and in the output, so the ends of
            // each id's stretch in THIS sort tile bound its range from inside; min / max over the sort tiles give {begin, end}.
            // ranges[] starts as {0xFFFFFFFF, 0} (lg_duplicate); ids without a key keep that (lg_tile_sort turns it into {0, 0}).
            const uint32_t t = (uint32_t)(kq >> range_shift);
            if (q == 0u || (uint32_t)(stage[q - 1u] >> range_shift) != t) atomicMin(&ranges[t].x, pos);
            if (q + 1u == tile_n || (uint32_t)(stage[q + 1u] >> range_shift) != t) atomicMax(&ranges[t].y, pos + 1u);
        }
    }
}

// Stable ascending sort of bits [begin_bit, end_bit) of the 64-bit keys.  keys_in is preserved, the result lands in keys_out.
// temp == NULL: size query.  hist_ready: the caller already accumulated the digit histograms into temp (+ L.hist_off) AFTER
// clearing lg_sort_clear_bytes() bytes of temp -- the rasterizer's K3 does; otherwise this function clears and counts.
// counters: optional device words {[0] abort flag, [3] key count} (see lg_onesweep_pass); n = capacity = upper bound of the count.
// err: device word that receives LG_ABORT_SORT when a look-back gives up (NULL: counters[0], or -- without counters -- word 15 of
// the ticket block, which lg_debug_sort_keys reads back).
// ranges != NULL: the last pass also writes, per value t of key >> range_shift, {first, last + 1} output position into ranges[t]
// (see lg_onesweep_pass; the caller initialised ranges[] to {0xFFFFFFFF, 0}).
static hipError_t lg_sort_keys(void* temp, size_t& temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, uint32_t n, int begin_bit,
                               int end_bit, uint32_t* counters, bool hist_ready, hipStream_t stream, uint32_t poll_budget = LG_SORT_POLL_BUDGET,
                               uint2* ranges = nullptr, int range_shift = 0)
{
    const LgSortLayout L = lg_sort_layout(n);
    if (temp == nullptr) { temp_bytes = L.total; return hipSuccess; }
    if (n == 0) return hipSuccess;
    if (temp_bytes < L.total || n >= (1u << 30) || end_bit <= begin_bit || end_bit > 64) return hipErrorInvalidValue;
    const int passes = (end_bit - begin_bit + 7) / 8;
    char* base = (char*)temp;
    uint32_t* hist = (uint32_t*)(base + L.hist_off);
    uint32_t* tickets = (uint32_t*)(base + L.ticket_off);
    uint32_t* states = (uint32_t*)(base + L.state_off);
    uint64_t* keys_tmp = (uint64_t*)(base + L.keys_tmp_off);
    hipError_t e;
    if (!hist_ready) {
        if ((e = lg_zero_async(base, lg_sort_clear_bytes(L, passes), stream)) != hipSuccess) return e;
        const unsigned hb = (unsigned)std::min<size_t>(((size_t)n + 2047) / 2048, 1024);
        lg_sort_hist<<<hb, 256, 0, stream>>>(keys_in, n, begin_bit, end_bit, hist);
    }
    bool to_output = (passes - 1) % 2 == 0;          // ping-pong so that the LAST pass writes keys_out
    const uint64_t* src = keys_in;
    for (int p = 0; p < passes; p++) {
        const int bit = begin_bit + 8 * p, nb = std::min(8, end_bit - bit);
        uint64_t* dst = to_output ? keys_out : keys_tmp;
        lg_onesweep_pass<<<L.tiles, LG_SORT_BLOCK, 0, stream>>>(src, dst, counters, n, bit, nb, hist + (size_t)p * 256, tickets + p,
                                                                 states + (size_t)p * L.tiles * 256, counters ? counters : tickets + 15,
                                                                 poll_budget, p == passes - 1 ? ranges : nullptr, range_shift);
        src = dst;
        to_output = !to_output;
    }
    return hipGetLastError();
}
