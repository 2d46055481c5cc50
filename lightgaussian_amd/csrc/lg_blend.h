// lg_blend.h -- per-tile kernels: K6 lg_blend_fwd (+count), score kernel, K7 lg_blend_bwd and their pair steps / reductions
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"
#include "lg_preprocess.h" // LG_ID_MASK
#include "lg_binning.h" // lg_slot_of
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// tile <-> workgroup mapping.  Workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only).
// Neighbouring tiles share Gaussians, so each XCD (own 4 MB L2) is given runs of 4 consecutive tiles, interleaved with the other XCDs' runs
// over the whole image (a contiguous band per XCD and other run lengths: EXPERIMENTS.md, "tile -> XCD map").
// The map is a bijection on [0, ntiles_pad) for ntiles_pad a multiple of 32; callers guard tile < ntiles.
#ifndef LG_XCD_RUN_LOG2
#define LG_XCD_RUN_LOG2 2 // runs of 4 consecutive tiles per XCD
#endif
#define LG_TILE_GRID_ALIGN (8 << LG_XCD_RUN_LOG2)
__device__ __forceinline__ int xcd_tile(int b, int ntiles_pad)
{
    (void)ntiles_pad;
    const int r = b >> 3; // index of this workgroup inside its XCD's stream
    return ((r >> LG_XCD_RUN_LOG2) << (LG_XCD_RUN_LOG2 + 3)) + ((b & 7) << LG_XCD_RUN_LOG2) + (r & ((1 << LG_XCD_RUN_LOG2) - 1));
}

#define LG_Q 64 // LDS queue depth per wave = one batch

// Threshold guard of the hardware-exp variants.  alpha >= 1/255 is a discontinuity of the algorithm: a pair
// that flips in or out changes its pixel by up to T/255.  v_exp_f32 and the canonical lg_exp differ by a few 1e-7
// relative, so on the rare lanes whose alpha lies within 5e-6 (relative) of the threshold the canonical value is
// used instead (wave-uniform branch, taken for ~1e-5 of the evaluations).  Every include/exclude decision of the
// fast path then equals the canonical path's, forward and backward alike.
__device__ __forceinline__ float guard_alpha(float alpha, float opacity, float power_le0 /* clamped to <= 0 here */)
{
    const bool near = fabsf(alpha - LG_ALPHA_MIN) < 2.0e-8f;
    if (__ballot(near) != 0) {
        const float ac = fminf(LG_ALPHA_MAX, opacity * lg_exp(fminf(power_le0, 0.0f)));
        alpha = near ? ac : alpha;
    }
    return alpha;
}
// (the guard folded into the threshold test: EXPERIMENTS.md, "K6 / K7 pair step")

// Does the 8 x 8 pixel block [x0, x0 + 7] x [y0, y0 + 7] hold a pixel the splat can reach (alpha >= 1/255)?
// First the axis-aligned box of the alpha >= 1/255 ellipse (hx, hy from lg_project; inf = culling off), then the ellipse
// itself: power(dx, dy) = ha dx^2 + nb dx dy + hc dy^2 is concave, so its maximum over the block is 0 when the centre is
// inside and otherwise lies on one of the (at most two) block edges that face the centre -- two clamped 1-D maximisations.
// A block is dropped only when that maximum is below -(ln(255 opacity) + margin): the margin (1e-3 in the exponent, far
// above the rounding of this estimate and of the per-pixel evaluation, far below anything visible -- it only keeps blocks)
// makes the test conservative, so every include / exclude decision per pixel is unchanged.  On the benchmark scene the box
// passes 1.81 blocks per (tile, splat), the ellipse 1.47 (diagonal, elongated splats): a fifth of the pair evaluations.
struct LgReach { float tau, r2ha, r2hc; bool cull; };
__device__ __forceinline__ LgReach lg_reach(const float4& r0, const float4& r1, const float4& r2)
{
    LgReach e;
    e.cull = r2.y < 1.0e30f;
    e.tau = __logf(255.0f * r1.y) + 1.0e-3f;
    e.r2ha = __builtin_amdgcn_rcpf(2.0f * r0.z);
    e.r2hc = __builtin_amdgcn_rcpf(2.0f * r1.x);
    return e;
}
// ext = last pixel offset of the (square) block: 7 for the 8 x 8 blocks of K7, 3 for the 4 x 4 blocks of K6's quarter-wave walk.
__device__ __forceinline__ bool lg_block_hit(const float4& r0, const float4& r1, const float4& r2, const LgReach& e, float x0, float y0,
                                             float ext = 7.0f)
{
    const float x1 = x0 + ext, y1 = y0 + ext;
    const bool box = (r0.x + r2.y >= x0) && (r0.x - r2.y <= x1) && (r0.y + r2.z >= y0) && (r0.y - r2.z <= y1);
    const float ha = r0.z, nb = r0.w, hc = r1.x;
    const float dxe = fminf(fmaxf(r0.x, x0), x1) - r0.x, dye = fminf(fmaxf(r0.y, y0), y1) - r0.y;   // nearest point of the block
    // on the vertical line through the nearest point: dy* = -nb dxe / (2 hc), clamped to the block
    const float dy1 = fminf(fmaxf(r0.y - nb * dxe * e.r2hc, y0), y1) - r0.y;
    const float p1 = (ha * dxe + nb * dy1) * dxe + hc * dy1 * dy1;
    const float dx2 = fminf(fmaxf(r0.x - nb * dye * e.r2ha, x0), x1) - r0.x;
    const float p2 = (ha * dx2 + nb * dye) * dx2 + hc * dye * dye;
    const bool reach = fmaxf(p1, p2) >= -e.tau;
    return box && (reach || !e.cull);
}

// Select-based (no divergent control flow) front-to-back step of one list entry for the 64 pixels of a wave.  The lanes' `done` state
// (saturated or outside the image) is ONE scalar lane mask: every compare is balloted, the masks are combined as 64-bit scalars and the
// contributing set goes back to a lane predicate through llvm.amdgcn.inverse.ballot (no instruction) -- 9 scalar instructions per pair step
// where a loop-carried per-lane bool cost 13 (EXPERIMENTS.md, "K6 / K7 pair step").  Rejected lanes compute and discard; contributing lanes
// run the canonical operations of the oracle, so results are bit-identical.  Returns the mask of the lanes the entry contributed to.
template <bool EXACT, bool COLOR = true>
__device__ __forceinline__ uint64_t fwd_pair_m(const float4& a, const float4& b, const float4& c, uint64_t& donem, float pxf, float pyf, float& T,
                                               float& C0, float& C1, float& C2, uint32_t& last, uint32_t rel, float& alpha_out, float& w_out)
{
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy);
    const float ex = EXACT ? lg_exp(fminf(power, 0.0f)) : __expf(power);
    float alpha = fminf(LG_ALPHA_MAX, b.y * ex);
    if (!EXACT) alpha = guard_alpha(alpha, b.y, power);
    const uint64_t okm = (__builtin_amdgcn_ballot_w64(power <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= LG_ALPHA_MIN)) & ~donem;
    const float test_T = T * (1.0f - alpha);
    const uint64_t satm = okm & __builtin_amdgcn_ballot_w64(test_T < LG_T_MIN);
    const uint64_t cm = okm ^ satm;
    donem |= satm;
    const bool contrib = __builtin_amdgcn_inverse_ballot_w64(cm);
    if (COLOR) {
        const float w = contrib ? alpha * T : 0.0f;
        C0 = fmaf(b.z, w, C0); C1 = fmaf(b.w, w, C1); C2 = fmaf(c.x, w, C2);
        w_out = w;
    }
    T = contrib ? test_T : T;
    last = contrib ? rel : last;
    alpha_out = alpha;
    return cm;
}

// Significance weights that differ from hit to hit (LG_W_ALPHA: alpha, LG_W_ALPHA_T: alpha T) are accumulated in 64-bit FIXED POINT,
// Q24.40: every hit's fp32 weight w (3.9e-7 <= w <= 0.99) is rounded to the nearest multiple of 2^-40 (ties to even; exact for
// w >= 2^-17) and the integers are added -- associative, so the per-view sum does not depend on the order the waves' atomics land in:
// bit-reproducible run to run, equal to the oracle's sequential loop, and any partition of the views over ranks gives the same scores.
// The quantisation is one double-precision add: (double) w + 4096 has its unit in the last place at 2^-40, so the low 40 bits of the
// sum's mantissa ARE round(w 2^40); adding the bit patterns as integers and subtracting n x bits(4096.0) leaves the sum of the n
// quantised weights (no carry can reach the exponent: 64 x 2^40 < 2^52).
// (lg_fix40_bits / LG_FIX_MAGIC / lg_fix40_score: lg_math.h, pinned on the CPU by tests/test_math_harness.py)
//
// The 64 x (entries) matrix of a wave's weights is transposed through LDS, 8 entries at a time: every pair step stores one row of fp32
// weights (ds_write_b32, 64 lanes; 0 where the pixel does not contribute); after 8 rows lane 8 r + s quantises and adds columns 8 s ..
// 8 s + 7 of row r (two ds_read_b128, 8 x {cvt, add}, 7 integer adds) and three DPP steps inside each group of 8 lanes (quad xor 1, quad
// xor 2, half-row mirror) leave the row total in all 8: ~4 vector instructions per entry where a DPP wave reduction per entry costs 7 (+ a
// readlane), and no cross-lane traffic in the pair step itself.  Lane 8 r + c keeps the total of chunk c: lane l ends up owning compacted
// entry 8 (l & 7) + (l >> 3) of the batch (LG_WQ_OWNER) -- its hit count (ballot popcount of the pair step) AND its weight total.
//
// Where the totals go (round 6, measured -- tools/ubench/atomic_rate.hip, profiles/r06_call_b.log, EXPERIMENTS.md Part II section 1): a
// random-address global atomic costs the device ~40 ns of its atomic pipe whatever its width (4.8 M per view = 0.19 ms: hidden behind the
// 0.3 ms of arithmetic when there is ONE per (wave, entry), as in the integer-weight variant; a second one per (wave, entry) -- count[id] and a
// 64-bit sum[id] -- made the kernel atomic-bound, 0.33 -> 0.61 ms).  So count and weight total travel in ONE 64-bit word: per (tile,
// Gaussian) INSTANCE the ranges are small -- at most 256 hits, weight below 256 -- and {count : 16 | Q8.40 : 48} fits.  The word lives at the
// instance's pre-sort slot (lg_slot_of: a closed form of the Gaussian's tile rectangle; the radix sort's input buffer, free by now, holds the
// slots) and lg_score_slots sums every Gaussian's consecutive slots into out_count / out_score -- no atomics there.
//   * significance-only pass (no colour: the variant the prune pass runs): the four waves of a tile MERGE in LDS first -- a ring of
//     LG_TM_SLOTS batch accumulators (ds_add_u64 per (wave, entry)), an arrival counter per batch, and the last wave to leave a batch
//     writes its 64 words with plain stores (tile_leave in lg_blend_fwd).  One store per instance, zeros included: no global atomic, no
//     clear of the slots, one binning-record gather per (tile, entry) instead of one per (wave, entry).  0.440 -> 0.400 ms, 1352 -> 1448 views/s.
//     The waves stay autonomous: nobody waits at a batch except for a ring slot whose flush is still in progress.
//   * count forwards that also return the image: one atomic per (wave, entry) into the cleared slots.
#define LG_WQ_ROWS 8
#define LG_WQ_STRIDE 68                               // row stride in floats (64 lanes + 16 bytes of skew)
#define LG_WQ_OWNER(lane) (8u * ((lane) & 7u) + ((lane) >> 3))
#define LG_SLOT_COUNT_SHIFT 48
#ifndef LG_TM_SLOTS
#define LG_TM_SLOTS 4
#endif                                                // batches of a tile that may be "open" at once (waves of a tile drift apart by at most this many)
__device__ __forceinline__ uint64_t lg_wq_rowsum(const float* wq, uint32_t lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float4* src = reinterpret_cast<const float4*>(wq + (lane >> 3) * LG_WQ_STRIDE + (lane & 7u) * 8u);
    const float4 x0 = src[0], x1 = src[1];
    // this lane's eight columns; modulo 2^64, minus 8 x bits(4096.0): exactly the sum of 8 quantised weights, below 2^43
    const uint64_t s = (((lg_fix40_bits(x0.x) + lg_fix40_bits(x0.y)) + (lg_fix40_bits(x0.z) + lg_fix40_bits(x0.w))) +
                        ((lg_fix40_bits(x1.x) + lg_fix40_bits(x1.y)) + (lg_fix40_bits(x1.z) + lg_fix40_bits(x1.w)))) - 8ull * LG_FIX_MAGIC;
    // across the 8 lanes of the row in two 32-bit limbs (24 + 19 bits: eight of them cannot carry), so that each step is ONE
    // v_add_u32_dpp per limb (hipcc has no 64-bit DPP add: two v_mov_b32_dpp + two 64-bit adds per step)
    uint32_t lo = (uint32_t)s & 0xFFFFFFu, hi = (uint32_t)(s >> 24);
    lo = dpp_add_u32<0xB1>(lo); hi = dpp_add_u32<0xB1>(hi);          // quad_perm [1,0,3,2]
    lo = dpp_add_u32<0x4E>(lo); hi = dpp_add_u32<0x4E>(hi);          // quad_perm [2,3,0,1]
    lo = dpp_add_u32<0x141>(lo); hi = dpp_add_u32<0x141>(hi);        // row_half_mirror: lane i <-> 7 - i of each group of 8 (the other quad)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                // (the next chunk overwrites the rows)
    return ((uint64_t)hi << 24) + lo;
}

// K6 / K6c: forward blend.  FSCORE: 0, or the per-hit weight policy (LG_W_ALPHA / LG_W_ALPHA_T) accumulated into fix[] (Q24.40)
template <bool COUNT, int FSCORE, bool EXACT, bool COLOR = true>
__global__ void __launch_bounds__(256)
lg_blend_fwd(int W, int H, int gx, int ntiles, int ntiles_pad, const uint2* __restrict__ ranges,
             const uint64_t* __restrict__ entries, uint32_t gid_mask, const float4* __restrict__ rec, const float* __restrict__ bg,
             float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             int32_t* __restrict__ count, unsigned long long* __restrict__ slots, const uint4* __restrict__ tinfo, uint32_t slot_cap,
             int S, float4* __restrict__ ckpt,
             uint2* __restrict__ work, uint32_t* __restrict__ meta, uint2* __restrict__ par_work, const uint32_t* __restrict__ counters,
             int long_mode, uint32_t* __restrict__ par_arrived)
{
    static_assert(FSCORE == 0 || ((FSCORE == LG_W_ALPHA || FSCORE == LG_W_ALPHA_T) && COUNT), "FSCORE is 0 or a per-hit weight policy of the count variant");
    __shared__ float4 q0[4][LG_Q], q1[4][LG_Q], q2[4][LG_Q];
    __shared__ __attribute__((aligned(16))) float wq[FSCORE ? 4 : 1][FSCORE ? LG_WQ_ROWS * LG_WQ_STRIDE : 1];
    // Significance-only pass with per-hit weights (MERGE): the four waves of a tile add their per-entry {count | weight} words into a ring of
    // LG_TM_SLOTS batch accumulators in LDS; whichever wave is the LAST to pass a batch writes the batch's 64 words to the instances' slots
    // with plain stores -- one store per (tile, Gaussian) instance, zeros included, so the slots need neither atomics nor a clear.
    constexpr bool MERGE = FSCORE != 0 && !COLOR;
    __shared__ unsigned long long tacc[MERGE ? LG_TM_SLOTS : 1][MERGE ? LG_Q : 1];
    __shared__ uint32_t tarr[MERGE ? LG_TM_SLOTS : 1];      // waves that have passed the batch in the slot
    __shared__ uint32_t tgen[MERGE ? LG_TM_SLOTS : 1];      // flushes the slot has seen: batch b may use slot b % LG_TM_SLOTS once tgen == b / LG_TM_SLOTS
    // lists longer than par_min (when non-zero) are left to the parallel long-tile kernels below: a pure function of this
    // view's own numbers (counters[3] = its instance count), evaluated identically by every workgroup
    const uint32_t par_min = lg_par_min(long_mode, S, counters[3], ntiles);
    if (blockIdx.x == (uint32_t)ntiles_pad) {
        // the one workgroup past the tiles: work list of the backward blend (lg_binning.h), overlapped with the blending
        uint32_t* scratch = reinterpret_cast<uint32_t*>(&q0[0][0]);
        lg_work_order_body(ntiles, S, ranges, work, meta, scratch, scratch + 256, threadIdx.x, 256, par_work, par_min, par_arrived);
        return;
    }
    // (quarter-wave walk, longest-list-first dispatch: EXPERIMENTS.md, "K6 structure")
    const int tile = xcd_tile(blockIdx.x, ntiles_pad);
    if (tile >= ntiles) return;
    const int wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int wx0 = tx * LG_TILE + (wave & 1) * 8, wy0 = ty * LG_TILE + (wave >> 1) * 8;
    const int pxi = wx0 + (int)(lane & 7), pyi = wy0 + (int)(lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float bx0 = (float)wx0, bx1 = (float)(wx0 + 7), by0 = (float)wy0, by1 = (float)(wy0 + 7);
    const uint2 range = ranges[tile];
    const uint32_t pix_in_tile = (uint32_t)(pyi - ty * LG_TILE) * 16u + (uint32_t)(pxi - tx * LG_TILE);   // checkpoint slot of this pixel

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    uint64_t donem = ~__builtin_amdgcn_ballot_w64(inside);    // saturated or outside the image: one scalar mask per wave (fwd_pair_m)
    // Long list (more than one segment of S entries): leave a checkpoint record per pixel at the end of every segment --
    // {T there, colour accumulated INSIDE the segment (absolute weights alpha T: a sum of non-negative terms, no
    // cancellation)} -- from which the backward starts each segment independently (lg_blend_bwd).  Record j of this tile is
    // ckpt[(2 (range.x / S) + j) * 256 + pixel]; 2 floor(x / S) leaves room for ceil(n / S) records before the next long tile.
    const bool longt = COLOR && (range.y - range.x) > (uint32_t)S;            // block-uniform
    // lists longer than par_min (when set): their segments are walked in parallel by lg_blend_fwd_seg / _scan / _rewalk (below) --
    // or, in the significance-only pass (no colour: par_min is only non-zero there for the integer weights), by lg_count_seg / _rewalk / _fixup
    if ((longt || !COLOR) && par_min != 0u && (range.y - range.x) > par_min) return;
    if (MERGE) {
        for (uint32_t i = threadIdx.x; i < LG_TM_SLOTS * LG_Q; i += 256u) (&tacc[0][0])[i] = 0ull;
        if (threadIdx.x < LG_TM_SLOTS) { tarr[threadIdx.x] = 0u; tgen[threadIdx.x] = 0u; }
        __syncthreads();
    }
    // MERGE: batch `bi` of the tile is left behind by this wave -- wait until its ring slot is free (the flush of batch bi - LG_TM_SLOTS has
    // happened: every wave passed that batch long ago, so the wait is for a flush in progress at most), add what `add()` has, then arrive; the
    // fourth wave to arrive owns the flush.  A wave that stops early (all its pixels saturated) still arrives at every remaining batch.
    auto tile_leave = [&](uint32_t bi, auto add) {
        const uint32_t ts = bi % LG_TM_SLOTS, want = bi / LG_TM_SLOTS;
        while (__hip_atomic_load(&tgen[ts], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        add(ts);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t old = 0;
        if (lane == 0u) old = __hip_atomic_fetch_add(&tarr[ts], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        if (old == 3u) {                                             // wave-uniform: the last of the tile's four waves
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t first = range.x + bi * LG_Q;
            const unsigned long long v = tacc[ts][lane];
            tacc[ts][lane] = 0ull;
            if (first + lane < range.y) {
                const uint32_t id = (uint32_t)entries[first + lane] & gid_mask;
                const uint32_t sl = lg_slot_of(tinfo[id], tx, ty);
                if (sl < slot_cap) slots[sl] = v;                    // exactly one store per instance of the view: no atomics, no clear
            }
            if (lane == 0u) tarr[ts] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0u) __hip_atomic_store(&tgen[ts], want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // the walk exists twice: tiles of one segment (every tile of the uniform benchmark scene) run the LONG = false copy,
    // which carries neither the segment accumulators nor the boundary test
    auto walk = [&](auto long_tag) {
        constexpr bool LONG = decltype(long_tag)::value;
        float Cs0 = 0.0f, Cs1 = 0.0f, Cs2 = 0.0f;
        uint32_t seg = 0;
        float4* ck = nullptr;
        if (LONG) ck = ckpt + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix_in_tile;
        // (software pipeline over the batches: EXPERIMENTS.md, "K6 structure")
        for (uint32_t base = range.x; base < range.y; base += LG_Q) {
            if (LONG && base != range.x && (base - range.x) % (uint32_t)S == 0u) {
                ck[(size_t)seg * 256] = make_float4(T, Cs0, Cs1, Cs2);
                seg++; Cs0 = Cs1 = Cs2 = 0.0f;
            }
            if (~donem == 0ull) {            // every pixel of this wave is saturated or outside
                if (MERGE) for (uint32_t b = (base - range.x) / LG_Q; b * LG_Q < range.y - range.x; b++) tile_leave(b, [](uint32_t) {});
                break;
            }
            const uint32_t idx = base + lane;
            bool hit = false;
            float4 r0, r1, r2;
            if (idx < range.y) {
                const uint32_t id = (uint32_t)entries[idx] & gid_mask;
                r0 = rec[LG_REC_F4 * (size_t)id]; r1 = rec[LG_REC_F4 * (size_t)id + 1]; r2 = rec[LG_REC_F4 * (size_t)id + 2];
                // footprint box (x +- hx, y +- hy) vs this wave's 8x8 pixel block; hx = inf when culling is off
                hit = lg_block_hit(r0, r1, r2, lg_reach(r0, r1, r2), bx0, by0);
            }
            const uint64_t mask = __ballot(hit);
            if (mask == 0) {
                if (MERGE) tile_leave((base - range.x) / LG_Q, [](uint32_t) {});
                continue;
            }
            if (hit) {
                const uint32_t pos = prefix_popc(mask);
                // the queue copy carries the entry's contributor index (1-based position in the tile's list) where the record has
                // the box half-extent hx, which nothing reads after the block test: it arrives with the colour in ONE ds_read_b64,
                // and the walk below is a counted loop (was: pop the lowest set bit of the hit mask per step -- s_ff1, two 64-bit
                // scalar ops, an add and a v_mov per hit)
                r2.y = __uint_as_float(idx - range.x + 1u);
                // per-hit weights: the entry's pre-sort slot rides where the box half-extent hy was (read by nothing after the block test)
                if (FSCORE && !MERGE) r2.z = __uint_as_float(lg_slot_of(tinfo[__float_as_uint(r2.w) & LG_ID_MASK], tx, ty));
                q0[wave][pos] = r0; q1[wave][pos] = r1; q2[wave][pos] = r2;
            }
            __builtin_amdgcn_wave_barrier();
            int mycnt = 0;
            uint64_t myfix = 0ull;
            const uint32_t nhit = (uint32_t)__popcll(mask);
            const uint32_t owned = FSCORE ? LG_WQ_OWNER(lane) : lane;       // the compacted entry whose totals this lane collects
            auto pair = [&](uint32_t j, float* wrow) {
                const float4 a = q0[wave][j], b = q1[wave][j], c = q2[wave][j];
                float alpha = 0.0f, Tprev = T, w = 0.0f;
                const uint64_t cm = fwd_pair_m<EXACT, COLOR>(a, b, c, donem, pxf, pyf, T, C0, C1, C2, last, __float_as_uint(c.y), alpha, w);
                const bool res = __builtin_amdgcn_inverse_ballot_w64(cm);
                if (LONG) { Cs0 = fmaf(b.z, w, Cs0); Cs1 = fmaf(b.w, w, Cs1); Cs2 = fmaf(c.x, w, Cs2); }
                if (COUNT) {
                    if (owned == j) mycnt = (int)__popcll(cm);
                    if (FSCORE) *wrow = res ? (FSCORE == LG_W_ALPHA ? alpha : alpha * Tprev) : 0.0f;
                }
            };
            // (two pair steps per loop trip: EXPERIMENTS.md, "K6 / K7 pair step")
            if (FSCORE) {
                // chunks of LG_WQ_ROWS pair steps, each leaving one row of quantised weights; then the rows are summed (lg_wq_rowsum).  (Rows past
                // the end of a last, partial chunk hold older weights: their totals go to lanes whose entry is >= nhit, which issue nothing.)
                for (uint32_t c0 = 0; c0 < nhit; c0 += LG_WQ_ROWS) {
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)min((uint32_t)LG_WQ_ROWS, nhit - c0));
                    float* wrow = &wq[wave][lane];
                    for (uint32_t jj = 0; jj < m; jj++, wrow += LG_WQ_STRIDE) pair(c0 + jj, wrow);
                    const uint64_t tot = lg_wq_rowsum(wq[wave], lane);
                    if ((lane & 7u) == (c0 >> 3)) myfix = tot;
                }
            } else {
                for (uint32_t j = 0; j < nhit; j++) pair(j, nullptr);
            }
            if (COUNT) {
                // lane j owns compacted entry j: one atomic per (wave, Gaussian), issued 64-wide (their cost: EXPERIMENTS.md, "significance pass")
                if (!FSCORE) {
                    if (lane < nhit && mycnt > 0) {
                        const uint32_t id = __float_as_uint(q2[wave][lane].w) & LG_ID_MASK;
                        atomicAdd(&count[id], mycnt);
                    }
                } else if (MERGE) {
                    // {count : 16 | Q8.40 : 48} of this wave's 8 x 8 block into the tile's accumulator of the entry (its position in the batch)
                    tile_leave((base - range.x) / LG_Q, [&](uint32_t ts) {
                        if (owned < nhit && mycnt > 0) {
                            const uint32_t src = __float_as_uint(q2[wave][owned].y) - 1u - (base - range.x);
                            atomicAdd(&tacc[ts][src], ((unsigned long long)(uint32_t)mycnt << LG_SLOT_COUNT_SHIFT) + myfix);
                        }
                    });
                } else if (owned < nhit && mycnt > 0) {
                    // count forwards that also return the image: one atomic per (wave, entry) into the (tile, Gaussian) instance's slot
                    const uint32_t slot = __float_as_uint(q2[wave][owned].z);
                    if (slot < slot_cap) atomicAdd(&slots[slot], ((unsigned long long)(uint32_t)mycnt << LG_SLOT_COUNT_SHIFT) + myfix);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (LONG) {
            // the current segment's record, and -- when the wave stopped early -- those of the segments it never entered
            // (nothing contributed there: T stays, colour 0), so that every record of the tile is valid for every pixel
            const uint32_t nseg = (range.y - range.x + (uint32_t)S - 1u) / (uint32_t)S;
            for (; seg < nseg; seg++) {
                ck[(size_t)seg * 256] = make_float4(T, Cs0, Cs1, Cs2);
                Cs0 = Cs1 = Cs2 = 0.0f;
            }
        }
    };
    if (longt) walk(std::true_type{}); else walk(std::false_type{});
    if (COLOR && inside) {   // !COLOR: forward-only significance pass, nothing per pixel is kept
        const size_t pid = (size_t)pyi * W + pxi, HW = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = fmaf(T, bg[0], C0);
        out_color[HW + pid] = fmaf(T, bg[1], C1);
        out_color[2 * HW + pid] = fmaf(T, bg[2], C2);
    }
}

// ------------------------------------------------------------------------------------------------
// Long tiles of the hardware-exp colour forward, segments in parallel.  The walk of a 24 000-entry list is a serial
// chain of ~0.65 ms for the one workgroup that owns the tile.  Everything but the early termination is independent of what came
// before: with T_local starting at 1, a segment yields per pixel P = prod (1 - alpha) and C = sum alpha T_local c over its
// contributing entries, and the sequential result is T = prod P_s, colour = sum_s (prod_{s' < s} P_s') C_s.
//   pass 1  lg_blend_fwd_seg : one workgroup per (long tile, segment) from the backward's work list walks its S entries with
//           no termination test and leaves {P, C} and the last contributing list position per pixel (in the checkpoint slots);
//   pass 2  lg_scan_tile (run by the workgroup of pass 1 that finishes a tile's last segment), a thread per pixel: prefix products over the segments.  Pixels that
//           never come near the termination threshold are finished (image, final T, n_contrib, the checkpoint records {T at
//           the end of the segment, colour accumulated inside it} the backward starts from); a pixel whose transmittance would
//           fall under the threshold INSIDE segment s* (T P_s* < 1e-4, with a margin) is parked at s*;
//   pass 3  lg_blend_fwd_rewalk: one workgroup per (long tile, segment) again: the pixels parked at this segment walk it
//           sequentially from their true T with the published pair step (fwd_pair) -- exact stop position, exact contributor
//           index -- and are finished there.  Parallel over the segments like pass 1: no tile waits for a serial chain.
// Same include / exclude decisions as the serial walk (alpha tests do not depend on T; termination is resolved by the exact
// re-walk); transmittances are regrouped products, so images agree to float rounding, not bit for bit -- the canonical
// (count / EXACT) path keeps the serial walk.  Which lists: the free-running pass evaluates every entry for every pixel (the
// serial walk stops when its 64 pixels are saturated) and costs three more launches, so it pays only for lists whose serial
// walk would BE the forward's critical path.  "auto" (the default; lg_par_min, lg_binning.h) takes lists longer than two
// segments and four times the view's mean list -- decided on the device from this view's own instance count, no history: the
// three kernels are launched for every hardware-exp colour forward as small persistent grids over the par_work list that the
// forward's work-list workgroup leaves (meta[4] items: zero on the uniform benchmark scene, where each launch is one scalar
// load per workgroup); LG_FLAG_LONG_PARALLEL sends every multi-segment list (tests), LG_FLAG_LONG_SERIAL none.
__device__ __forceinline__ void fwd_free(const float4& a, const float4& b, const float4& c, bool live, float pxf, float pyf, float& T,
                                         float& C0, float& C1, float& C2, uint32_t& last, uint32_t rel)
{
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy);
    float alpha = fminf(LG_ALPHA_MAX, b.y * __expf(power));
    alpha = guard_alpha(alpha, b.y, power);
    const bool ok = live && (power <= 0.0f) && (alpha >= LG_ALPHA_MIN);
    const float w = ok ? alpha * T : 0.0f;
    C0 = fmaf(b.z, w, C0); C1 = fmaf(b.w, w, C1); C2 = fmaf(c.x, w, C2);
    T = ok ? T * (1.0f - alpha) : T;
    last = ok ? rel : last;
}

// One batch-by-batch walk of list entries [lo, hi) of a tile by one wave (its 8x8 block): gather, block test, ballot compaction
// into the wave's LDS queue, then `step(a, b, c, rel)` per hit in list order; `stop()` (wave-uniform) ends the walk early.
template <typename Step, typename Stop>
__device__ __forceinline__ void lg_walk_block(uint32_t list0, uint32_t lo, uint32_t hi, const uint64_t* __restrict__ entries, uint32_t gid_mask,
                                              const float4* __restrict__ rec, float bx0, float by0, float4* q0, float4* q1, float4* q2,
                                              uint32_t lane, Step step, Stop stop)
{
    for (uint32_t base = lo; base < hi; base += LG_Q) {
        if (stop()) break;
        const uint32_t idx = base + lane;
        bool hit = false;
        float4 r0, r1, r2;
        if (idx < hi) {
            const uint32_t id = (uint32_t)entries[list0 + idx] & gid_mask;
            r0 = rec[LG_REC_F4 * (size_t)id]; r1 = rec[LG_REC_F4 * (size_t)id + 1]; r2 = rec[LG_REC_F4 * (size_t)id + 2];
            hit = lg_block_hit(r0, r1, r2, lg_reach(r0, r1, r2), bx0, by0);
        }
        uint64_t mask = __ballot(hit);
        if (mask == 0) continue;
        if (hit) {
            const uint32_t pos = prefix_popc(mask);
            q0[pos] = r0; q1[pos] = r1; q2[pos] = r2;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t j = 0;
        while (mask) {
            const uint32_t src = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1;
            step(q0[j], q1[j], q2[j], base + src + 1u);       // rel = 1-based position in the tile's list
            j++;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// pass 2: per long tile, per pixel: prefix products over the segments.  A pixel whose transmittance never comes near the
// termination threshold is finished here; one that would stop inside segment s* is parked -- {T before s*, colour so far} in
// the checkpoint slot of s*, s* itself in the last-contributor word of slot 0 -- for lg_blend_fwd_rewalk.
// Round 3: not a launch of its own any more.  The workgroup of lg_blend_fwd_seg that finishes the LAST segment of a tile (one
// agent-scope arrival counter per tile, zeroed by the work-list workgroup of lg_blend_fwd) runs the tile's scan right away:
// nothing waits (no spinning, no co-residency assumption), one launch less per view, and on scenes with outlier lists the
// scans overlap the other tiles' segment walks.
#define LG_NO_SEG 0xFFFFFFFFu
__device__ __forceinline__ void lg_scan_tile(int W, int H, int gx, int S, int tile, const uint2 range, const float* __restrict__ bg,
                                             float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                             float4* __restrict__ ckpt, uint32_t* __restrict__ ckpt_last, int wave, uint32_t lane)
{
    const uint32_t n = range.y - range.x;
    const uint32_t nseg = (n + (uint32_t)S - 1u) / (uint32_t)S;
    const int tx = tile % gx, ty = tile / gx;
    const int pxi = tx * LG_TILE + (wave & 1) * 8 + (int)(lane & 7), pyi = ty * LG_TILE + (wave >> 1) * 8 + (int)(lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const uint32_t pix = ((uint32_t)(wave >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(wave & 1) * 8u + (lane & 7u);
    float4* ck = ckpt + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
    uint32_t* cl = ckpt_last + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0, sstar = LG_NO_SEG;
    if (inside) {
        for (uint32_t s = 0; s < nseg; s++) {
            const float4 r = ck[(size_t)s * 256];
            const uint32_t ll = cl[(size_t)s * 256];
            const float Tend = T * r.x;
            // would this pixel stop inside the segment?  (T P_s < 1e-4 up to the rounding of the regrouped product: the margin
            // only sends a few more pixels through the exact re-walk)
            if (!(Tend >= LG_T_MIN * 1.001f)) { sstar = s; break; }
            const float in0 = T * r.y, in1 = T * r.z, in2 = T * r.w;
            C0 += in0; C1 += in1; C2 += in2;
            T = Tend;
            last = ll ? ll : last;
            ck[(size_t)s * 256] = make_float4(T, in0, in1, in2);   // what the backward starts segment s from
        }
        if (sstar == LG_NO_SEG) {
            const size_t pid = (size_t)pyi * W + pxi, HW = (size_t)H * W;
            final_T[pid] = T;
            n_contrib[pid] = last;
            out_color[pid] = fmaf(T, bg[0], C0);
            out_color[HW + pid] = fmaf(T, bg[1], C1);
            out_color[2 * HW + pid] = fmaf(T, bg[2], C2);
        } else {
            ck[(size_t)sstar * 256] = make_float4(T, C0, C1, C2);
            if (sstar > 0u) cl[(size_t)sstar * 256] = last;       // (s* = 0: nothing contributed before it)
        }
    }
    cl[0] = sstar;                                                  // read by every (tile, segment) item of lg_blend_fwd_rewalk
}

#ifndef LG_PAR_GRID
#define LG_PAR_GRID 1024
#endif
//   // persistent workgroups of the three long-tile kernels (they loop over the par_work items)
__global__ void __launch_bounds__(256)
lg_blend_fwd_seg(int W, int H, int gx, int S, const uint2* __restrict__ par_work, const uint32_t* __restrict__ meta, const uint2* __restrict__ ranges,
                 const uint64_t* __restrict__ entries, uint32_t gid_mask, const float4* __restrict__ rec, float4* __restrict__ ckpt,
                 uint32_t* __restrict__ ckpt_last, uint32_t* par_arrived, const float* __restrict__ bg, float* __restrict__ out_color,
                 float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
{
    __shared__ float4 q0[4][LG_Q], q1[4][LG_Q], q2[4][LG_Q];
    __shared__ uint32_t s_last;
    const uint32_t nitems = meta[4];
    const int wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint2 item = par_work[it];
        const int tile = (int)item.x;
        const uint2 range = ranges[tile];
        const uint32_t n = range.y - range.x;
        const int tx = tile % gx, ty = tile / gx;
        const int wx0 = tx * LG_TILE + (wave & 1) * 8, wy0 = ty * LG_TILE + (wave >> 1) * 8;
        const int pxi = wx0 + (int)(lane & 7), pyi = wy0 + (int)(lane >> 3);
        const bool inside = pxi < W && pyi < H;
        const float pxf = (float)pxi, pyf = (float)pyi;
        const uint32_t lo = item.y * (uint32_t)S, hi = min(n, lo + (uint32_t)S);
        float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
        uint32_t last = 0;
        lg_walk_block(range.x, lo, hi, entries, gid_mask, rec, (float)wx0, (float)wy0, q0[wave], q1[wave], q2[wave], lane,
                      [&](const float4& a, const float4& b, const float4& c, uint32_t rel) { fwd_free(a, b, c, inside, pxf, pyf, T, C0, C1, C2, last, rel); },
                      [&]() { return false; });
        const uint32_t pix = ((uint32_t)(wave >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(wave & 1) * 8u + (lane & 7u);
        const size_t slot = ((size_t)2 * (range.x / (uint32_t)S) + item.y) * 256 + pix;
        ckpt[slot] = make_float4(T, C0, C1, C2);
        ckpt_last[slot] = last;
        // the last segment of this tile to arrive scans the tile (lg_scan_tile): release my records, count my arrival, and --
        // if every segment of the tile is in -- acquire the others' records
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t nseg = (n + (uint32_t)S - 1u) / (uint32_t)S;
            s_last = (__hip_atomic_fetch_add(&par_arrived[tile], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nseg - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            lg_scan_tile(W, H, gx, S, tile, range, bg, out_color, final_T, n_contrib, ckpt, ckpt_last, wave, lane);
        }
        __syncthreads();                                            // (s_last is rewritten by the next item)
    }
}

// pass 3: one workgroup per (long tile, segment) again; a wave has work only if one of its pixels was parked at this segment.
// Those pixels walk the segment SEQUENTIALLY from their true T with the published pair step -- exact stop position, exact
// contributor index -- and are finished here.  (A parked pixel that turns out not to stop inside its segment -- the margin of
// the scan -- simply keeps walking the following segments the same way: rare, and exact.)
__global__ void __launch_bounds__(256)
lg_blend_fwd_rewalk(int W, int H, int gx, int S, const uint2* __restrict__ par_work, const uint32_t* __restrict__ meta, const uint2* __restrict__ ranges,
                    const uint64_t* __restrict__ entries, uint32_t gid_mask, const float4* __restrict__ rec, const float* __restrict__ bg,
                    float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float4* __restrict__ ckpt,
                    const uint32_t* __restrict__ ckpt_last)
{
    __shared__ float4 q0[4][LG_Q], q1[4][LG_Q], q2[4][LG_Q];
    const uint32_t nitems = meta[4];
    const int wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint2 item = par_work[it];
        const int tile = (int)item.x;
        const uint2 range = ranges[tile];
        const uint32_t n = range.y - range.x;
        const uint32_t nseg = (n + (uint32_t)S - 1u) / (uint32_t)S;
        const int tx = tile % gx, ty = tile / gx;
        const int wx0 = tx * LG_TILE + (wave & 1) * 8, wy0 = ty * LG_TILE + (wave >> 1) * 8;
        const int pxi = wx0 + (int)(lane & 7), pyi = wy0 + (int)(lane >> 3);
        const bool inside = pxi < W && pyi < H;
        const float pxf = (float)pxi, pyf = (float)pyi;
        const uint32_t pix = ((uint32_t)(wave >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(wave & 1) * 8u + (lane & 7u);
        float4* ck = ckpt + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
        const uint32_t* cl = ckpt_last + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
        const bool mine = inside && cl[0] == item.y;                    // parked at this segment
        if (__ballot(mine) == 0) continue;
        float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
        uint32_t last = 0;
        if (mine) {
            const float4 st = ck[(size_t)item.y * 256];
            T = st.x; C0 = st.y; C1 = st.z; C2 = st.w;
            last = item.y > 0u ? cl[(size_t)item.y * 256] : 0u;
        }
        uint64_t dnm = ~__builtin_amdgcn_ballot_w64(mine);              // finished (or not parked here): one scalar mask per wave (fwd_pair_m)
        uint32_t cur = item.y, mynext = item.y;                         // mynext: first segment this pixel did not enter
        for (; cur < nseg && ~dnm != 0ull; cur++) {                     // wave-uniform
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
            uint32_t ls = 0;
            const bool entered = !__builtin_amdgcn_inverse_ballot_w64(dnm);
            const uint32_t lo = cur * (uint32_t)S, hi = min(n, lo + (uint32_t)S);
            lg_walk_block(range.x, lo, hi, entries, gid_mask, rec, (float)wx0, (float)wy0, q0[wave], q1[wave], q2[wave], lane,
                          [&](const float4& a, const float4& b, const float4& c, uint32_t rel) {
                              float alpha = 0.0f, w = 0.0f;
                              (void)fwd_pair_m<false, true>(a, b, c, dnm, pxf, pyf, T, s0, s1, s2, ls, rel, alpha, w);
                          },
                          [&]() { return ~dnm == 0ull; });
            if (entered) {
                C0 += s0; C1 += s1; C2 += s2;
                last = ls ? ls : last;
                ck[(size_t)cur * 256] = make_float4(T, s0, s1, s2);
                mynext = cur + 1u;
            }
        }
        if (mine) {
            for (uint32_t j = mynext; j < nseg; j++) ck[(size_t)j * 256] = make_float4(T, 0.0f, 0.0f, 0.0f);   // segments never entered
            const size_t pid = (size_t)pyi * W + pxi, HW = (size_t)H * W;
            final_T[pid] = T;
            n_contrib[pid] = last;
            out_color[pid] = fmaf(T, bg[0], C0);
            out_color[HW + pid] = fmaf(T, bg[1], C1);
            out_color[2 * HW + pid] = fmaf(T, bg[2], C2);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Long tiles of the SIGNIFICANCE-ONLY pass (count forward, integer weights, LG_FLAG_SKIP_COLOR), segments in parallel -- round 5,
// r4 verdict item 5.  The pass is north_star's signature kernel and walked every list serially: on the heavy-tailed scene one
// workgroup owns a 24 000-entry list for ~0.6 ms while the device idles (blend_fwd_count 1.0 ms against 0.42 on the uniform scene).
// What the pass must reproduce bit for bit are the hit COUNTS: sums of include / exclude decisions -- power <= 0 and alpha >= 1/255
// (independent of everything before the entry) and "the pixel has not stopped yet", i.e. the exact position at which the
// SEQUENTIAL float product T (1 - alpha_1)(1 - alpha_2)... first falls under 1e-4.  A regrouped product (segment products multiplied
// together) differs from the sequential one by rounding, so a stop position derived from it would be off by one entry for a few
// pixels per view (P ~ 2 delta / log-step ~ 1e-5 per saturating pixel).  Hence intervals instead of values: with K factors so far both
// products lie within K u (u = 2^-24) of the exact real product; every comparison of the regrouped T against the threshold is made
// with the band LG_CNT_BAND(K) >= 2.5 K u around it, and only a comparison that falls INSIDE the band is undecided:
//   pass 1  lg_count_seg: one workgroup per (long tile, segment): per pixel P_s = prod (1 - alpha) over the segment's contributing
//           entries (sequential inside the segment, from 1.0f) and their number k_s.  No counting yet;
//   pass 2  lg_count_scan_tile (run by the last segment of a tile to arrive): per pixel, segment by segment, T <- T P_s, K <- K + k_s
//           while T is CERTAINLY >= 1e-4 (beyond the band): the pixel is alive through those segments; the first segment s* for
//           which that cannot be said (the pixel stops there, or may) parks it with {T, K} at the start of s*;
//   pass 3  lg_count_rewalk: one workgroup per (long tile, segment) again.  Pixels certainly alive through this segment count every
//           included entry -- no T needed: ballot + popcount per entry, one atomic per (wave, entry), as lg_blend_fwd<COUNT>.  Pixels
//           parked here walk it sequentially from their regrouped T: a step whose test value T (1 - alpha) lies outside the band is
//           decided (contributes / stops) exactly as the sequential walk decides it; the first step INSIDE the band freezes the pixel
//           and records the entry's list position (flag).  A parked pixel that survives its segment walks on into the next ones;
//   pass 4  lg_count_fixup: the frozen pixels -- a handful per view -- are resolved exactly, one wave per pixel with the ENTRIES on
//           the lanes: alpha of every entry for that one pixel in parallel, then the sequential product over the contributing
//           entries in list order (the very operation sequence of the serial walk) up to the stop; entries from the flagged
//           position on are counted.
// Counts are therefore bit-identical to the serial walk and to the oracle whatever the scheduling (integers: order-free; every
// decision either provably equal or recomputed sequentially).  Images are not produced (LG_FLAG_SKIP_COLOR); count forwards that
// return an image, and the float weight policies, keep the serial walk.  Which lists: lg_par_min(), as for the colour forward.
#define LG_CNT_BAND(k) ((float)((k) + 16u) * 1.5e-7f * band_mul)      // >= 2.5 u per factor, u = 2^-24: twice the rounding of a K-factor float product, with margin
// (band_mul: 1 in production; LG_FLAG_COUNT_WIDE_BAND -- tests -- widens the band 4096 x so that a fifth of the saturating pixels go
//  through the exact fix-up instead of a handful per view: the counts must not change)

// canonical alpha of one (entry, pixel) pair + the T-independent half of its include decision
__device__ __forceinline__ uint64_t lg_count_alpha(const float4& a, const float4& b, float pxf, float pyf, float& alpha)
{
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy);
    alpha = fminf(LG_ALPHA_MAX, b.y * lg_exp(fminf(power, 0.0f)));
    return __builtin_amdgcn_ballot_w64(power <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= LG_ALPHA_MIN);
}

// one wave, its 8 x 8 block, list entries [lo, hi) of a tile: gather, block test, ballot compaction into the wave's LDS queue, then
// step(a, b, rel) -> mask of the lanes (pixels) the entry is counted for; with `count` != NULL lane j adds the popcount of entry j
template <typename Step, typename Stop>
__device__ __forceinline__ void lg_count_walk(uint32_t list0, uint32_t lo, uint32_t hi, const uint64_t* __restrict__ entries, uint32_t gid_mask,
                                              const float4* __restrict__ rec, float bx0, float by0, float4* q0, float4* q1, float4* q2, uint32_t lane,
                                              int32_t* __restrict__ count, Step step, Stop stop)
{
    for (uint32_t base = lo; base < hi; base += LG_Q) {
        if (stop()) break;
        const uint32_t idx = base + lane;
        bool hit = false;
        float4 r0, r1, r2;
        if (idx < hi) {
            const uint32_t id = (uint32_t)entries[list0 + idx] & gid_mask;
            r0 = rec[LG_REC_F4 * (size_t)id]; r1 = rec[LG_REC_F4 * (size_t)id + 1]; r2 = rec[LG_REC_F4 * (size_t)id + 2];
            hit = lg_block_hit(r0, r1, r2, lg_reach(r0, r1, r2), bx0, by0);
        }
        const uint64_t mask = __ballot(hit);
        if (mask == 0) continue;
        if (hit) {
            const uint32_t pos = prefix_popc(mask);
            r2.y = __uint_as_float(idx + 1u);                      // 1-based position in the tile's list (the box half-extent is not needed any more)
            q0[pos] = r0; q1[pos] = r1; q2[pos] = r2;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t nhit = (uint32_t)__popcll(mask);
        int mycnt = 0;
        for (uint32_t j = 0; j < nhit; j++) {
            const float4 a = q0[j], b = q1[j], c = q2[j];
            const uint64_t cm = step(a, b, __float_as_uint(c.y));
            if (lane == j) mycnt = (int)__popcll(cm);
        }
        if (count != nullptr && lane < nhit && mycnt > 0) atomicAdd(&count[__float_as_uint(q2[lane].w) & LG_ID_MASK], mycnt);
        __builtin_amdgcn_wave_barrier();
    }
}

// pass 2 (see above); slot layout of a long tile as in the colour forward: record j at ckpt[(2 (range.x / S) + j) * 256 + pixel]
__device__ __forceinline__ void lg_count_scan_tile(int W, int H, int gx, int S, int tile, const uint2 range, float4* __restrict__ ckpt,
                                                   uint32_t* __restrict__ ckpt_last, int wave, uint32_t lane, float band_mul)
{
    const uint32_t n = range.y - range.x;
    const uint32_t nseg = (n + (uint32_t)S - 1u) / (uint32_t)S;
    const int tx = tile % gx, ty = tile / gx;
    const int pxi = tx * LG_TILE + (wave & 1) * 8 + (int)(lane & 7), pyi = ty * LG_TILE + (wave >> 1) * 8 + (int)(lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const uint32_t pix = ((uint32_t)(wave >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(wave & 1) * 8u + (lane & 7u);
    float4* ck = ckpt + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
    uint32_t* cl = ckpt_last + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
    float T = 1.0f;
    uint32_t K = 0, sstar = LG_NO_SEG;
    const float P0 = ck[0].x;
    if (inside) {
        for (uint32_t s = 0; s < nseg; s++) {
            const float P = s == 0u ? P0 : ck[(size_t)s * 256].x;
            const uint32_t k = cl[(size_t)s * 256];
            const float Tend = T * P;
            // certainly alive through segment s?  (the regrouped and the sequential product of K + k factors agree to the band)
            if (!(Tend >= LG_T_MIN * (1.0f + LG_CNT_BAND(K + k + s)))) { sstar = s; break; }
            T = Tend; K += k;
        }
        if (sstar != LG_NO_SEG) { float4 r = ck[(size_t)sstar * 256]; r.z = T; r.w = __uint_as_float(K); ck[(size_t)sstar * 256] = r; }
    }
    // word y of record 0: where this pixel is parked (read by every item of lg_count_rewalk); last-word of record 0: its flag (0 = none)
    { float4 r = ck[0]; r.x = P0; r.y = __uint_as_float(sstar); ck[0] = r; }
    cl[0] = 0u;
}

// The exact resolution of ONE pixel (passes 4 of the parallel walk and of the banded walk below): the pixel's walk from the start of the
// tile's list with the ENTRIES on the lanes -- canonical alpha of 64 entries at a time, then the sequential product over the contributing
// ones in list order (T is wave-uniform), the very operation sequence of the serial canonical walk -- up to the stop; entries from list
// position jstar on are counted.  No block culling: it only ever drops entries that contribute to no pixel of the block.
__device__ __forceinline__ void lg_count_exact_pixel(const uint2 range, const uint64_t* __restrict__ entries, uint32_t gid_mask,
                                                     const float4* __restrict__ rec, float pxf, float pyf, uint32_t jstar, int32_t* __restrict__ count,
                                                     uint32_t lane)
{
    const uint32_t n = range.y - range.x;
    float T = 1.0f;
    bool stopped = false;
    for (uint32_t base = 0; base < n && !stopped; base += LG_Q) {
        const uint32_t e = base + lane;
        float alpha = 0.0f;
        bool ok = false;
        uint32_t id = 0;
        if (e < n) {
            id = (uint32_t)entries[range.x + e] & gid_mask;
            const float4 a = rec[LG_REC_F4 * (size_t)id], b = rec[LG_REC_F4 * (size_t)id + 1];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy);
            alpha = fminf(LG_ALPHA_MAX, b.y * lg_exp(fminf(power, 0.0f)));
            ok = (power <= 0.0f) && (alpha >= LG_ALPHA_MIN);
        }
        uint64_t cntm = 0ull;
        for (uint64_t okm = __builtin_amdgcn_ballot_w64(ok); okm != 0ull; okm &= okm - 1ull) {
            const int bit = (int)__builtin_ctzll(okm);
            const float al = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha), bit));
            const float test_T = T * (1.0f - al);
            if (test_T < LG_T_MIN) { stopped = true; break; }
            T = test_T;
            if (base + (uint32_t)bit + 1u >= jstar) cntm |= 1ull << bit;
        }
        if ((cntm >> lane) & 1ull) atomicAdd(&count[id], 1);
    }
}

__global__ void __launch_bounds__(256)
lg_count_seg(int W, int H, int gx, int S, const uint2* __restrict__ par_work, const uint32_t* __restrict__ meta, const uint2* __restrict__ ranges,
             const uint64_t* __restrict__ entries, uint32_t gid_mask, const float4* __restrict__ rec, float4* __restrict__ ckpt,
             uint32_t* __restrict__ ckpt_last, uint32_t* par_arrived, float band_mul)
{
    __shared__ float4 q0[4][LG_Q], q1[4][LG_Q], q2[4][LG_Q];
    __shared__ uint32_t s_last;
    const uint32_t nitems = meta[4];
    const int wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint2 item = par_work[it];
        const int tile = (int)item.x;
        const uint2 range = ranges[tile];
        const uint32_t n = range.y - range.x;
        const int tx = tile % gx, ty = tile / gx;
        const int wx0 = tx * LG_TILE + (wave & 1) * 8, wy0 = ty * LG_TILE + (wave >> 1) * 8;
        const float pxf = (float)(wx0 + (int)(lane & 7)), pyf = (float)(wy0 + (int)(lane >> 3));
        const uint32_t lo = item.y * (uint32_t)S, hi = min(n, lo + (uint32_t)S);
        float P = 1.0f;
        uint32_t k = 0;
        lg_count_walk(range.x, lo, hi, entries, gid_mask, rec, (float)wx0, (float)wy0, q0[wave], q1[wave], q2[wave], lane, nullptr,
                      [&](const float4& a, const float4& b, uint32_t) -> uint64_t {
                          float alpha;
                          const bool ok = __builtin_amdgcn_inverse_ballot_w64(lg_count_alpha(a, b, pxf, pyf, alpha));
                          const float Pn = P * (1.0f - alpha);
                          P = ok ? Pn : P;
                          k += ok ? 1u : 0u;
                          return 0ull;
                      },
                      [&]() { return false; });
        const uint32_t pix = ((uint32_t)(wave >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(wave & 1) * 8u + (lane & 7u);
        const size_t slot = ((size_t)2 * (range.x / (uint32_t)S) + item.y) * 256 + pix;
        ckpt[slot] = make_float4(P, 0.0f, 0.0f, 0.0f);
        ckpt_last[slot] = k;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t nseg = (n + (uint32_t)S - 1u) / (uint32_t)S;
            s_last = (__hip_atomic_fetch_add(&par_arrived[tile], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nseg - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            lg_count_scan_tile(W, H, gx, S, tile, range, ckpt, ckpt_last, wave, lane, band_mul);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
lg_count_rewalk(int W, int H, int gx, int S, const uint2* __restrict__ par_work, const uint32_t* __restrict__ meta, const uint2* __restrict__ ranges,
                const uint64_t* __restrict__ entries, uint32_t gid_mask, const float4* __restrict__ rec, const float4* __restrict__ ckpt,
                uint32_t* __restrict__ ckpt_last, int32_t* __restrict__ count, float band_mul)
{
    __shared__ float4 q0[4][LG_Q], q1[4][LG_Q], q2[4][LG_Q];
    const uint32_t nitems = meta[4];
    const int wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint2 item = par_work[it];
        const int tile = (int)item.x;
        const uint2 range = ranges[tile];
        const uint32_t n = range.y - range.x;
        const uint32_t nseg = (n + (uint32_t)S - 1u) / (uint32_t)S;
        const int tx = tile % gx, ty = tile / gx;
        const int wx0 = tx * LG_TILE + (wave & 1) * 8, wy0 = ty * LG_TILE + (wave >> 1) * 8;
        const int pxi = wx0 + (int)(lane & 7), pyi = wy0 + (int)(lane >> 3);
        const bool inside = pxi < W && pyi < H;
        const float pxf = (float)pxi, pyf = (float)pyi;
        const uint32_t pix = ((uint32_t)(wave >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(wave & 1) * 8u + (lane & 7u);
        const float4* ck = ckpt + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
        uint32_t* cl = ckpt_last + (size_t)2 * (range.x / (uint32_t)S) * 256 + pix;
        const uint32_t sstar = __float_as_uint(ck[0].y);
        const bool mine = inside && sstar == item.y;                                     // parked at this segment
        uint64_t freem = __builtin_amdgcn_ballot_w64(inside && (sstar == LG_NO_SEG || item.y < sstar));   // certainly alive through it
        uint64_t alivem = __builtin_amdgcn_ballot_w64(mine);
        if ((freem | alivem) == 0ull) continue;
        float T = 1.0f;
        uint32_t K = 0, flag = 0;
        if (mine) { const float4 st = ck[(size_t)item.y * 256]; T = st.z; K = __float_as_uint(st.w); }
        for (uint32_t cur = item.y; cur < nseg && (freem | alivem) != 0ull; cur++) {    // wave-uniform
            const uint32_t lo = cur * (uint32_t)S, hi = min(n, lo + (uint32_t)S);
            lg_count_walk(range.x, lo, hi, entries, gid_mask, rec, (float)wx0, (float)wy0, q0[wave], q1[wave], q2[wave], lane, count,
                          [&](const float4& a, const float4& b, uint32_t rel) -> uint64_t {
                              float alpha;
                              const uint64_t okm = lg_count_alpha(a, b, pxf, pyf, alpha);
                              const uint64_t am = okm & alivem;
                              uint64_t con = 0ull;
                              if (am != 0ull) {                                          // (wave-uniform: only while a parked pixel is walking)
                                  const float test_T = T * (1.0f - alpha);
                                  const uint64_t unc = am & __builtin_amdgcn_ballot_w64(fabsf(test_T - LG_T_MIN) <= LG_T_MIN * LG_CNT_BAND(K));
                                  const uint64_t sat = am & ~unc & __builtin_amdgcn_ballot_w64(test_T < LG_T_MIN);
                                  con = am & ~(unc | sat);
                                  const bool isunc = __builtin_amdgcn_inverse_ballot_w64(unc), iscon = __builtin_amdgcn_inverse_ballot_w64(con);
                                  flag = isunc ? rel : flag;
                                  T = iscon ? test_T : T;
                                  K += iscon ? 1u : 0u;
                                  alivem &= ~(unc | sat);
                              }
                              return (okm & freem) | con;
                          },
                          [&]() { return (freem | alivem) == 0ull; });
            freem = 0ull;                       // the following segments are walked for the surviving parked pixels only
        }
        if (flag != 0u) cl[0] = flag;           // frozen at list position `flag`: lg_count_fixup resolves it exactly
    }
}

__global__ void __launch_bounds__(256)
lg_count_fixup(int W, int H, int gx, int S, const uint2* __restrict__ par_work, uint32_t* __restrict__ meta, const uint2* __restrict__ ranges,
               const uint64_t* __restrict__ entries, uint32_t gid_mask, const float4* __restrict__ rec, const uint32_t* __restrict__ ckpt_last,
               int32_t* __restrict__ count)
{
    const uint32_t nitems = meta[4];
    const int wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint2 item = par_work[it];
        if (item.y != 0u) continue;                                                      // one item per long tile
        const int tile = (int)item.x;
        const uint2 range = ranges[tile];
        const uint32_t n = range.y - range.x;
        const int tx = tile % gx, ty = tile / gx;
        const int pxi = tx * LG_TILE + (wave & 1) * 8 + (int)(lane & 7), pyi = ty * LG_TILE + (wave >> 1) * 8 + (int)(lane >> 3);
        const uint32_t pix = ((uint32_t)(wave >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(wave & 1) * 8u + (lane & 7u);
        const uint32_t myflag = (pxi < W && pyi < H) ? ckpt_last[(size_t)2 * (range.x / (uint32_t)S) * 256 + pix] : 0u;
        for (uint64_t todo = __builtin_amdgcn_ballot_w64(myflag != 0u); todo != 0ull; todo &= todo - 1ull) {
            const int src = (int)__builtin_ctzll(todo);
            const uint32_t jstar = (uint32_t)__builtin_amdgcn_readlane((int)myflag, src);
            if (lane == 0u) atomicAdd(&meta[5], 1u);
            const float pxf = (float)__builtin_amdgcn_readlane(pxi, src), pyf = (float)__builtin_amdgcn_readlane(pyi, src);
            lg_count_exact_pixel(range, entries, gid_mask, rec, pxf, pyf, jstar, count, lane);
        }
    }
}

// (hardware-exp count walk with an error band: EXPERIMENTS.md, "significance pass")
// per-view score from the exact integer count (ONE / OPACITY weights): score = lg_seqsum32(weight, count), the float that `count`
// sequential additions of the weight leave (lg_math.h).
// (workgroup compaction by count magnitude, persistent workgroups: EXPERIMENTS.md, "significance pass")
__global__ void __launch_bounds__(256)
lg_score_kernel(int N, const int32_t* __restrict__ count, const float* __restrict__ weight, float* __restrict__ score, int32_t* __restrict__ count_sum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int c = count[i];
    score[i] = c > 0 ? lg_seqsum32(weight ? weight[i] : 1.0f, (uint32_t)c) : 0.0f;
    if (count_sum && c) count_sum[i] += c;       // lg_view.count_sum: the caller's running hit count (one view at a time per accumulator)
}

// per-view count and score of the ALPHA / ALPHA_T policies: every Gaussian sums the {count : 16 | Q8.40 : 48} words of its own instances
// (consecutive pre-sort slots, offsets - touched .. offsets) -- integer adds, any order -- and rounds the Q24.40 total to fp32 ONCE (nearest even),
// times 2^-40 (exact).  A lane walks up to LG_SLOT_SOLO slots itself; Gaussians with more (screen-filling splats) are summed by the whole wave.
#define LG_SLOT_SOLO 16u
__global__ void __launch_bounds__(256)
lg_score_slots(int N, const uint32_t* __restrict__ touched, const uint32_t* __restrict__ offsets, const unsigned long long* __restrict__ slots,
               uint32_t slot_cap, int32_t* __restrict__ count, float* __restrict__ score, int32_t* __restrict__ count_sum,
               const uint32_t* __restrict__ counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n = 0, base = 0;
    if (i < N) { n = touched[i]; base = offsets[i] - n; }          // offsets = inclusive scan of touched (lg_duplicate): the slot base without the 16-byte record
    // a view that outgrew its capacity (or whose sort gave up) is void: its slots were never written -- it contributes zeros
    if (counters[0] != 0u || base >= slot_cap || n > slot_cap - base) n = 0;
    uint64_t fix = 0; uint32_t cnt = 0;
    if (n <= LG_SLOT_SOLO)
        for (uint32_t k = 0; k < n; k++) { const uint64_t w = slots[base + k]; cnt += (uint32_t)(w >> LG_SLOT_COUNT_SHIFT); fix += w & ((1ull << LG_SLOT_COUNT_SHIFT) - 1ull); }
    uint64_t big = __ballot(n > LG_SLOT_SOLO);
    while (big) {                                                 // wave-uniform
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t bn = (uint32_t)__builtin_amdgcn_readlane((int)n, src), bb = (uint32_t)__builtin_amdgcn_readlane((int)base, src);
        uint64_t f = 0; uint32_t c = 0;
        for (uint32_t k = lane; k < bn; k += 64u) { const uint64_t w = slots[bb + k]; c += (uint32_t)(w >> LG_SLOT_COUNT_SHIFT); f += w & ((1ull << LG_SLOT_COUNT_SHIFT) - 1ull); }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) { c += __shfl_xor(c, sft, 64); f += __shfl_xor(f, sft, 64); }
        if ((int)lane == src) { cnt = c; fix = f; }
    }
    if (i < N) {
        count[i] = (int32_t)cnt; score[i] = lg_fix40_score(fix);
        if (count_sum && cnt) count_sum[i] += (int32_t)cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// K7: backward blend.  One wave per 16x16 tile, FOUR pixels per lane (the four 8x8 sub-blocks), so a
// (tile, Gaussian) instance is reduced across lanes once, not once per 8x8 block.  Per batch of 64 list
// entries: lane l gathers entry l and computes its 4-bit sub-block overlap mask; the wave then walks the
// batch back to front, evaluating an entry only on the sub-blocks it overlaps (scalar branches on the
// mask).  The 9 partials are reduced through an LDS transpose (wave_reduce9_via_lds), parked in LDS, and written once per batch as 48-byte gradient
// rows (lane j owns entry j): part [R][12] floats (9 used) = the five pixel-offset moments | sum t | drgb (see
// lg_rows_to_grads), addressed by the instance's pre-sort slot -- no atomics, every row written exactly once.
// K7's wave reduction of the 9 partial sums goes through LDS (the register-only permlane / DPP fold of round 1 compiled to 36 VALU
// instructions + hazard nops per entry).  The 9 partials of every lane go to a 9 x 64 matrix `red` (row stride 68 floats:
// conflict-free for the dwordx4 reads below); lane 4 r + q then adds columns 16 q .. 16 q + 15 of row r (four ds_read_b128,
// 15 adds) and two quad-DPP adds join the four quarters: 17 VALU instructions per entry instead of the 36 (+ hazard nops) of
// the register-only fold above -- the fold was a quarter of K7's instruction stream.  The LDS operations of one wave execute
// in order, so the reads see every lane's writes without a barrier; lanes >= 36 redo row 8 (clamped row: no exec masking).
#define LG_RED_STRIDE 68
#define LG_RED_FLOATS (9 * LG_RED_STRIDE)
__device__ __forceinline__ void wave_reduce9_via_lds(const float (&p)[9], float* red, float* dst, uint32_t dst_off, uint32_t lane)
{
#pragma unroll
    for (int v = 0; v < 9; v++) red[v * LG_RED_STRIDE + (int)lane] = p[v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t r = min(lane >> 2, 8u), q = lane & 3u;
    const float4* src = reinterpret_cast<const float4*>(red + r * LG_RED_STRIDE + q * 16u);
    const float4 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
    float s = (((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w))) +
              (((x2.x + x2.y) + (x2.z + x2.w)) + ((x3.x + x3.y) + (x3.z + x3.w)));
    s = dpp_add<0xB1, 0xf>(s);                      // quad_perm [1,0,3,2]
    s = dpp_add<0x4E, 0xf>(s);                      // quad_perm [2,3,0,1]: every lane of the quad holds the row total
    // dst is VALUE-major, [9][LG_Q]: total v of entry j at dst[v * LG_Q + j].  (Entry-major, `stage + 9 j` + v, cost a 64-bit
    // v_mad_u64_u32 per entry for j * 36 + base; here the address is a per-lane constant plus a scalar shift of j.)
    if (q == 0u && lane < 36u) dst[(lane >> 2) * LG_Q + dst_off] = s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                // (the next entry overwrites `red`)
}

// (two entries per reduction pass, quads summed in registers first, packed adds: EXPERIMENTS.md, "K7 reduction")

// Gradient rows hold MOMENTS, not finished gradients.  With t = G * dL/dalpha per (pixel, Gaussian) pair, dx = x_g - px:
//   p[0] = sum t dx   p[1] = sum t dy   p[2] = sum t dx^2   p[3] = sum t dx dy   p[4] = sum t dy^2   p[5] = sum t
//   p[6..8] = sum alpha T dL/dC_c
// The published per-pair expressions are linear in these sums with per-Gaussian coefficients (conic, opacity):
//   dL/dmean2D.x = op (2 ha p0 + nb p1)   dL/dmean2D.y = op (2 hc p1 + nb p0)     (ha = -A/2, nb = -B, hc = -C/2)
//   dL/dA = -op p2 / 2   dL/dB = -op p3   dL/dC = -op p4 / 2   dL/dopacity = p5
// so K9 applies them ONCE per Gaussian after summing its rows (lg_rows_to_grads) instead of K7 once per pair: 9 VALU
// instructions per pair evaluation instead of 19 for this part, same values up to float rounding.
// (lg_rows_to_grads lives in lg_math.h)

// One (pixel, Gaussian) step of the back-to-front replay.  EXACT = canonical arithmetic for every include / exclude
// decision (same sequence as the oracle); otherwise hardware exp / rcp and contraction allowed (training path, 1e-4 contract).
template <bool EXACT>
__device__ __forceinline__ bool bwd_pair(const float4& a, const float4& b, const float4& c, float pxf, float pyf, float& T, float Tfb,
                                         float g0, float g1, float g2, float& a0, float& a1, float& a2, float& last_alpha,
                                         float& lc0, float& lc1, float& lc2, float (&p)[9])
{
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy);
    if (power > 0.0f) return false;
    const float op = b.y;
    if (EXACT) {
        const float G = lg_exp(power);
        const float alpha = fminf(LG_ALPHA_MAX, op * G);
        if (alpha < LG_ALPHA_MIN) return false;
        const float om = 1.0f - alpha;
        T = T / om;
        const float dch = alpha * T;
        const float c0 = b.z, c1 = b.w, c2 = c.x;
        a0 = last_alpha * lc0 + (1.0f - last_alpha) * a0;
        a1 = last_alpha * lc1 + (1.0f - last_alpha) * a1;
        a2 = last_alpha * lc2 + (1.0f - last_alpha) * a2;
        lc0 = c0; lc1 = c1; lc2 = c2;
        float dL_dalpha = (c0 - a0) * g0 + (c1 - a1) * g1 + (c2 - a2) * g2;
        dL_dalpha = dL_dalpha * T;
        last_alpha = alpha;
        dL_dalpha = dL_dalpha + (-Tfb / om);
        const float t = G * dL_dalpha;
        const float tdx = t * dx, tdy = t * dy;
        p[0] += tdx;
        p[1] += tdy;
        p[2] += tdx * dx;
        p[3] += tdx * dy;
        p[4] += tdy * dy;
        p[5] += t;
        p[6] += dch * g0; p[7] += dch * g1; p[8] += dch * g2;
        return true;
    }
    return false;
}

// Training-path variant (hardware exp / rcp, contraction allowed), written BRANCH-FREE: every lane runs
// the whole sequence and invalid lanes are neutralised by zeroing t and the colour weight and by
// selecting the old state.  (A branchy version makes hipcc copy the 9 accumulators at every nesting level.  A scalar
// early-out when no lane of the 8x8 block takes the entry -- `if (__ballot(ok) == 0) return` -- was measured: 0.946 vs
// 0.927 ms; blocks that pass the box test almost always have a contributing pixel.)
__device__ __forceinline__ uint64_t bwd_pair_fast(const float4& a, const float4& b, const float4& c, bool live, float pxf, float pyf,
                                                  float& T, float Tfb, float g0, float g1, float g2, float& S, float (&p)[9])
{
#pragma clang fp contract(fast)
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy); // identical to the forward's expression
    const float G = __expf(power);                       // (power > 0: rejected by `ok`; see fwd_pair)
    const float op = b.y;
    const float alpha = guard_alpha(fminf(LG_ALPHA_MAX, op * G), op, power); // same decisions as the forward
    // the three compares as scalar lane masks, the valid set back to a predicate through inverse.ballot (round 5, as fwd_pair_m): the mask is
    // also what the caller wants back -- rounds 1-4 returned ballot(am > 0), a fourth compare per pair step for the same set
    const uint64_t okm = __builtin_amdgcn_ballot_w64(live) & __builtin_amdgcn_ballot_w64(power <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= LG_ALPHA_MIN);
    const bool ok = __builtin_amdgcn_inverse_ballot_w64(okm);
    // am = alpha on valid lanes, 0 elsewhere: with am = 0 the colour recurrence below is the identity (S + 0 * d = S)
    // and dch = 0 (v_cndmask / v_cmp / v_min cost ~1.7x an fma on gfx950, tools/ubench/valu_rate2.hip).
    const float am = ok ? alpha : 0.0f;
    const float inv = __builtin_amdgcn_rcpf(1.0f - am);
    const float Tn = T * inv;
    // S = (colour accumulated behind this entry) . dL/dC of this pixel, carried as ONE scalar (round 3): the published
    // recurrence a <- a + alpha (c - a) is linear, so its projection on g obeys S <- S + alpha (c.g - S), and dL/dalpha only
    // ever needs (c - a) . g = c.g - S.  5 instructions (c.g: mul + 2 fma; d; S update) where the vector form took 9 (three
    // differences, their dot product, three updates), and one state register per pixel instead of three.
    const float d = (b.z * g0 + b.w * g1 + c.x * g2) - S;
    const float dL_dalpha = d * Tn - Tfb * inv;          // Tfb = T_final * (bg . dL/dC), per pixel
    const float t = ok ? G * dL_dalpha : 0.0f;
    // no select on T: on invalid lanes am = 0, and v_rcp_f32(1.0f) is exactly 1.0f on gfx950, so Tn == T bit for bit there
    // (measured: gradients bit-identical to the version with `T = ok ? Tn : T`, K7 1.2 % faster)
    T = Tn;
    S = am * d + S;
    const float dch = am * Tn;
    const float tdx = t * dx, tdy = t * dy;
    // (a `fresh` flag under which the first sub-block assigns the nine sums: EXPERIMENTS.md, "K7 structure")
    p[0] += tdx;
    p[1] += tdy;
    p[2] += tdx * dx;
    p[3] += tdx * dy;
    p[4] += tdy * dy;
    p[5] += t;
    p[6] += dch * g0; p[7] += dch * g1; p[8] += dch * g2;
    // the lanes that contributed: the scalar mask itself (rounds 1-3 balloted the bool `ok`, which hipcc materialises as v_cndmask(0, 1) +
    // v_cmp_ne; round 4 balloted am > 0, one compare; now none)
    return okm;
}

#ifndef LG_K7_WAVES
#define LG_K7_WAVES 5      // waves per SIMD the register allocation of lg_blend_bwd aims for (84 VGPRs as written: 5)
#endif
template <bool EXACT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LG_K7_WAVES, 8)))
lg_blend_bwd(int W, int H, int gx, int S, const uint2* __restrict__ work, const uint32_t* __restrict__ meta, const uint2* __restrict__ ranges,
             const uint64_t* __restrict__ entries, uint32_t gid_mask, const uint4* __restrict__ tinfo, const float4* __restrict__ rec,
             const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
             const float* __restrict__ dL_dpix, const float4* __restrict__ ckpt, float* __restrict__ part)
{
    __shared__ float4 q0[LG_Q], q1[LG_Q];
    __shared__ float stage[LG_Q * 9];
    // the entry's third colour channel lives in row 8 of `stage` until the entry's own totals overwrite it (its reduction runs after
    // its last read): 256 bytes of LDS less per wave, 6800 in all (round 4: K7 0.692 -> 0.682 ms bracketed.  Six waves per SIMD on top
    // of it -- 80 VGPRs, 24 workgroups per CU now fit -- change nothing further, 0.683: measured again, as in round 3)
    float* const q2 = stage + 8 * LG_Q;
    __shared__ __attribute__((aligned(16))) float red[LG_RED_FLOATS];
    if (blockIdx.x >= meta[0]) return;            // the grid is sized for the worst case: tiles + R / S work items
    if (meta[2] != (uint32_t)S) return;           // another segment length than the forward's (see lg_preprocess_bwd)
    const uint2 item = work[blockIdx.x];          // {tile, segment}, longest first (lg_work_order)
    const int tile = (int)item.x;
    const uint32_t lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)H * W;
    const float bgr = bg[0], bgg = bg[1], bgb = bg[2];

    float pxf[4], pyf[4], T[4], Tfb[4], g0[4], g1[4], g2[4], a0[4], a1[4], a2[4], la[4], lc0[4], lc1[4], lc2[4];
    float Sd[4];            // hardware-exp variant: (colour behind) . dL/dC per pixel, instead of a0..a2 (bwd_pair_fast)
    uint32_t last[4];
    bool inside4[4];
    uint32_t wmax = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int pxi = tx * LG_TILE + (s & 1) * 8 + (int)(lane & 7), pyi = ty * LG_TILE + (s >> 1) * 8 + (int)(lane >> 3);
        const bool inside = pxi < W && pyi < H;
        const size_t pid = (size_t)pyi * W + pxi;
        pxf[s] = (float)pxi; pyf[s] = (float)pyi;
        inside4[s] = inside;
        T[s] = inside ? final_T[pid] : 0.0f;
        last[s] = inside ? n_contrib[pid] : 0u;
        g0[s] = inside ? dL_dpix[pid] : 0.0f;
        g1[s] = inside ? dL_dpix[HW + pid] : 0.0f;
        g2[s] = inside ? dL_dpix[2 * HW + pid] : 0.0f;
        Tfb[s] = T[s] * (bgr * g0[s] + bgg * g1[s] + bgb * g2[s]);   // T_final * (bg . dL/dC): the background term of dL/dalpha
        a0[s] = a1[s] = a2[s] = la[s] = lc0[s] = lc1[s] = lc2[s] = Sd[s] = 0.0f;
        wmax = max(wmax, last[s]);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, sh));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    const uint32_t n_list = range.y - range.x;
    if (n_list == 0) return;
    // this work item = list entries [seg_lo, seg_hi) of the tile (the whole list unless it is longer than S)
    const uint32_t nseg = (n_list + (uint32_t)S - 1u) / (uint32_t)S;
    const uint32_t seg_lo = item.y * (uint32_t)S, seg_hi = min(n_list, seg_lo + (uint32_t)S);
    if (wmax > seg_hi) wmax = seg_hi;
    if (item.y + 1u < nseg) {
        // not the last segment: start from the forward's checkpoints instead of the end of the list.  T = transmittance at
        // the end of this segment; colour behind = (colour accumulated inside all later segments) / T -- a quotient of a sum
        // of non-negative terms, as well conditioned as the published back-to-front accumulation.
        const float4* cr = ckpt + (size_t)2 * (range.x / (uint32_t)S) * 256;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t pix = ((uint32_t)(s >> 1) * 8u + (lane >> 3)) * 16u + (uint32_t)(s & 1) * 8u + (lane & 7u);
            const float4 here = cr[(size_t)item.y * 256 + pix];
            float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
            for (uint32_t j = nseg - 1u; j > item.y; j--) {            // back to front, like the replay itself
                const float4 r = cr[(size_t)j * 256 + pix];
                b0 += r.y; b1 += r.z; b2 += r.w;
            }
            if (inside4[s]) {
                const float inv = 1.0f / here.x;
                T[s] = here.x;
                a0[s] = b0 * inv; a1[s] = b1 * inv; a2[s] = b2 * inv;
                Sd[s] = (b0 * g0[s] + b1 * g1[s] + b2 * g2[s]) * inv;
            }
        }
    }
    const float tbx = (float)(tx * LG_TILE), tby = (float)(ty * LG_TILE);
    float4* rows = reinterpret_cast<float4*>(part);

    // every list entry of the tile writes exactly one 48-byte row (zeros when nothing contributed) at its
    // PRE-SORT slot, where the rows of one Gaussian are contiguous: no zero-fill pass, no atomics, and K9
    // reads its rows sequentially and sums them in a fixed order (deterministic gradients)
    for (int k = (int)((seg_hi - 1) / LG_Q); k >= (int)(seg_lo / LG_Q); k--) {
        const uint32_t base = range.x + (uint32_t)k * LG_Q;
        const uint32_t nbt = min((uint32_t)LG_Q, seg_hi - (uint32_t)k * LG_Q);                             // entries of this batch
        const uint32_t nb = wmax > (uint32_t)k * LG_Q ? min((uint32_t)LG_Q, wmax - (uint32_t)k * LG_Q) : 0u; // ... that any pixel reached
        uint64_t hitmask = 0;
        // lane <-> list entry base + lane for the whole batch: its Gaussian id (low field of the sorted key) and, issued
        // here so that the gather completes behind the blending below, its tile rectangle for the row address
        uint32_t id = 0;
        uint4 trect = make_uint4(0, 0, 0, 0);
        if (lane < nbt) {
            id = (uint32_t)entries[base + lane] & gid_mask;
            trect = tinfo[id];
        }
        if (nb > 0) {
            // sub-block overlap masks of the whole batch as four 64-bit SCALAR masks (bit j = entry j overlaps sub-block s): the
            // walk below tests bits and jumps from set bit to set bit -- no per-entry LDS read + readfirstlane round trip
            bool hit[4] = {false, false, false, false};
            if (lane < nb) {
                const float4 r0 = rec[LG_REC_F4 * (size_t)id], r1 = rec[LG_REC_F4 * (size_t)id + 1], r2 = rec[LG_REC_F4 * (size_t)id + 2];
                const LgReach reach = lg_reach(r0, r1, r2);
#pragma unroll
                for (int s = 0; s < 4; s++) hit[s] = lg_block_hit(r0, r1, r2, reach, tbx + (float)((s & 1) * 8), tby + (float)((s >> 1) * 8));
                q0[lane] = r0; q1[lane] = r1; q2[lane] = r2.x;
            }
            const uint64_t M0 = __ballot(hit[0]), M1 = __ballot(hit[1]), M2 = __ballot(hit[2]), M3 = __ballot(hit[3]);
            __builtin_amdgcn_wave_barrier();
            // (occupancy sweep, early request of the next record: EXPERIMENTS.md, "K7 structure")
            for (uint64_t any = (M0 | M1) | (M2 | M3); any != 0;) {
                const int jcur = 63 - __builtin_clzll(any);            // back to front
                any &= ~(1ull << jcur);
                const float4 a = q0[jcur], b = q1[jcur];
                const float4 c = make_float4(q2[jcur], 0.0f, 0.0f, 0.0f);
                const uint32_t m = (uint32_t)((M0 >> jcur) & 1ull) | ((uint32_t)((M1 >> jcur) & 1ull) << 1) | ((uint32_t)((M2 >> jcur) & 1ull) << 2) |
                                   ((uint32_t)((M3 >> jcur) & 1ull) << 3);
                const uint32_t rel = (uint32_t)k * LG_Q + (uint32_t)jcur + 1u;
                float p[9];
                uint64_t cmask = 0;                        // lanes for which the entry contributed in any sub-block (scalar)
                if (EXACT) {
#pragma unroll
                    for (int v = 0; v < 9; v++) p[v] = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        if (m & (1u << s)) {
                            bool cb = false;
                            if (rel <= last[s])
                                cb = bwd_pair<true>(a, b, c, pxf[s], pyf[s], T[s], Tfb[s], g0[s], g1[s], g2[s], a0[s], a1[s], a2[s],
                                                    la[s], lc0[s], lc1[s], lc2[s], p);
                            cmask |= __ballot(cb);
                        }
                    }
                } else {
                    // (dispatch on the lowest hit sub-block instead of zeroing the sums: EXPERIMENTS.md, "K7 structure")
#pragma unroll
                    for (int v = 0; v < 9; v++) p[v] = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; s++)
                        if (m & (1u << s))
                            cmask |= bwd_pair_fast(a, b, c, rel <= last[s], pxf[s], pyf[s], T[s], Tfb[s], g0[s], g1[s], g2[s], Sd[s], p);
                }
                if (cmask != 0) {
                    wave_reduce9_via_lds(p, red, stage, (uint32_t)jcur, lane);
                    hitmask |= 1ull << jcur;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < nbt) {
            float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = o0;
            if ((hitmask >> lane) & 1ull) {
                const float* src = stage + lane;          // value-major: [9][LG_Q]
                o0 = make_float4(src[0], src[LG_Q], src[2 * LG_Q], src[3 * LG_Q]);
                o1 = make_float4(src[4 * LG_Q], src[5 * LG_Q], src[6 * LG_Q], src[7 * LG_Q]);
                o2 = make_float4(src[8 * LG_Q], 0.0f, 0.0f, 0.0f);
            }
            float4* dst = rows + 3 * (size_t)lg_slot_of(trect, tx, ty);
            // (48 bytes to a slot of its own per lane: three partial-line writes per instance.  Upper bound of what they cost, measured by
            //  leaving them out: K7 0.683 -> 0.654 ms -- not worth an inverse map that would let them go out in list order.)
            dst[0] = o0; dst[1] = o1; dst[2] = o2;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// (K7 with the splats on the lanes -- lg_blend_bwd_splat, round 5: EXPERIMENTS.md, "K7 structure")

// diagnostics: K7's wave reduction (wave_reduce9_via_lds) on one wave (64 x 9 inputs -> 9 sums); used by tests/test_gpu_parity.py
__global__ void lg_debug_reduce9_kernel(const float* __restrict__ in, float* __restrict__ out)
{
    __shared__ float dst[9 * LG_Q];
    float p[9];
    for (int c = 0; c < 9; c++) p[c] = in[threadIdx.x * 9 + c];
    __shared__ __attribute__((aligned(16))) float red[LG_RED_FLOATS];
    wave_reduce9_via_lds(p, red, dst, 0u, threadIdx.x);
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    if (threadIdx.x < 9) out[threadIdx.x] = dst[threadIdx.x * LG_Q];
}
