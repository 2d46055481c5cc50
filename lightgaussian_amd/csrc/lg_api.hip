// lg_api.hip -- C ABI (include/lightgaussian.h) of the gfx950 (CDNA4, wave64) LightGaussian rasterizer.
// The single translation unit of liblightgaussian_hip.so; the kernels live in the headers it includes:
//   lg_math.h        scalar float arithmetic shared with the CPU test harness (canonical operation order)
//   lg_host.h        error strings, optional hipEvent profiler, scratch carving (GeomView / ImgView / BinView)
//   lg_wave.h        wave64 primitives (DPP / permlane reductions)
//   lg_preprocess.h  K1 lg_preprocess<RAW>, K8+K9 lg_preprocess_bwd<RAW>            (per Gaussian, HBM-bound)
//   lg_binning.h     K2 lg_scan_blocks, K3 lg_duplicate, lg_tile_sort / _long (second sort stage), lg_tile_ranges (one-stage cross-check only),
//                    lg_work_order (K2-K5 all hand-written; lg_sort.h = K4)
//   lg_loss.h        lg_loss_fwd / lg_loss_bwd: fused L1 + SSIM of the training step             (a wave per 64-column strip, register ring)
//   lg_prune.h       lg_select_pass, lg_v_imp_score_kernel, lg_prune_mask_kernel: device-resident prune epilogue (radix selects)
//   lg_knn.h         distCUDA2 (simple-knn): exact 3-nearest-neighbour mean squared distance on a multi-level uniform grid
//   lg_compact.h     lg_compact_plan / lg_compact_rows: one scan + one launch compacting all Gaussian tensors after a prune
//   lg_vq.h          lg_vq_nearest: nearest-code search of the VecTree quantiser on f32 MFMA (32x32x2), fused row argmin
//   lg_blend.h       K6 lg_blend_fwd<COUNT,FSCORE,EXACT>, lg_score_kernel, K7 lg_blend_bwd<EXACT>   (per tile, VALU-bound)
//
// Pipeline of one view:
//   K1 project + EWA + SH->RGB + exact footprint culling  ->  K2 scan of instance counts, blocking read of R
//   K3 packed keys tile|depth|id  ->  K4 stable keys-only radix passes over the TILE bits (the last one also leaves the tile ranges)
//   K5b every tile's list ordered by depth inside LDS (lg_tile_sort); the sorted keys ARE the per-tile lists (no id / slot arrays)
//   K6 front-to-back blend (4 autonomous waves per 16x16 tile, LDS queue, select-based pair step, ballot early exit)
//   K7 back-to-front replay (1 wave per tile, longest lists first, 4 px/lane, packed permlane reduction, one 48-B
//      gradient row per instance at its pre-sort slot, recomputed from the Gaussian's tile rectangle)
//   K9 per-Gaussian gather of its contiguous rows + cov2D/cov3D/projection/SH backward
// Written for wave64; no CUDA compatibility paths.
#include "lg_host.h"
#include "lg_wave.h"
#include "lg_preprocess.h"
#include "lg_binning.h"
#include "lg_blend.h"
#include "lg_loss.h"
#include "lg_prune.h"
#include "lg_knn.h"
#include "lg_compact.h"
#include "lg_vq.h"

// ------------------------------------------------------------------------------------------------
// host side
static int check_args(const lg_view* v, const lg_gaussians* g)
{
    if (!v || !g) return fail(LG_ERR_INVALID_ARGUMENT, "null view/gaussians");
    if (g->N < 0 || v->image_width <= 0 || v->image_height <= 0) return fail(LG_ERR_INVALID_ARGUMENT, "bad sizes");
    if (v->segment_length != 0 && (v->segment_length < 64 || v->segment_length % 64 != 0))
        return fail(LG_ERR_INVALID_ARGUMENT, "lg_view.segment_length must be 0 (default 512) or a multiple of 64");
    if ((v->flags & LG_FLAG_LONG_SERIAL) && (v->flags & LG_FLAG_LONG_PARALLEL))
        return fail(LG_ERR_INVALID_ARGUMENT, "LG_FLAG_LONG_SERIAL and LG_FLAG_LONG_PARALLEL exclude each other");
    if (g->N == 0) return LG_OK; // nothing to validate against: empty tensors carry no pointers
    if (g->N >= (1 << LG_ID_BITS)) return fail(LG_ERR_INVALID_ARGUMENT, "more than 2^29-1 Gaussians (the blend record packs the id in 29 bits)");
    if ((g->shs == nullptr) == (g->colors_precomp == nullptr))
        return fail(LG_ERR_INVALID_ARGUMENT, "Please provide excatly one of either SHs or precomputed colors!");
    const bool sr = g->scales != nullptr && g->rotations != nullptr;
    if ((g->scales != nullptr) != (g->rotations != nullptr) || sr == (g->cov3D_precomp != nullptr))
        return fail(LG_ERR_INVALID_ARGUMENT, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (g->shs_rest && !(v->flags & LG_FLAG_RAW_PARAMS)) return fail(LG_ERR_INVALID_ARGUMENT, "shs_rest needs LG_FLAG_RAW_PARAMS");
    if ((v->flags & LG_FLAG_RAW_PARAMS) && (g->cov3D_precomp || g->colors_precomp))
        return fail(LG_ERR_INVALID_ARGUMENT, "LG_FLAG_RAW_PARAMS takes raw scales/rotations/opacities and SH tensors only");
    if ((v->flags & LG_FLAG_RAW_PARAMS) && g->shs && g->M > 1 && !g->shs_rest)
        return fail(LG_ERR_INVALID_ARGUMENT, "LG_FLAG_RAW_PARAMS with M > 1 needs shs (dc) and shs_rest");
    if (g->shs) {
        if (!(g->M == 1 || g->M == 4 || g->M == 9 || g->M == 16)) return fail(LG_ERR_INVALID_ARGUMENT, "M must be 1, 4, 9 or 16");
        if (v->sh_degree < 0 || v->sh_degree > 3 || (v->sh_degree + 1) * (v->sh_degree + 1) > g->M)
            return fail(LG_ERR_INVALID_ARGUMENT, "sh_degree needs (D+1)^2 <= M, D <= 3");
    }
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->campos || !g->means3D || !g->opacities)
        return fail(LG_ERR_INVALID_ARGUMENT, "missing required pointer");
    const int gx = (v->image_width + LG_TILE - 1) / LG_TILE, gy = (v->image_height + LG_TILE - 1) / LG_TILE;
    if (gx >= 65536 || gy >= 65536) return fail(LG_ERR_INVALID_ARGUMENT, "image too large");
    return LG_OK;
}

// 64-byte pinned host slots for the forward's read-back, recycled through a process-wide free list (a thread_local slot
// would be allocated -- and leaked -- by every short-lived host thread of the views-in-flight helpers).
static std::mutex g_pin_mu;
static std::vector<std::pair<uint32_t*, hipEvent_t>> g_pin_free;
struct PinnedSlot {
    uint32_t* p = nullptr;
    hipEvent_t ev = nullptr;   // marks the copy into p (validated bounded forward: the host waits for THIS, not for the stream)
    PinnedSlot()
    {
        {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            if (!g_pin_free.empty()) { p = g_pin_free.back().first; ev = g_pin_free.back().second; g_pin_free.pop_back(); }
        }
        if (!p) {
            if (hipHostMalloc((void**)&p, 64, hipHostMallocDefault) != hipSuccess) { p = nullptr; return; }
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(p); p = nullptr; ev = nullptr; }
        }
    }
    ~PinnedSlot()
    {
        if (p) { std::lock_guard<std::mutex> lk(g_pin_mu); g_pin_free.emplace_back(p, ev); }
    }
    PinnedSlot(const PinnedSlot&) = delete;
    PinnedSlot& operator=(const PinnedSlot&) = delete;
};

#define KCHECK(name)                                                                         \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) return fail(LG_ERR_DEVICE, name " launch", _e);                \
        if (debug) {                                                                         \
            _e = hipStreamSynchronize(stream);                                               \
            if (_e != hipSuccess) return fail(LG_ERR_DEVICE, name " execution", _e);         \
        }                                                                                    \
    } while (0)

// Layout of the sort key of one view: tile | (depth bits - bias) >> store_drop | Gaussian id.  Exact forward: from the
// read-back depth maximum.  Bounded forward: from the caller's depth bound (nothing is read back).
#define LG_STATUS_PENDING 0xFFFFFFFFu   // sentinel of the host-visible status word 0 (a real abort word has only its low bits set)
#define LG_NARROW_KEY_BITS 40   // LG_FLAG_NARROW_KEY (cross-check): lay the key out as if only this many bits were available
struct KeyPlan {
    int tile_bits, gid_bits, depth_bits;   // field widths; depth_bits = width of the FULL depth pattern (minus bias) of this view
    int store_drop;                        // low depth bits that are not stored in the key (the fields exceed 64 bits): 0 at C3
    bool two_stage;                        // radix passes on the tile bits only + lg_tile_sort (default); false = LG_FLAG_SORT_ALL_BITS
    uint32_t gid_mask;
    int stored() const { return depth_bits - store_drop; }
    int tile_shift() const { return gid_bits + stored(); }
    // bit span of the global radix passes: the tile field (at least one bit: a single-tile image still needs its keys moved to the
    // output buffer), preceded by every stored depth bit in the one-stage scheme
    int sort_begin() const { return two_stage ? tile_shift() : gid_bits; }
    int sort_end() const { return tile_shift() + (tile_bits > 0 ? tile_bits : 1); }
};
static KeyPlan make_key_plan(int ntiles, int N, uint32_t dmax_bits, uint32_t flags)
{
    KeyPlan k;
    k.tile_bits = bits_for((uint32_t)ntiles);
    k.gid_bits = bits_for((uint32_t)(N > 1 ? N : 2));
    const uint32_t dspan = dmax_bits > LG_DEPTH_BIAS ? dmax_bits - LG_DEPTH_BIAS : 0u;
    k.depth_bits = bits_for(dspan + 1u) > 0 ? bits_for(dspan + 1u) : 1;
    // The three fields must fit 64 bits.  When they do not (6 M Gaussians at 3840x2160: 15 + 27 + 23 = 65; 20 M at 1080p; ...)
    // the lowest depth bits are left out of the key and the tile sort reads the full depth pattern from the binning record
    // (tinfo) -- r2 fell back to a (tile << 32 | depth, id) pair sort through hipCUB there, without the bounded
    // forward, the graph and the fused histograms.  At least one depth bit is always stored (tile <= 32 bits, id <= 29).
    const int avail = (flags & LG_FLAG_NARROW_KEY) ? LG_NARROW_KEY_BITS : 64;
    k.store_drop = std::max(0, std::min(k.depth_bits - 1, k.tile_bits + k.depth_bits + k.gid_bits - avail));
    // Two-stage sort (default, round 3): the global radix passes cover the tile bits only (13 bits at 1080p: two 8-bit passes
    // instead of the four that tile + 19 depth bits took in round 2) and lg_tile_sort orders every list on ALL depth bits inside
    // LDS.  LG_FLAG_SORT_ALL_BITS keeps the one-stage scheme -- every stored bit through the global passes, lg_tile_ranges
    // finishing the bits a key beyond 64 bits does not store -- as an independent cross-check.
    k.two_stage = !(flags & LG_FLAG_SORT_ALL_BITS);
    k.gid_mask = k.gid_bits >= 32 ? 0xFFFFFFFFu : ((1u << k.gid_bits) - 1u);
    return k;
}

// Arguments of the capacity-bounded forward (lg_forward_bounded); NULL = exact forward with its one read-back.
struct Bounded { void* binning; int64_t capacity; float max_depth; uint32_t* status; uint32_t* host_status; };

static int forward_impl(const lg_view* v, const lg_gaussians* g, void* geom_p, void* img_p, lg_alloc_fn alloc, void* alloc_user,
                        const Bounded* bounded, int weight_policy, float* out_color, int32_t* out_radii, int32_t* out_count,
                        float* out_score, void** binning_out, int64_t* num_rendered, void* stream_p)
{
    int rc = check_args(v, g);
    if (rc != LG_OK) return rc;
    if (!geom_p || !img_p || !out_color || (!out_radii && g->N > 0) || (!alloc && !bounded)) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    const bool count = out_count != nullptr;
    if (count && !out_score) return fail(LG_ERR_INVALID_ARGUMENT, "count needs score");
    if (count && (weight_policy < 0 || weight_policy > 3)) return fail(LG_ERR_INVALID_ARGUMENT, "bad weight policy");
    // per-hit weights are summed in Q24.40 per view: a Gaussian can collect at most 0.99 per pixel
    if (count && weight_policy >= LG_WEIGHT_ALPHA && (int64_t)v->image_width * v->image_height > (1ll << 24))
        return fail(LG_ERR_INVALID_ARGUMENT, "ALPHA / ALPHA_T weights: images beyond 2^24 pixels overflow the Q24.40 per-view sums");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = v->flags & LG_FLAG_DEBUG, prof = v->flags & LG_FLAG_PROFILE, fast = v->flags & LG_FLAG_FAST_EXP;
    const int N = g->N, W = v->image_width, H = v->image_height;
    const int gx = (W + LG_TILE - 1) / LG_TILE, gy = (H + LG_TILE - 1) / LG_TILE, ntiles = gx * gy;
    const int ntiles_pad = (ntiles + LG_TILE_GRID_ALIGN - 1) / LG_TILE_GRID_ALIGN * LG_TILE_GRID_ALIGN; // grid of the per-tile kernels
    const int nblk = (N + LG_PP - 1) / LG_PP;
    GeomView geo = carve_geom(geom_p, N);
    ImgView img = carve_img(img_p, W, H);
    const int S = lg_segment_of(v);     // list entries per segment of a long tile (checkpoints for the backward): part of the view
    if (binning_out) *binning_out = nullptr;
    if (num_rendered) *num_rendered = 0;

    KeyPlan kp{};
    BinView bin{};
    int64_t cap = 0;          // instances the binning buffer holds: R itself (exact) or the caller's capacity (bounded)
    if (bounded) {
        if (!bounded->binning || bounded->capacity <= 0 || bounded->capacity >= (1ll << 30) || !(bounded->max_depth > 0.2f))
            return fail(LG_ERR_INVALID_ARGUMENT, "lg_forward_bounded: binning buffer, 0 < max_rendered < 2^30 and max_depth > 0.2 required");
        kp = make_key_plan(ntiles, N, __builtin_bit_cast(uint32_t, bounded->max_depth), v->flags);
        cap = bounded->capacity;
        bin = carve_bin(bounded->binning, cap, W, H, S);
        if (binning_out) *binning_out = bounded->binning;
        if (num_rendered) *num_rendered = cap;
        if (N == 0) {
            HIP_TRY(lg_zero_async(geo.counters, 16, stream));
            if (bounded->status) HIP_TRY(lg_zero_async(bounded->status, 16, stream));
            HIP_TRY(lg_zero_async(bin.ranges, (size_t)ntiles * 8, stream));
        }
    }
    // validated bounded forward: K2 writes its four status words STRAIGHT into pinned host memory (system-scope release on word 0)
    // and the host waits for word 0 to leave its sentinel once everything of the view has been
    // enqueued -- no device-to-host copy node behind K2 (a 4 us blit kernel + its launch gap on the critical path of every view)
    PinnedSlot vslot;
    const bool host_words = bounded && bounded->host_status;
    if (host_words) {
        if (!vslot.p) return fail(LG_ERR_ALLOC, "hipHostMalloc of the status slot failed");
        if (N > 0) { vslot.p[1] = vslot.p[2] = vslot.p[3] = 0u; __atomic_store_n(&vslot.p[0], LG_STATUS_PENDING, __ATOMIC_RELEASE); }
        else memset(vslot.p, 0, 16);
    }
    // bounded forward: K1 also clears the sort's histograms / tickets / states (the buffer and the key layout are known already)
    uint32_t* k1_clear = nullptr;
    uint32_t k1_nclear = 0;
    if (bounded && N > 0 && cap > 0) {
        const LgSortLayout SL0 = lg_sort_layout((size_t)cap);
        const int sb = kp.sort_begin(), se = kp.sort_end();
        k1_clear = (uint32_t*)bin.sort_temp;
        k1_nclear = (uint32_t)(lg_sort_clear_bytes(SL0, (unsigned)((se - sb + 7) / 8)) / 4);
    }
    if (N > 0) {
        {
            ProfScope ps(prof, "preprocess", stream);
            // K1 runs faster with FEWER waves in flight where it reads SH rows: unused dynamic LDS caps it at 12 waves per CU there (the sweep,
            // K9's opposite behaviour and the significance pass's A/B: EXPERIMENTS.md, "K1 / K9")
            // (the count / score accumulators are cleared here for the integer weights; the per-hit policies accumulate in the slots and
            //  lg_score_slots writes both outputs)
            int32_t* const k1_zero_count = (count && weight_policy >= LG_WEIGHT_ALPHA) ? nullptr : out_count;
            const size_t k1_dyn = (g->shs && !g->colors_precomp && !(v->flags & LG_FLAG_SKIP_COLOR)) ? LG_K1_PAD_LDS : 0;
#define LAUNCH_PP(RAWP, DIR)                                                                                                         \
    lg_preprocess<RAWP, DIR><<<nblk, LG_PP, (DIR) ? k1_dyn : 0, stream>>>(N, g->M, v->sh_degree, W, H, v->tanfovx, v->tanfovy,                          \
                                                                      v->scale_modifier, v->prefiltered, (v->flags & LG_FLAG_SKIP_COLOR) ? 1 : 0, v->viewmatrix, v->projmatrix, \
                                                                      v->campos, g->means3D, g->shs, g->shs_rest, g->colors_precomp,   \
                                                                      g->opacities, g->scales, g->rotations, g->cov3D_precomp, geo, out_radii, k1_zero_count, out_score, \
                                                                      k1_clear, k1_nclear, (v->flags & LG_FLAG_SAVE_SH_JACOBIAN) ? 1 : 0)
            // SH rows are read directly by their lanes (dword-aligned dwordx4 loads); LG_K1_LDS=1 selects the LDS-staged reads
            const bool direct = !(v->flags & LG_FLAG_K1_LDS);
            const bool raw = v->flags & LG_FLAG_RAW_PARAMS;
            if (raw && direct) LAUNCH_PP(true, true);
            else if (raw) LAUNCH_PP(true, false);
            else if (direct) LAUNCH_PP(false, true);
            else LAUNCH_PP(false, false);
#undef LAUNCH_PP
        }
        KCHECK("lg_preprocess");
        {
            // K2: one workgroup scans the per-K1-workgroup instance counts and reduces the depth maxima (replaces a
            // device-wide scan of N words + a separate reduction kernel)
            ProfScope ps(prof, "scan", stream);
            lg_scan_blocks<<<(nblk + LG_PART - 1) / LG_PART, LG_PART, 0, stream>>>(nblk, geo.blk_sum, geo.blk_dmax, geo.blk_off, geo.part_sum,
                                                                                  geo.part_dmax, geo.part_prefix, geo.counters + 8,
                                                                                  bounded ? (uint32_t)cap : 0xFFFFFFFFu,
                                                                                  bounded ? kp.depth_bits : 32, geo.counters,
                                                                                  host_words ? vslot.p : (bounded ? (uint32_t*)bounded->status : nullptr));
        }
        KCHECK("lg_scan_blocks");
    }
    int64_t R = cap;
    if (!bounded) {
        uint32_t h_counters[4] = {0, 0, 0, 0};
        if (N > 0) {
            // The exact forward has ONE blocking read-back: the instance count R (it sizes the binning buffer), with the
            // depth maximum and the prefiltered flag riding along in one 16-byte copy into pinned host memory.
            // lg_forward_bounded has none.
            PinnedSlot slot;                                      // process-wide pool: host threads come and go (views in flight)
            if (!slot.p) return fail(LG_ERR_ALLOC, "hipHostMalloc of the read-back slot failed");
            HIP_TRY(hipMemcpyAsync(slot.p, geo.counters, 16, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (int k = 0; k < 4; k++) h_counters[k] = slot.p[k];
            if (v->prefiltered && h_counters[1]) return fail(LG_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
            if (h_counters[0] & 1u) return fail(LG_ERR_INVALID_ARGUMENT, "more than 2^32-1 tile instances in one view");
        }
        R = h_counters[3];
        if (R >= (1ll << 30)) return fail(LG_ERR_INVALID_ARGUMENT, "more than 2^30-1 tile instances in one view");
        kp = make_key_plan(ntiles, N, h_counters[2], v->flags);
        void* bin_p = alloc(alloc_user, carve_bin(nullptr, R, W, H, S).total);
        if (!bin_p) return fail(LG_ERR_ALLOC, "binning allocator returned NULL");
        if (binning_out) *binning_out = bin_p;
        bin = carve_bin(bin_p, R, W, H, S);
        cap = R;
        if (num_rendered) *num_rendered = R;
        if (R == 0) HIP_TRY(lg_zero_async(bin.ranges, (size_t)ntiles * 8, stream)); // otherwise cleared by lg_duplicate
    } else {
        // (bounded->status is written by lg_scan_blocks itself, or cleared above for N == 0: no copy node)
        // (validated mode: K2 wrote the words into vslot.p itself; the host looks at them at the end of this function)
    }
    g_stats.num_rendered = bounded ? -1 : R;
    g_stats.num_visible = -1; // not tracked on the device (see lg_preprocess); callers count radii > 0

    const int sort_begin = kp.sort_begin(), sort_end = kp.sort_end();
    if (cap > 0 && N > 0) {
        const LgSortLayout SL = lg_sort_layout((size_t)cap);
        // one clear for the digit histograms, the tile tickets and the look-back states of every radix pass (exact forward: the
        // buffer was allocated a moment ago; the bounded forward's K1 did it already)
        if (!k1_clear) HIP_TRY(lg_zero_async(bin.sort_temp, lg_sort_clear_bytes(SL, (unsigned)((sort_end - sort_begin + 7) / 8)), stream));
        uint32_t* hist = (uint32_t*)((char*)bin.sort_temp + SL.hist_off);
        {
            ProfScope ps(prof, "duplicate", stream);
            const int dgrid = std::max(1, std::min((nblk + 4 * LG_DUP_WAVES - 1) / (4 * LG_DUP_WAVES), LG_DUP_GRID));
            lg_duplicate<<<dgrid, LG_DUP_THREADS, 0, stream>>>(N, nblk, gx, kp.stored(), kp.store_drop, kp.gid_bits, sort_begin, sort_end, (uint32_t)cap, geo.touched,
                                                              geo.blk_off, geo.part_prefix, geo.counters, geo.offsets, geo.tinfo, bin.keys_in, ntiles, bin.ranges, hist,
                                                              kp.two_stage ? 0xFFFFFFFFu : 0u, bin.long_tiles);
        }
        KCHECK("lg_duplicate");
        {
            ProfScope ps(prof, "sort", stream);
            size_t tb = bin.sort_temp_bytes;
            // two-stage scheme: the last pass also leaves the tile ranges (no lg_tile_ranges launch)
            HIP_TRY(lg_sort_keys(bin.sort_temp, tb, bin.keys_in, bin.entries, (uint32_t)cap, sort_begin, sort_end, geo.counters, true, stream,
                                 LG_SORT_POLL_BUDGET, kp.two_stage ? bin.ranges : nullptr, kp.tile_shift()));
        }
        KCHECK("lg_sort_keys");
        if (!kp.two_stage) {
            ProfScope ps(prof, "tile_ranges", stream);
            const uint32_t rgrid = (uint32_t)((cap + 255) / 256);
            lg_tile_ranges<<<rgrid, 256, 0, stream>>>(geo.counters, kp.tile_shift(), kp.gid_bits, kp.gid_mask, kp.two_stage ? 0 : kp.store_drop, kp.store_drop,
                                                      bin.entries, bin.keys_in, geo.tinfo, bin.ranges, bounded ? bounded->status : nullptr);
        }
        KCHECK("lg_tile_ranges");
        if (kp.two_stage) {
            // second stage: one WAVE per tile orders its list by depth in LDS (lists up to 1024 entries; up to 4096: a whole workgroup,
            // same launch); the few longer ones go through a persistent grid of 1024-thread workgroups (an empty launch otherwise)
            ProfScope ps(prof, "tile_sort", stream);
            lg_tile_sort<<<(ntiles + 3) / 4 + ntiles, LG_TS_THREADS, 0, stream>>>(ntiles, geo.counters, bin.ranges, bin.entries, kp.gid_bits, kp.gid_mask, kp.store_drop,
                                                                                  kp.depth_bits, geo.tinfo, bin.long_tiles, bounded ? bounded->status : nullptr);
            lg_tile_sort_long<<<std::min(ntiles, LG_TL_GRID), LG_TL_THREADS, 0, stream>>>(geo.counters, bin.ranges, bin.entries, bin.keys_in, kp.gid_bits, kp.gid_mask,
                                                                                          kp.store_drop, kp.depth_bits, geo.tinfo, bin.long_tiles);
            KCHECK("lg_tile_sort");
        }
    }
    const uint32_t gid_mask = kp.gid_mask;
    // (count / score accumulators of the count variant were cleared by lg_preprocess)
    // Long tiles of the hardware-exp colour forward: which lists go through the parallel kernels below is decided ON THE DEVICE
    // from this view's own instance count (lg_par_min, lg_binning.h) -- no history, no host hint: two renders of the same
    // inputs run the same kernels on the same lists whatever the process rendered before.  LG_FLAG_LONG_SERIAL / _PARALLEL
    // override the default rule per call; the canonical / count variants always walk serially (bit-pinned).
    // Round 5: the significance-only pass (count forward, canonical arithmetic, no colour, integer weights) has a parallel long-tile walk
    // of its own (lg_count_seg / _rewalk / _fixup: bit-identical counts through interval comparisons + an exact fix-up); count forwards
    // that return an image and the float weight policies walk serially.
    const bool cnt_par = count && !fast && (v->flags & LG_FLAG_SKIP_COLOR) && (weight_policy == LG_WEIGHT_ONE || weight_policy == LG_WEIGHT_OPACITY);
    // The default rule ("auto") applies to the colour forward only.  For the significance pass it was measured and lost (heavy-tailed scene,
    // four views in flight as prune_list_sharded runs them: 1150 views/s against 1497 serial): the serial walk of a pile stops
    // early in every wave whose pixels saturate, the parallel one walks every segment twice, and with other views in flight the device is
    // never idle behind the one long walk -- total work decides, not the critical path.  LG_FLAG_LONG_PARALLEL selects it explicitly.
    const int long_mode = (!count && fast && cap > 0 && N > 0) ? ((v->flags & LG_FLAG_LONG_SERIAL) ? 0 : (v->flags & LG_FLAG_LONG_PARALLEL) ? 2 : 1)
                        : (cnt_par && cap > 0 && N > 0 && (v->flags & LG_FLAG_LONG_PARALLEL)) ? 2 : 0;
    const bool par_long = long_mode != 0;
    {
        ProfScope ps(prof, count ? "blend_fwd_count" : "blend_fwd", stream);
        // + 1: the last workgroup builds the backward's work list from the tile ranges (colour forwards only: the
        // significance-only pass has no backward)
        const bool nocolor_pass = count && !fast && (v->flags & LG_FLAG_SKIP_COLOR);
        dim3 grid(ntiles_pad + ((nocolor_pass && !par_long) ? 0 : 1)), block(256);     // (the parallel long-tile walk needs the par_work list)
#define LAUNCH_FWD(CNT, FS, EX, COL)                                                                                                 \
    lg_blend_fwd<CNT, FS, EX, COL><<<grid, block, 0, stream>>>(W, H, gx, ntiles, ntiles_pad, bin.ranges, bin.entries, gid_mask, geo.rec, v->bg, \
                                                         out_color, img.final_T, img.n_contrib, out_count, (unsigned long long*)bin.keys_in, geo.tinfo, (uint32_t)cap, S, bin.ckpt, bin.work, bin.meta, bin.par_work, geo.counters, long_mode, bin.par_arrived)
        // per-hit weights (ALPHA / ALPHA_T): the kernel is instantiated per policy and adds {count | Q8.40 weight} words into the instances'
        // pre-sort slots -- the radix sort's input buffer, free since lg_tile_sort and cleared here
        const int fs = !count ? 0 : weight_policy == LG_WEIGHT_ALPHA ? 2 : weight_policy == LG_WEIGHT_ALPHA_T ? 3 : 0;
        // (the significance-only variant writes every slot exactly once -- its waves merge in LDS -- and needs no clear)
        if (fs && cap > 0 && N > 0 && !(!fast && (v->flags & LG_FLAG_SKIP_COLOR))) HIP_TRY(lg_zero_async(bin.keys_in, (size_t)cap * 8, stream));
        const bool nocolor = count && !fast && (v->flags & LG_FLAG_SKIP_COLOR);   // significance-only pass: no colour, no per-pixel outputs
        if (!count) { if (fast) LAUNCH_FWD(false, 0, false, true); else LAUNCH_FWD(false, 0, true, true); }
        else if (nocolor) {
            if (fs == 2) LAUNCH_FWD(true, LG_W_ALPHA, true, false);
            else if (fs == 3) LAUNCH_FWD(true, LG_W_ALPHA_T, true, false);
            else LAUNCH_FWD(true, 0, true, false);
        }
        else if (!fs) { if (fast) LAUNCH_FWD(true, 0, false, true); else LAUNCH_FWD(true, 0, true, true); }
        else if (fs == 2) { if (fast) LAUNCH_FWD(true, LG_W_ALPHA, false, true); else LAUNCH_FWD(true, LG_W_ALPHA, true, true); }
        else { if (fast) LAUNCH_FWD(true, LG_W_ALPHA_T, false, true); else LAUNCH_FWD(true, LG_W_ALPHA_T, true, true); }
#undef LAUNCH_FWD
    }
    KCHECK("lg_blend_fwd");
    // (a second HIP stream for the long-tile chain / for the memory-bound front of other views: EXPERIMENTS.md, "streams")
    if (par_long && cnt_par) {
        ProfScope ps(prof, "blend_fwd_count_long", stream);
        const uint32_t pgrid = (uint32_t)std::min<int64_t>((int64_t)ntiles + cap / S + 1, LG_PAR_GRID);
        const float band_mul = (v->flags & LG_FLAG_COUNT_WIDE_BAND) ? 4096.0f : 1.0f;
        lg_count_seg<<<pgrid, 256, 0, stream>>>(W, H, gx, S, bin.par_work, bin.meta, bin.ranges, bin.entries, gid_mask, geo.rec, bin.ckpt, bin.ckpt_last, bin.par_arrived, band_mul);
        lg_count_rewalk<<<pgrid, 256, 0, stream>>>(W, H, gx, S, bin.par_work, bin.meta, bin.ranges, bin.entries, gid_mask, geo.rec, bin.ckpt, bin.ckpt_last, out_count, band_mul);
        lg_count_fixup<<<std::min<uint32_t>(pgrid, 256u), 256, 0, stream>>>(W, H, gx, S, bin.par_work, bin.meta, bin.ranges, bin.entries, gid_mask, geo.rec, bin.ckpt_last, out_count);
        KCHECK("lg_count_long");
    } else if (par_long) {
        // persistent grids over the par_work list left by the forward's work-list workgroup (meta[4] items; none on scenes
        // without outlier lists: each launch is then one scalar load per workgroup)
        ProfScope ps(prof, "blend_fwd_long", stream);
        const uint32_t pgrid = (uint32_t)std::min<int64_t>((int64_t)ntiles + cap / S + 1, LG_PAR_GRID);
        // (pass 2, the per-tile scan, runs inside the first launch: the workgroup that finishes a tile's last segment does it)
        lg_blend_fwd_seg<<<pgrid, 256, 0, stream>>>(W, H, gx, S, bin.par_work, bin.meta, bin.ranges, bin.entries, gid_mask, geo.rec, bin.ckpt, bin.ckpt_last,
                                                   bin.par_arrived, v->bg, out_color, img.final_T, img.n_contrib);
        lg_blend_fwd_rewalk<<<pgrid, 256, 0, stream>>>(W, H, gx, S, bin.par_work, bin.meta, bin.ranges, bin.entries, gid_mask, geo.rec, v->bg,
                                                      out_color, img.final_T, img.n_contrib, bin.ckpt, bin.ckpt_last);
        KCHECK("lg_blend_fwd_long");
    }
    if (count && N > 0) {
        ProfScope ps(prof, "score", stream);
        if (weight_policy == LG_WEIGHT_ONE || weight_policy == LG_WEIGHT_OPACITY)
            lg_score_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, out_count, weight_policy == LG_WEIGHT_OPACITY ? g->opacities : nullptr, out_score, v->count_sum);
        else
            lg_score_slots<<<(N + 255) / 256, 256, 0, stream>>>(N, geo.touched, geo.offsets, (const unsigned long long*)bin.keys_in, (uint32_t)cap, out_count, out_score, v->count_sum, geo.counters);
        KCHECK("lg_score_kernel");
    }
    if (debug && N > 0) {
        // debug: the abort word as it stands at the END of the view (the radix sort's look-back can only report after K2)
        uint32_t h_abort = 0;
        HIP_TRY(hipMemcpyAsync(&h_abort, geo.counters, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (h_abort & LG_ABORT_SORT) return fail(LG_ERR_DEVICE, "radix sort look-back gave up (a predecessor tile never published): the view is void");
    }
    if (bounded && bounded->host_status) {
        // K2's words arrive while the blend kernels are still running.  Spin on word 0 (written last, released system-wide); every
        // few thousand polls ask the stream whether it is still working, so that a device error cannot turn into a host hang
        uint32_t w0 = __atomic_load_n(&vslot.p[0], __ATOMIC_ACQUIRE);
        for (uint64_t spins = 1; w0 == LG_STATUS_PENDING; spins++) {
            if ((spins & 0xFFFu) == 0u) {
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipErrorNotReady) {                       // stream drained (or failed): the words are final
                    w0 = __atomic_load_n(&vslot.p[0], __ATOMIC_ACQUIRE);
                    if (w0 == LG_STATUS_PENDING) return fail(LG_ERR_DEVICE, "lg_forward_bounded: the status words never arrived", q);
                    break;
                }
            }
            __builtin_ia32_pause();
            w0 = __atomic_load_n(&vslot.p[0], __ATOMIC_ACQUIRE);
        }
        bounded->host_status[0] = w0;
        for (int k = 1; k < 4; k++) bounded->host_status[k] = vslot.p[k];
        g_stats.num_rendered = vslot.p[3];
    }
    return LG_OK;
}

extern "C" int lg_forward(const lg_view* view, const lg_gaussians* g, void* geom, void* img, lg_alloc_fn alloc, void* alloc_user,
                          float* out_color, int32_t* out_radii, void** binning_out, int64_t* num_rendered, void* stream)
{
    return forward_impl(view, g, geom, img, alloc, alloc_user, nullptr, LG_WEIGHT_OPACITY, out_color, out_radii, nullptr, nullptr, binning_out,
                        num_rendered, stream);
}

extern "C" int lg_forward_count(const lg_view* view, const lg_gaussians* g, void* geom, void* img, lg_alloc_fn alloc, void* alloc_user,
                                int32_t weight_policy, float* out_color, int32_t* out_radii, int32_t* out_count, float* out_score,
                                void** binning_out, int64_t* num_rendered, void* stream)
{
    if (g && g->N > 0 && (!out_count || !out_score)) return fail(LG_ERR_INVALID_ARGUMENT, "count/score outputs required");
    return forward_impl(view, g, geom, img, alloc, alloc_user, nullptr, weight_policy, out_color, out_radii, out_count, out_score, binning_out,
                        num_rendered, stream);
}

extern "C" int lg_forward_bounded(const lg_view* view, const lg_gaussians* g, void* geom, void* img, void* binning, int64_t max_rendered,
                                  float max_depth, int32_t weight_policy, float* out_color, int32_t* out_radii, int32_t* out_count,
                                  float* out_score, uint32_t* status, uint32_t* host_status, void* stream)
{
    if (g && g->N > 0 && ((out_count == nullptr) != (out_score == nullptr))) return fail(LG_ERR_INVALID_ARGUMENT, "count and score go together");
    const Bounded b = { binning, max_rendered, max_depth, status, host_status };
    return forward_impl(view, g, geom, img, nullptr, nullptr, &b, weight_policy, out_color, out_radii, out_count, out_score, nullptr, nullptr, stream);
}

static int backward_impl(const lg_view* v, const lg_gaussians* g, const int32_t* radii, const void* geom_p, const void* bin_p,
                         const void* img_p, int64_t R, const float* dL_dcolor, float* dL_dmeans2D, float* dL_dmeans3D,
                         float* dL_dshs, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                         float* dL_dcov3D, float* dL_dshs_rest, void* scratch, void* stream_p, int chunks, lg_chunk_fn on_chunk, void* user)
{
    int rc = check_args(v, g);
    if (rc != LG_OK) return rc;
    if (g->N == 0) return LG_OK;   // empty model: torch hands over NULL pointers for empty tensors, and there is nothing to write
    // SH inputs without dL_dshs but WITH dL_dcolors: K9's rgb_only mode (dL/d rgb per Gaussian instead of the coefficient gradients;
    // lg_sh_grad_from_rgb rebuilds them) -- then dL_dshs_rest is not needed either
    const bool rgb_only = g->shs && !dL_dshs && dL_dcolors;
    if (g->shs_rest && !dL_dshs_rest && !rgb_only) return fail(LG_ERR_INVALID_ARGUMENT, "missing gradient output for shs_rest");
    if (rgb_only && dL_dshs_rest) return fail(LG_ERR_INVALID_ARGUMENT, "dL_dshs_rest without dL_dshs");
    if (!radii || !geom_p || !bin_p || !img_p || !dL_dcolor || !dL_dmeans2D || !dL_dmeans3D || !dL_dopacity || !scratch)
        return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    if ((g->shs && !dL_dshs && !rgb_only) || (g->colors_precomp && !dL_dcolors) || (g->scales && (!dL_dscales || !dL_drotations)) ||
        (g->cov3D_precomp && !dL_dcov3D))
        return fail(LG_ERR_INVALID_ARGUMENT, "missing gradient output for a provided input");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = v->flags & LG_FLAG_DEBUG, prof = v->flags & LG_FLAG_PROFILE, fast = v->flags & LG_FLAG_FAST_EXP;
    const int N = g->N, W = v->image_width, H = v->image_height;
    if (N == 0) return LG_OK;
    const int gx = (W + LG_TILE - 1) / LG_TILE, gy = (H + LG_TILE - 1) / LG_TILE, ntiles = gx * gy;
    const int ntiles_pad = (ntiles + LG_TILE_GRID_ALIGN - 1) / LG_TILE_GRID_ALIGN * LG_TILE_GRID_ALIGN; // grid of the per-tile kernels
    GeomView geo = carve_geom(const_cast<void*>(geom_p), N);
    ImgView img = carve_img(const_cast<void*>(img_p), W, H);
    const int S = lg_segment_of(v);   // must be the forward's (same lg_view); the kernels compare it with meta[2] and refuse otherwise
    BinView bin = carve_bin(const_cast<void*>(bin_p), R, W, H, S);
    const int gid_bits = bits_for((uint32_t)(N > 1 ? N : 2));          // same field width as the forward used
    const uint32_t gid_mask = gid_bits >= 32 ? 0xFFFFFFFFu : ((1u << gid_bits) - 1u);
    float* rows = (float*)scratch; // [R][12] gradient rows, every row written by lg_blend_bwd
    const uint32_t max_items = (uint32_t)(ntiles + R / S + 1);
    // (the work list of the backward blend -- one item per (tile, segment of S entries), longest first -- was left in the binning
    // buffer by the forward: one extra workgroup of lg_blend_fwd)
    if (R > 0) {
        ProfScope ps(prof, "blend_bwd", stream);
        if (fast)
            lg_blend_bwd<false><<<max_items, 64, 0, stream>>>(W, H, gx, S, bin.work, bin.meta, bin.ranges, bin.entries, gid_mask, geo.tinfo, geo.rec, v->bg,
                                                              img.final_T, img.n_contrib, dL_dcolor, bin.ckpt, rows);
        else
            lg_blend_bwd<true><<<max_items, 64, 0, stream>>>(W, H, gx, S, bin.work, bin.meta, bin.ranges, bin.entries, gid_mask, geo.tinfo, geo.rec, v->bg,
                                                             img.final_T, img.n_contrib, dL_dcolor, bin.ckpt, rows);
    }
    KCHECK("lg_blend_bwd");
    {
        // K9 runs over Gaussian ranges: `chunks` launches of consecutive 64-Gaussian workgroups.  After each launch is enqueued
        // the caller is told (on_chunk): rows [first, first + count) of every gradient tensor are final once the stream reaches
        // that point -- a data-parallel trainer starts their all-reduce there, while K9 computes the next range.
        ProfScope ps(prof, "preprocess_bwd", stream);
        const int nblk = (N + LG_PP - 1) / LG_PP;
        if (chunks < 1) chunks = 1;
        if (chunks > nblk) chunks = nblk;
        const int per = (nblk + chunks - 1) / chunks;
        for (int first_blk = 0; first_blk < nblk; first_blk += per) {
            const int nb = std::min(per, nblk - first_blk);
#define LAUNCH_PPB(RAWP, JACP)                                                                                                       \
    lg_preprocess_bwd<RAWP, JACP><<<nb, LG_PP, 0, stream>>>(                                                                                \
        N, first_blk, g->M, v->sh_degree, W, H, v->tanfovx, v->tanfovy, v->scale_modifier, v->viewmatrix, v->projmatrix, v->campos, g->means3D,  \
        g->shs, g->shs_rest, g->colors_precomp, g->opacities, g->scales, g->rotations, g->cov3D_precomp, radii, geo.rec,   \
        geo.counters, bin.meta, (uint32_t)S, geo.touched, geo.offsets, reinterpret_cast<const float4*>(rows), geo.shjac, dL_dmeans2D, dL_dmeans3D, dL_dshs, dL_dshs_rest, dL_dcolors, dL_dopacity,   \
        dL_dscales, dL_drotations, dL_dcov3D)
            // (the view of a backward is the view of its forward: LG_FLAG_SAVE_SH_JACOBIAN says K1 left the SH direction Jacobians)
            const bool jac = (v->flags & LG_FLAG_SAVE_SH_JACOBIAN) && g->shs && (dL_dshs || rgb_only);
            if (v->flags & LG_FLAG_RAW_PARAMS) { if (jac) LAUNCH_PPB(true, true); else LAUNCH_PPB(true, false); }
            else { if (jac) LAUNCH_PPB(false, true); else LAUNCH_PPB(false, false); }
#undef LAUNCH_PPB
            if (on_chunk) on_chunk(user, first_blk * LG_PP, std::min(N - first_blk * LG_PP, nb * LG_PP));
        }
    }
    KCHECK("lg_preprocess_bwd");
    if (debug && R > 0) {
        uint32_t h_seg = 0;
        HIP_TRY(hipMemcpyAsync(&h_seg, bin.meta + 2, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (h_seg != (uint32_t)S) return fail(LG_ERR_INVALID_ARGUMENT, "lg_backward: lg_view.segment_length differs from the forward's (gradients are zero)");
        if ((v->flags & LG_FLAG_SAVE_SH_JACOBIAN) && g->shs && (dL_dshs || rgb_only)) {
            uint32_t h_mark = 0;
            HIP_TRY(hipMemcpyAsync(&h_mark, geo.counters + 9, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (h_mark != LG_SHJAC_MAGIC)
                return fail(LG_ERR_INVALID_ARGUMENT, "lg_backward: the view carries LG_FLAG_SAVE_SH_JACOBIAN but its forward did not (gradients are zero)");
        }
    }
    return LG_OK;
}

extern "C" int lg_backward(const lg_view* v, const lg_gaussians* g, const int32_t* radii, const void* geom_p, const void* bin_p,
                           const void* img_p, int64_t R, const float* dL_dcolor, float* dL_dmeans2D, float* dL_dmeans3D,
                           float* dL_dshs, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                           float* dL_dcov3D, float* dL_dshs_rest, void* scratch, void* stream_p)
{
    return backward_impl(v, g, radii, geom_p, bin_p, img_p, R, dL_dcolor, dL_dmeans2D, dL_dmeans3D, dL_dshs, dL_dcolors, dL_dopacity,
                         dL_dscales, dL_drotations, dL_dcov3D, dL_dshs_rest, scratch, stream_p, 1, nullptr, nullptr);
}

extern "C" int lg_backward_chunked(const lg_view* v, const lg_gaussians* g, const int32_t* radii, const void* geom_p, const void* bin_p,
                                   const void* img_p, int64_t R, const float* dL_dcolor, float* dL_dmeans2D, float* dL_dmeans3D,
                                   float* dL_dshs, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                                   float* dL_dcov3D, float* dL_dshs_rest, void* scratch, void* stream_p, int32_t chunks,
                                   lg_chunk_fn on_chunk, void* user)
{
    return backward_impl(v, g, radii, geom_p, bin_p, img_p, R, dL_dcolor, dL_dmeans2D, dL_dmeans3D, dL_dshs, dL_dcolors, dL_dopacity,
                         dL_dscales, dL_drotations, dL_dcov3D, dL_dshs_rest, scratch, stream_p, chunks, on_chunk, user);
}

extern "C" int lg_sh_grad_from_rgb(int32_t N, int32_t M, int32_t sh_degree, int32_t V, const float* means3D, const float* campos,
                                   const float* drgb, int64_t view_stride, float divisor, int32_t accumulate, float* dL_dshs, float* dL_dshs_rest,
                                   void* stream_p)
{
    if (N < 0 || V < 1 || !(M == 1 || M == 4 || M == 9 || M == 16) || sh_degree < 0 || sh_degree > 3 || (sh_degree + 1) * (sh_degree + 1) > M)
        return fail(LG_ERR_INVALID_ARGUMENT, "lg_sh_grad_from_rgb: N >= 0, V >= 1, M in {1, 4, 9, 16}, (D + 1)^2 <= M required");
    if (N == 0) return LG_OK;
    if (!means3D || !campos || !drgb || !dL_dshs || view_stride < 3 * (int64_t)N || !(divisor > 0.0f))
        return fail(LG_ERR_INVALID_ARGUMENT, "lg_sh_grad_from_rgb: missing buffer, view_stride < 3 N or divisor <= 0");
    if (dL_dshs_rest && M == 1) dL_dshs_rest = nullptr;     // degree 0: nothing beyond the dc row
    hipStream_t stream = (hipStream_t)stream_p;
    lg_sh_grad_from_rgb_kernel<<<(N + LG_PP - 1) / LG_PP, LG_PP, 0, stream>>>(N, M, sh_degree, V, means3D, campos, drgb, (size_t)view_stride, divisor,
                                                                           accumulate ? 1 : 0, dL_dshs, dL_dshs_rest);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_sh_grad_from_rgb_kernel launch", e);
    return LG_OK;
}

extern "C" int lg_score_from_count(int32_t N, const int32_t* count, const float* weight, float* score, void* stream_p)
{
    if (N < 0 || (N > 0 && (!count || !score))) return fail(LG_ERR_INVALID_ARGUMENT, "bad arguments");
    if (N == 0) return LG_OK;
    hipStream_t stream = (hipStream_t)stream_p;
    lg_score_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, count, weight, score, nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_score_kernel launch", e);
    return LG_OK;
}

extern "C" size_t lg_prune_scratch_bytes(int32_t N)
{
    (void)N;
    return align_up(2 * sizeof(LgSelect));
}

extern "C" int lg_prune_epilogue(int32_t N, const float* scaling, const float* imp_list, float v_pow, double prune_percent,
                                 float* v_list, uint8_t* mask, float* thresholds, void* scratch, uint32_t flags, void* stream_p)
{
    if (N <= 0) return fail(LG_ERR_INVALID_ARGUMENT, "prune epilogue needs N >= 1 (the reference indexes an empty sort)");
    if (!scaling || !imp_list || !v_list || !mask || !thresholds || !scratch) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    if (!(prune_percent >= 0.0 && prune_percent <= 1.0)) return fail(LG_ERR_INVALID_ARGUMENT, "prune_percent must be in [0, 1]");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = flags & LG_FLAG_DEBUG, prof = flags & LG_FLAG_PROFILE;
    LgSelect* st = (LgSelect*)scratch;
    // prune.py:122-124: element `index = int(N * 0.9)` of the DESCENDING sort = ascending rank N - 1 - index
    const int index = (int)((double)N * 0.9);
    const uint32_t rank_volume = (uint32_t)(N - 1 - (index < N ? index : N - 1));
    // scene/gaussian_model.py:778-779: ascending index int(percent * (N - 1)), evaluated like Python (double)
    const uint32_t rank_score = (uint32_t)(prune_percent * (double)(N - 1));
    const int blocks = (int)std::min<int64_t>(((int64_t)N + 255) / 256, 2048);
    ProfScope ps(prof, "prune_epilogue", stream);
    HIP_TRY(hipMemsetAsync(st, 0, 2 * sizeof(LgSelect), stream));
    for (int p = 0; p < 4; p++) {
        lg_select_pass<0><<<blocks, 256, 0, stream>>>(N, p, rank_volume, scaling, &st[0]);
        KCHECK("lg_select_pass<volume>");
    }
    lg_v_imp_score_kernel<<<blocks, 256, 0, stream>>>(N, rank_volume, scaling, imp_list, v_pow, &st[0], v_list, &st[1], thresholds);
    KCHECK("lg_v_imp_score_kernel");
    for (int p = 1; p < 4; p++) {
        lg_select_pass<1><<<blocks, 256, 0, stream>>>(N, p, rank_score, v_list, &st[1]);
        KCHECK("lg_select_pass<score>");
    }
    lg_prune_mask_kernel<<<blocks, 256, 0, stream>>>(N, rank_score, v_list, &st[1], mask, thresholds);
    KCHECK("lg_prune_mask_kernel");
    return LG_OK;
}

// rank-th smallest element of values[0..N) by one radix select (four 8-bit histogram passes; exact: the element a sort puts
// at that index) and, when mask != NULL, mask[i] = values[i] <= that element.  The two halves of the prune epilogue around the
// reference's own torch.pow (prune.prune_epilogue): no sort, no host read-back.
extern "C" int lg_select_mask(int32_t N, const float* values, int64_t rank, uint8_t* mask, float* out_value, void* scratch, void* stream_p)
{
    if (N <= 0 || rank < 0 || rank >= N) return fail(LG_ERR_INVALID_ARGUMENT, "lg_select_mask needs N >= 1 and 0 <= rank < N");
    if (!values || !out_value || !scratch) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = false;
    LgSelect* st = (LgSelect*)scratch;
    const int blocks = (int)std::min<int64_t>(((int64_t)N + 255) / 256, 2048);
    HIP_TRY(hipMemsetAsync(st, 0, sizeof(LgSelect), stream));
    for (int p = 0; p < 4; p++) {
        lg_select_pass<1><<<blocks, 256, 0, stream>>>(N, p, (uint32_t)rank, values, st);
        KCHECK("lg_select_pass");
    }
    lg_select_finish_kernel<<<mask ? blocks : 1, 256, 0, stream>>>(N, (uint32_t)rank, values, st, mask, out_value);
    KCHECK("lg_select_finish_kernel");
    return LG_OK;
}

extern "C" int lg_ordered_sum(int32_t V, int64_t n, const float* rows, int64_t row_stride, float* out, void* stream_p)
{
    if (V <= 0 || n < 0 || row_stride < n) return fail(LG_ERR_INVALID_ARGUMENT, "bad shape");
    if (n == 0) return LG_OK;
    if (!rows || !out) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    lg_ordered_sum_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(V, (size_t)n, rows, (size_t)row_stride, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_ordered_sum_kernel launch", e);
    return LG_OK;
}

// ---- compaction of the Gaussian tensors after a prune (scene/gaussian_model.py:564-600) ----
extern "C" size_t lg_compact_scratch_bytes(int32_t N)
{
    const size_t nb = ((size_t)(N > 0 ? N : 1) + LG_COMPACT_ROWS - 1) / LG_COMPACT_ROWS;
    return 2 * align_up(nb * 4);
}

extern "C" int lg_compact_plan(int32_t N, const uint8_t* keep, int32_t* dest, int32_t* count, void* scratch, void* stream_p)
{
    if (N < 0) return fail(LG_ERR_INVALID_ARGUMENT, "bad row count");
    if (!count || (N > 0 && (!keep || !dest || !scratch))) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = false;
    if (N == 0) { HIP_TRY(hipMemsetAsync(count, 0, 4, stream)); return LG_OK; }
    const int nb = (N + LG_COMPACT_ROWS - 1) / LG_COMPACT_ROWS;
    uint32_t* blk_sum = (uint32_t*)scratch;
    uint32_t* blk_off = (uint32_t*)((char*)scratch + align_up((size_t)nb * 4));
    lg_compact_count<<<nb, 256, 0, stream>>>(N, keep, blk_sum);
    KCHECK("lg_compact_count");
    lg_scan_words<<<1, 1024, 0, stream>>>(nb, blk_sum, blk_off, count);
    KCHECK("lg_scan_words");
    lg_compact_dest<<<nb, 256, 0, stream>>>(N, keep, blk_off, dest);
    KCHECK("lg_compact_dest");
    return LG_OK;
}

extern "C" int lg_compact_rows(int32_t N, const int32_t* dest, int32_t num_tensors, const void* const* src, void* const* dst,
                               const int32_t* row_bytes, void* stream_p)
{
    if (N < 0 || num_tensors < 0 || num_tensors > LG_COMPACT_MAX_TENSORS) return fail(LG_ERR_INVALID_ARGUMENT, "bad tensor count (max 32 per call)");
    if (N == 0 || num_tensors == 0) return LG_OK;
    if (!dest || !src || !dst || !row_bytes) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = false;
    LgCompactArgs a;
    memset(&a, 0, sizeof(a));
    size_t widest = 1;
    for (int t = 0; t < num_tensors; t++) {
        if (row_bytes[t] <= 0 || (row_bytes[t] & 3) || !src[t] || !dst[t] || ((uintptr_t)src[t] & 3) || ((uintptr_t)dst[t] & 3))
            return fail(LG_ERR_INVALID_ARGUMENT, "lg_compact_rows: rows must be non-empty multiples of 4 bytes, 4-byte aligned");
        a.src[t] = (const uint32_t*)src[t]; a.dst[t] = (uint32_t*)dst[t]; a.words[t] = (uint32_t)(row_bytes[t] / 4);
        widest = std::max<size_t>(widest, a.words[t]);
    }
    const size_t blocks = std::min<size_t>(((size_t)N * widest + 1023) / 1024, 8192);
    lg_compact_move<<<dim3((unsigned)std::max<size_t>(blocks, 1), (unsigned)num_tensors), 256, 0, stream>>>(N, dest, a);
    KCHECK("lg_compact_move");
    return LG_OK;
}

// ---- VecTree nearest-code search (vectree/vq.py:262-266) ----
extern "C" size_t lg_vq_scratch_bytes(int32_t K, int32_t d)
{
    const int dk2 = lg_vq_dk2(d);
    if (K <= 0 || d <= 0 || dk2 == 0) return 0;
    return align_up((size_t)lg_vq_kpad(K) * 2 * dk2 * sizeof(float));
}

extern "C" int lg_vq_nearest(int32_t n, int32_t d, int32_t K, const float* x, const float* codebook, int32_t* out_index, void* scratch,
                             uint32_t flags, void* stream_p)
{
    const int dk2 = lg_vq_dk2(d);
    if (n < 0 || K <= 0 || d <= 0 || dk2 == 0) return fail(LG_ERR_INVALID_ARGUMENT, "lg_vq_nearest: need n >= 0, K >= 1, 1 <= d <= 63");
    if (n == 0) return LG_OK;
    if (!x || !codebook || !out_index || !scratch) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = flags & LG_FLAG_DEBUG, prof = flags & LG_FLAG_PROFILE;
    const int Kpad = lg_vq_kpad(K), dk = 2 * dk2;
    float* cbA = (float*)scratch;
    ProfScope ps(prof, "vq_nearest", stream);
    lg_vq_prepare<<<(Kpad + 255) / 256, 256, 0, stream>>>(K, Kpad, d, dk, codebook, cbA);
    KCHECK("lg_vq_prepare");
    const unsigned grid = (unsigned)((n + 127) / 128);
#define LAUNCH_VQ(D2) lg_vq_nearest_kernel<D2><<<grid, 256, 0, stream>>>(n, d, Kpad, x, cbA, out_index)
    switch (dk2) {
        case 2: LAUNCH_VQ(2); break;
        case 4: LAUNCH_VQ(4); break;
        case 7: LAUNCH_VQ(7); break;
        case 8: LAUNCH_VQ(8); break;
        case 14: LAUNCH_VQ(14); break;
        case 16: LAUNCH_VQ(16); break;
        case 25: LAUNCH_VQ(25); break;
        default: LAUNCH_VQ(32); break;
    }
#undef LAUNCH_VQ
    KCHECK("lg_vq_nearest_kernel");
    return LG_OK;
}

extern "C" size_t lg_knn_scratch_bytes(int32_t P) { return P < 0 ? 0 : carve_knn(nullptr, P).total; }

extern "C" int lg_knn3_mean_dist2(int32_t P, const float* points, float* mean_dist2, void* scratch, uint32_t flags, void* stream_p)
{
    if (P < 0) return fail(LG_ERR_INVALID_ARGUMENT, "bad point count");
    if (P == 0) return LG_OK;
    if (!points || !mean_dist2 || !scratch) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = flags & LG_FLAG_DEBUG, prof = flags & LG_FLAG_PROFILE;
    KnnView kv = carve_knn(scratch, P);
    ProfScope ps(prof, "knn3", stream);
    HIP_TRY(hipMemsetAsync(kv.box, 0xFF, 12, stream));                 // min keys
    HIP_TRY(hipMemsetAsync(kv.box + 3, 0, 64 - 12, stream));           // max keys, open-point counters
    const int nb = (P + 255) / 256;
    lg_knn_bbox<<<std::min(nb, 1024), 256, 0, stream>>>(P, points, kv.box);
    KCHECK("lg_knn_bbox");
    int key_bits = 1;
    while ((1u << key_bits) < kv.cap) key_bits++;
    for (int level = 0; level < LG_KNN_LEVELS; level++) {
        lg_knn_cells<<<nb, 256, 0, stream>>>(P, kv.cap, level, points, kv.box, kv.pairs_in);
        KCHECK("lg_knn_cells");
        size_t tb = kv.sort_temp_bytes;
        HIP_TRY(lg_sort_keys(kv.sort_temp, tb, kv.pairs_in, kv.pairs_out, (uint32_t)P, 32, 32 + key_bits, nullptr, false, stream));
        HIP_TRY(hipMemsetAsync(kv.cell_start, 0, (size_t)kv.cap * 4, stream));
        HIP_TRY(hipMemsetAsync(kv.cell_end, 0, (size_t)kv.cap * 4, stream));
        lg_knn_ranges<<<nb, 256, 0, stream>>>(P, points, kv.pairs_out, kv.cell_start, kv.cell_end, kv.sorted,
                                              (const uint32_t*)((char*)kv.sort_temp + lg_sort_layout((size_t)P).ticket_off) + 15, kv.box);
        KCHECK("lg_knn_ranges");
        uint32_t* open_in = (level & 1) ? kv.open_a : kv.open_b;
        uint32_t* open_out = (level & 1) ? kv.open_b : kv.open_a;
        const int max_rings = level == LG_KNN_LEVELS - 1 ? (1 << 30) : LG_KNN_RINGS;
        lg_knn_query<<<nb, 256, 0, stream>>>(P, kv.cap, level, max_rings, points, kv.box, kv.sorted, kv.cell_start, kv.cell_end, open_in,
                                             kv.box + 8 + (level > 0 ? level - 1 : 0), open_out, kv.box + 8 + level, mean_dist2);
        KCHECK("lg_knn_query");
    }
    // one 4-byte read-back at the end (distCUDA2 runs once per training run, scene/gaussian_model.py:152): did any level's sort give up?
    uint32_t h_err = 0;
    HIP_TRY(hipMemcpyAsync(&h_err, kv.box + 15, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (h_err) return fail(LG_ERR_DEVICE, "lg_knn3_mean_dist2: the radix sort of a grid level gave up (look-back poll budget exhausted): the distances are void");
    return LG_OK;
}

extern "C" size_t lg_loss_state_bytes(int32_t C, int32_t H, int32_t W)
{
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    return carve_loss(nullptr, C, H, W).total;
}

extern "C" int lg_loss_forward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, void* state, float* out_l1_ssim,
                               uint32_t flags, void* stream_p)
{
    if (C <= 0 || H <= 0 || W <= 0 || C > 65535) return fail(LG_ERR_INVALID_ARGUMENT, "bad image shape");
    if (!img || !gt || !state || !out_l1_ssim) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = flags & LG_FLAG_DEBUG, prof = flags & LG_FLAG_PROFILE;
    LossView lv = carve_loss(state, C, H, W);
    dim3 grid(lg_loss_strips(W), lg_loss_segs(H), C);      // one wave per (strip of 64 columns, LG_LOSS_TH rows, plane)
    if (grid.y > 65535) return fail(LG_ERR_INVALID_ARGUMENT, "image too large");
    if (flags & LG_FLAG_L1_ONLY) {
        ProfScope ps(prof, "l1_fwd", stream);
        const size_t n = (size_t)C * H * W;
        const int blocks = (int)std::min<size_t>((n + 1023) / 1024, (size_t)grid.x * grid.y * grid.z);   // partials has one slot per tile
        lg_l1_fwd<<<blocks, 256, 0, stream>>>(n, img, gt, lv.partials);
        KCHECK("lg_l1_fwd");
        lg_loss_finalize<<<1, 256, 0, stream>>>(blocks, 1.0 / (double)n, lv.partials, out_l1_ssim);
        KCHECK("lg_loss_finalize");
        return LG_OK;
    }
    {
        ProfScope ps(prof, "loss_fwd", stream);
        lg_loss_fwd<<<grid, LG_LOSS_STRIP, 0, stream>>>(H, W, img, gt, lv.dmu1, lv.dsig1, lv.dsig12, lv.partials);
        KCHECK("lg_loss_fwd");
        lg_loss_finalize<<<1, 256, 0, stream>>>((int)(grid.x * grid.y * grid.z), 1.0 / ((double)C * H * W), lv.partials, out_l1_ssim);
        KCHECK("lg_loss_finalize");
    }
    return LG_OK;
}

extern "C" int lg_loss_backward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, const void* state,
                                const float* dL_dl1, float scale_l1, const float* dL_dssim, float scale_ssim, float* dL_dimg,
                                uint32_t flags, void* stream_p)
{
    if (C <= 0 || H <= 0 || W <= 0 || C > 65535) return fail(LG_ERR_INVALID_ARGUMENT, "bad image shape");
    if (!img || !gt || !state || !dL_dimg) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = flags & LG_FLAG_DEBUG, prof = flags & LG_FLAG_PROFILE;
    LossView lv = carve_loss(const_cast<void*>(state), C, H, W);
    dim3 grid(lg_loss_strips(W), lg_loss_segs(H), C);
    if (grid.y > 65535) return fail(LG_ERR_INVALID_ARGUMENT, "image too large");
    if (flags & LG_FLAG_L1_ONLY) {
        ProfScope ps(prof, "l1_bwd", stream);
        const size_t n = (size_t)C * H * W;
        lg_l1_bwd<<<(int)std::min<size_t>((n + 1023) / 1024, 65535 * 16), 256, 0, stream>>>(n, img, gt, dL_dl1, scale_l1 / (float)n, dL_dimg);
        KCHECK("lg_l1_bwd");
        return LG_OK;
    }
    {
        ProfScope ps(prof, "loss_bwd", stream);
        lg_loss_bwd<<<grid, LG_LOSS_STRIP, 0, stream>>>(H, W, img, gt, lv.dmu1, lv.dsig1, lv.dsig12, dL_dl1, scale_l1, dL_dssim, scale_ssim,
                                              (float)(1.0 / ((double)C * H * W)), dL_dimg);
        KCHECK("lg_loss_bwd");
    }
    return LG_OK;
}

// diagnostics: Gaussian id of the last contributor of every pixel (0xFFFFFFFF: none) from the state a forward saved -- the
// implementation-independent form of n_contrib (which is a position in THIS library's culled tile lists)
__global__ void __launch_bounds__(256)
lg_debug_last_contributor_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint64_t* __restrict__ entries, uint32_t gid_mask,
                                 const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ counters, uint32_t* __restrict__ out)
{
    const size_t pid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pid >= (size_t)W * H) return;
    const int x = (int)(pid % (size_t)W), y = (int)(pid / (size_t)W);
    const uint32_t n = counters[0] == 0u ? n_contrib[pid] : 0u;
    out[pid] = n > 0u ? ((uint32_t)entries[ranges[(y / LG_TILE) * gx + x / LG_TILE].x + n - 1u] & gid_mask) : 0xFFFFFFFFu;
}
extern "C" int lg_debug_tile_lists(const lg_view* v, const void* bin_p, int64_t R, uint32_t* out_ranges, uint64_t* out_entries, void* stream_p)
{
    if (!v || !bin_p || !out_ranges || !out_entries || R < 0 || v->image_width <= 0 || v->image_height <= 0)
        return fail(LG_ERR_INVALID_ARGUMENT, "lg_debug_tile_lists: missing buffer");
    const int W = v->image_width, H = v->image_height;
    const size_t ntiles = (size_t)((W + LG_TILE - 1) / LG_TILE) * ((H + LG_TILE - 1) / LG_TILE);
    BinView bin = carve_bin(const_cast<void*>(bin_p), R, W, H, lg_segment_of(v));
    hipError_t e = hipMemcpyAsync(out_ranges, bin.ranges, ntiles * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream_p);
    if (e == hipSuccess && R > 0) e = hipMemcpyAsync(out_entries, bin.entries, (size_t)R * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream_p);
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_debug_tile_lists copy", e);
    return LG_OK;
}

extern "C" int lg_debug_view_meta(const lg_view* v, const void* bin_p, int64_t R, uint32_t* out_meta16, void* stream_p)
{
    if (!v || !bin_p || !out_meta16 || R < 0 || v->image_width <= 0 || v->image_height <= 0) return fail(LG_ERR_INVALID_ARGUMENT, "lg_debug_view_meta: missing buffer");
    BinView bin = carve_bin(const_cast<void*>(bin_p), R, v->image_width, v->image_height, lg_segment_of(v));
    hipError_t e = hipMemcpyAsync(out_meta16, bin.meta, 64, hipMemcpyDeviceToDevice, (hipStream_t)stream_p);
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_debug_view_meta copy", e);
    return LG_OK;
}

extern "C" int lg_debug_last_contributor(const lg_view* v, int32_t N, const void* geom_p, const void* bin_p, const void* img_p, int64_t R,
                                         uint32_t* out_ids, void* stream_p)
{
    if (!v || N <= 0 || !geom_p || !bin_p || !img_p || !out_ids || v->image_width <= 0 || v->image_height <= 0)
        return fail(LG_ERR_INVALID_ARGUMENT, "lg_debug_last_contributor: missing buffer");
    const int W = v->image_width, H = v->image_height, gx = (W + LG_TILE - 1) / LG_TILE;
    GeomView geo = carve_geom(const_cast<void*>(geom_p), N);
    ImgView img = carve_img(const_cast<void*>(img_p), W, H);
    BinView bin = carve_bin(const_cast<void*>(bin_p), R, W, H, lg_segment_of(v));
    const int gid_bits = bits_for((uint32_t)(N > 1 ? N : 2));
    const uint32_t gid_mask = gid_bits >= 32 ? 0xFFFFFFFFu : ((1u << gid_bits) - 1u);
    const size_t P = (size_t)W * H;
    lg_debug_last_contributor_kernel<<<(unsigned)((P + 255) / 256), 256, 0, (hipStream_t)stream_p>>>(W, H, gx, bin.ranges, bin.entries, gid_mask, img.n_contrib,
                                                                                                  geo.counters, out_ids);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_debug_last_contributor launch", e);
    return LG_OK;
}

// diagnostics: the K4 radix sort on its own (stand-alone histogram pass + onesweep passes)
extern "C" size_t lg_debug_sort_temp_bytes(int64_t n) { return n < 0 ? 0 : lg_sort_layout((size_t)n).total; }
extern "C" int lg_debug_sort_keys(int64_t n, const uint64_t* keys_in, uint64_t* keys_out, int32_t begin_bit, int32_t end_bit, void* temp,
                                  void* stream_p)
{
    if (n < 0 || n >= (1ll << 30) || begin_bit < 0 || end_bit > 64 || end_bit <= begin_bit) return fail(LG_ERR_INVALID_ARGUMENT, "bad sort arguments");
    if (n == 0) return LG_OK;
    if (!keys_in || !keys_out || !temp) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    const LgSortLayout L = lg_sort_layout((size_t)n);
    size_t tb = L.total;
    HIP_TRY(lg_sort_keys(temp, tb, keys_in, keys_out, (uint32_t)n, begin_bit, end_bit, nullptr, false, (hipStream_t)stream_p));
    uint32_t h_err = 0;
    HIP_TRY(hipMemcpyAsync(&h_err, (char*)temp + L.ticket_off + 15 * 4, 4, hipMemcpyDeviceToHost, (hipStream_t)stream_p));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_p));
    if (h_err & LG_ABORT_SORT) return fail(LG_ERR_DEVICE, "radix sort look-back gave up (a predecessor tile never published)");
    return LG_OK;
}

// diagnostics: the failure path of the look-back.  One digit pass is launched with its ticket counter preset to 1, so the tile
// that runs has a predecessor (tile 0) that does not exist and never publishes; with a small poll budget the look-back must give
// up, set the error word and return -- LG_ERR_DEVICE here -- instead of hanging the device or passing a wrong order on silently.
extern "C" int lg_debug_sort_orphan(int64_t n, const uint64_t* keys_in, uint64_t* keys_out, void* temp, void* stream_p)
{
    if (n <= 0 || n > LG_SORT_TILE) return fail(LG_ERR_INVALID_ARGUMENT, "lg_debug_sort_orphan: 1 <= n <= one sort tile");
    if (!keys_in || !keys_out || !temp) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    hipStream_t stream = (hipStream_t)stream_p;
    const LgSortLayout L = lg_sort_layout((size_t)2 * LG_SORT_TILE);
    char* base = (char*)temp;
    HIP_TRY(lg_zero_async(base, lg_sort_clear_bytes(L, 1), stream));
    const uint32_t one = 1u;
    HIP_TRY(hipMemcpyAsync(base + L.ticket_off, &one, 4, hipMemcpyHostToDevice, stream));
    uint32_t* tickets = (uint32_t*)(base + L.ticket_off);
    // n_arg = one tile beyond the orphan so that tile 1 is inside the key range; it sorts keys_in[0..n) as its own keys
    lg_onesweep_pass<<<1, LG_SORT_BLOCK, 0, stream>>>(keys_in - LG_SORT_TILE, keys_out - LG_SORT_TILE, nullptr, (uint32_t)(LG_SORT_TILE + n), 0, 8,
                                                       (uint32_t*)(base + L.hist_off), tickets, (uint32_t*)(base + L.state_off), tickets + 15, 64u, nullptr, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_onesweep_pass launch", e);
    uint32_t h_err = 0;
    HIP_TRY(hipMemcpyAsync(&h_err, tickets + 15, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (h_err & LG_ABORT_SORT) return fail(LG_ERR_DEVICE, "radix sort look-back gave up (a predecessor tile never published)");
    return LG_OK;
}

// The four status words of a view as they stand when `stream` reaches this call: { abort flags, prefiltered violation,
// largest depth bit pattern, instance count }.  One blocking 16-byte read.  abort bit 2 (LG_ABORT_SORT) can only be set
// after the words lg_forward_bounded hands out were written, so a caller that must know reads them here.
extern "C" int lg_view_status(const void* geom, int32_t N, uint32_t* out4, void* stream_p)
{
    if (!geom || !out4 || N < 0) return fail(LG_ERR_INVALID_ARGUMENT, "lg_view_status: missing buffer");
    GeomView geo = carve_geom(const_cast<void*>(geom), N);
    hipStream_t stream = (hipStream_t)stream_p;
    HIP_TRY(hipMemcpyAsync(out4, geo.counters, 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (out4[0] & LG_ABORT_SORT) return fail(LG_ERR_DEVICE, "radix sort look-back gave up (a predecessor tile never published): the view is void");
    return LG_OK;
}

// diagnostics: the activations of the fused-getter path (K1 / K9, RAW) on their own, in several candidate operation orders,
// to be compared bit for bit with torch.exp / F.normalize / torch.sigmoid (tools/activation_probe.py)
__global__ void lg_debug_activations_kernel(int n, const float* __restrict__ s, const float* __restrict__ r, const float* __restrict__ o,
                                            float* __restrict__ out_s, float* __restrict__ out_r, float* __restrict__ out_o)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_s[i] = expf(s[i]);
    out_o[i] = lg_sigmoid(o[i]);
    out_o[n + i] = 1.0f / (1.0f + __expf(-o[i]));
    const float a = r[4 * i], b = r[4 * i + 1], c = r[4 * i + 2], d = r[4 * i + 3];
    const float n0 = fmaxf(sqrtf(((a * a + b * b) + c * c) + d * d), 1e-12f);
    const float n1 = fmaxf(sqrtf((a * a + b * b) + (c * c + d * d)), 1e-12f);
    const float n2 = fmaxf(sqrtf(fmaf(d, d, fmaf(c, c, fmaf(b, b, a * a)))), 1e-12f);
    const float n3 = fmaxf(sqrtf((a * a + c * c) + (b * b + d * d)), 1e-12f);
    const float nn[4] = {n0, n1, n2, n3};
    for (int v = 0; v < 4; v++) {
        out_r[(size_t)v * 4 * n + 4 * i] = a / nn[v]; out_r[(size_t)v * 4 * n + 4 * i + 1] = b / nn[v];
        out_r[(size_t)v * 4 * n + 4 * i + 2] = c / nn[v]; out_r[(size_t)v * 4 * n + 4 * i + 3] = d / nn[v];
    }
}
extern "C" int lg_debug_activations(int32_t n, const float* s, const float* r, const float* o, float* out_s, float* out_r, float* out_o, void* stream_p)
{
    if (n <= 0) return LG_OK;
    lg_debug_activations_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_p>>>(n, s, r, o, out_s, out_r, out_o);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_debug_activations launch", e);
    return LG_OK;
}

extern "C" int lg_debug_reduce9(const float* in_64x9, float* out_9, void* stream_p)
{
    lg_debug_reduce9_kernel<<<1, 64, 0, (hipStream_t)stream_p>>>(in_64x9, out_9);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_debug_reduce9 launch", e);
    return LG_OK;
}

#ifndef LG_BUILD_ID
#define LG_BUILD_ID "unknown"
#endif
extern "C" const char* lg_build_id(void) { return LG_BUILD_ID; }
extern "C" int lg_abi_version(void) { return LG_ABI_VERSION; }
extern "C" const char* lg_last_error(void) { return g_err.c_str(); }
extern "C" int lg_last_stats(lg_stats* out)
{
    if (!out) return LG_ERR_INVALID_ARGUMENT;
    *out = g_stats;
    return LG_OK;
}

extern "C" void lg_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_prof)
        for (auto& ev : p.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    g_prof.clear();
}

extern "C" int lg_profile_read(lg_kernel_time* out, int cap)
{
    int n = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_prof) {
        for (auto& ev : p.pending) {
            float ms = 0.0f;
            if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                p.ms += ms; p.n += 1;
            }
            (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
        }
        p.pending.clear();
        if (out && n < cap) {
            memset(&out[n], 0, sizeof(lg_kernel_time));
            strncpy(out[n].name, p.name.c_str(), sizeof(out[n].name) - 1);
            out[n].total_ms = p.ms; out[n].launches = p.n;
        }
        n++;
    }
    return n;
}

