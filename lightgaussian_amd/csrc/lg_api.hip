// lg_api.hip -- C ABI (include/lightgaussian.h) of the gfx950 (CDNA4, wave64) LightGaussian rasterizer.
// The single translation unit of liblightgaussian_hip.so; the kernels live in the headers it includes:
//   lg_math.h        scalar float arithmetic shared with the CPU test harness (canonical operation order)
//   lg_host.h        error strings, optional hipEvent profiler, scratch carving (GeomView / ImgView / BinView)
//   lg_wave.h        wave64 primitives (DPP / permlane reductions)
//   lg_preprocess.h  K1 lg_preprocess<RAW>, K8+K9 lg_preprocess_bwd<RAW>            (per Gaussian, HBM-bound)
//   lg_binning.h     lg_reduce_dmax, K3 lg_duplicate<PACKED>, K5 lg_tile_ranges, lg_tile_order (per instance; K2/K4 = rocPRIM scan / radix sort)
//   lg_loss.h        lg_loss_fwd / lg_loss_bwd: fused L1 + SSIM of the training step             (per 32x32 tile, LDS-tiled)
//   lg_prune.h       lg_select_pass, lg_v_imp_score_kernel, lg_prune_mask_kernel: device-resident prune epilogue (radix selects)
//   lg_knn.h         distCUDA2 (simple-knn): exact 3-nearest-neighbour mean squared distance on a multi-level uniform grid
//   lg_blend.h       K6 lg_blend_fwd<COUNT,FSCORE,EXACT>, lg_score_kernel, K7 lg_blend_bwd<EXACT>   (per tile, VALU-bound)
//
// Pipeline of one view:
//   K1 project + EWA + SH->RGB + exact footprint culling  ->  K2 scan of instance counts, blocking read of R
//   K3 packed keys tile|depth|id  ->  K4 stable keys-only radix sort (lowest depth bits skipped when that saves a pass)
//   K5 tile ranges + completion of the skipped bits; the sorted keys ARE the per-tile lists (no id / slot arrays)
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

// ------------------------------------------------------------------------------------------------
// host side
static int check_args(const lg_view* v, const lg_gaussians* g)
{
    if (!v || !g) return fail(LG_ERR_INVALID_ARGUMENT, "null view/gaussians");
    if (g->N < 0 || v->image_width <= 0 || v->image_height <= 0) return fail(LG_ERR_INVALID_ARGUMENT, "bad sizes");
    if (g->N == 0) return LG_OK; // nothing to validate against: empty tensors carry no pointers
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
static std::vector<uint32_t*> g_pin_free;
struct PinnedSlot {
    uint32_t* p = nullptr;
    PinnedSlot()
    {
        {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            if (!g_pin_free.empty()) { p = g_pin_free.back(); g_pin_free.pop_back(); }
        }
        if (!p && hipHostMalloc((void**)&p, 64, hipHostMallocDefault) != hipSuccess) p = nullptr;
    }
    ~PinnedSlot()
    {
        if (p) { std::lock_guard<std::mutex> lk(g_pin_mu); g_pin_free.push_back(p); }
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

static int forward_impl(const lg_view* v, const lg_gaussians* g, void* geom_p, void* img_p, lg_alloc_fn alloc, void* alloc_user,
                        int weight_policy, float* out_color, int32_t* out_radii, int32_t* out_count, float* out_score,
                        void** binning_out, int64_t* num_rendered, void* stream_p)
{
    int rc = check_args(v, g);
    if (rc != LG_OK) return rc;
    if (!geom_p || !img_p || !out_color || (!out_radii && g->N > 0) || !alloc) return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    const bool count = out_count != nullptr;
    if (count && !out_score) return fail(LG_ERR_INVALID_ARGUMENT, "count needs score");
    if (count && (weight_policy < 0 || weight_policy > 3)) return fail(LG_ERR_INVALID_ARGUMENT, "bad weight policy");
    hipStream_t stream = (hipStream_t)stream_p;
    const bool debug = v->flags & LG_FLAG_DEBUG, prof = v->flags & LG_FLAG_PROFILE, fast = v->flags & LG_FLAG_FAST_EXP;
    const int N = g->N, W = v->image_width, H = v->image_height;
    const int gx = (W + LG_TILE - 1) / LG_TILE, gy = (H + LG_TILE - 1) / LG_TILE, ntiles = gx * gy;
    const int ntiles_pad = (ntiles + LG_TILE_GRID_ALIGN - 1) / LG_TILE_GRID_ALIGN * LG_TILE_GRID_ALIGN; // grid of the per-tile kernels
    GeomView geo = carve_geom(geom_p, N);
    ImgView img = carve_img(img_p, W, H);
    if (binning_out) *binning_out = nullptr;
    if (num_rendered) *num_rendered = 0;
    uint32_t h_counters[3] = {0, 0, 0}, h_R = 0;
    if (N > 0) {
        {
            ProfScope ps(prof, "preprocess", stream);
#define LAUNCH_PP(RAWP, DIR)                                                                                                         \
    lg_preprocess<RAWP, DIR><<<(N + LG_PP - 1) / LG_PP, LG_PP, 0, stream>>>(N, g->M, v->sh_degree, W, H, v->tanfovx, v->tanfovy,      \
                                                                      v->scale_modifier, v->prefiltered, (v->flags & LG_FLAG_SKIP_COLOR) ? 1 : 0, v->viewmatrix, v->projmatrix, \
                                                                      v->campos, g->means3D, g->shs, g->shs_rest, g->colors_precomp,   \
                                                                      g->opacities, g->scales, g->rotations, g->cov3D_precomp, geo, out_radii, out_count, out_score)
            // SH rows are read directly by their lanes (dword-aligned dwordx4 loads); LG_K1_LDS=1 selects the LDS-staged reads
            const bool direct = getenv("LG_K1_LDS") == nullptr;
            const bool raw = v->flags & LG_FLAG_RAW_PARAMS;
            if (raw && direct) LAUNCH_PP(true, true);
            else if (raw) LAUNCH_PP(true, false);
            else if (direct) LAUNCH_PP(false, true);
            else LAUNCH_PP(false, false);
#undef LAUNCH_PP
        }
        KCHECK("lg_preprocess");
        {
            ProfScope ps(prof, "scan", stream);
            size_t tb = geo.scan_temp_bytes;
            HIP_TRY(hipcub::DeviceScan::InclusiveSum(geo.scan_temp, tb, geo.touched, geo.offsets, N, stream));
        }
        // The forward has ONE blocking read-back: the instance count R (it sizes the binning buffers), with the depth
        // maximum and the prefiltered flag riding along: lg_reduce_dmax gathers them into counters[0..3] and a single
        // 16-byte copy into pinned host memory fetches them.  (Two pageable copies, the first version, cost two host round
        // trips: ~100 us of idle GPU per view in the kernel trace, now ~70.  Polling a host-mapped mailbox written by the
        // kernel instead of hipStreamSynchronize was measured as well: no difference, so the plain form stays.)
        lg_reduce_dmax<<<1, 1024, 0, stream>>>((N + LG_PP - 1) / LG_PP, geo.blk_dmax, geo.offsets + (N - 1), geo.counters);
        KCHECK("lg_reduce_dmax");
        PinnedSlot slot;                                      // process-wide pool: host threads come and go (views in flight)
        if (!slot.p) return fail(LG_ERR_ALLOC, "hipHostMalloc of the read-back slot failed");
        HIP_TRY(hipMemcpyAsync(slot.p, geo.counters, 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        h_counters[0] = slot.p[0]; h_counters[1] = slot.p[1]; h_counters[2] = slot.p[2];
        h_R = slot.p[3];
        if (v->prefiltered && h_counters[1]) return fail(LG_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
    }
    const int64_t R = h_R;
    if (R > 0x7FFFFFFFll) return fail(LG_ERR_INVALID_ARGUMENT, "more than 2^31-1 tile instances in one view");
    g_stats.num_rendered = R;
    g_stats.num_visible = -1; // not tracked on the device (see lg_preprocess); callers count radii > 0
    if (num_rendered) *num_rendered = R;

    // key format: packed single-u64 keys when tile | depth | id fit 64 bits (they do for every BASELINE config)
    const int tile_bits = bits_for((uint32_t)ntiles), gid_bits = bits_for((uint32_t)(N > 1 ? N : 2));
    const uint32_t dspan = h_counters[2] > LG_DEPTH_BIAS ? h_counters[2] - LG_DEPTH_BIAS : 0u;
    const int depth_bits = bits_for(dspan + 1u) > 0 ? bits_for(dspan + 1u) : 1;
    const bool packed = (tile_bits + depth_bits + gid_bits <= 64) && (getenv("LG_FORCE_PAIR_SORT") == nullptr);
    // the radix sort works in 8-bit passes: when the tile + depth field is a few bits over a multiple of 8, those lowest
    // depth bits are left to lg_tile_ranges (runs of equal sorted bits are finished by insertion) and a whole pass is saved
    int drop = (tile_bits + depth_bits) % 8;
    if (!packed || depth_bits - drop < 12 || getenv("LG_SORT_ALL_BITS") != nullptr) drop = 0;

    void* bin_p = alloc(alloc_user, carve_bin(nullptr, R, W, H, packed).total);
    if (!bin_p) return fail(LG_ERR_ALLOC, "binning allocator returned NULL");
    if (binning_out) *binning_out = bin_p;
    BinView bin = carve_bin(bin_p, R, W, H, packed);
    if (R == 0) HIP_TRY(hipMemsetAsync(bin.ranges, 0, (size_t)ntiles * 8, stream)); // otherwise cleared by lg_duplicate
    const uint32_t gid_mask = gid_bits >= 32 ? 0xFFFFFFFFu : ((1u << gid_bits) - 1u);
    if (R > 0) {
        {
            ProfScope ps(prof, "duplicate", stream);
            if (packed)
                lg_duplicate<true><<<(N + 255) / 256, 256, 0, stream>>>(N, gx, depth_bits, gid_bits, geo.touched, geo.offsets, geo.tinfo,
                                                                        bin.keys_in, nullptr, ntiles, bin.ranges);
            else
                lg_duplicate<false><<<(N + 255) / 256, 256, 0, stream>>>(N, gx, 0, 0, geo.touched, geo.offsets, geo.tinfo, bin.keys_in,
                                                                         bin.vals_in, ntiles, bin.ranges);
        }
        KCHECK("lg_duplicate");
        {
            ProfScope ps(prof, "sort", stream);
            size_t tb = bin.sort_temp_bytes;
            if (packed)
                HIP_TRY(lg_sort_keys(bin.sort_temp, tb, bin.keys_in, bin.entries, (unsigned)R, (unsigned)(gid_bits + drop),
                                     (unsigned)(gid_bits + depth_bits + tile_bits), stream));
            else
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(bin.sort_temp, tb, bin.keys_in, bin.keys_tmp, bin.vals_in, bin.vals_out, (int)R, 0,
                                                           32 + tile_bits, stream));
        }
        {
            ProfScope ps(prof, "tile_ranges", stream);
            if (packed)
                lg_tile_ranges<true><<<(uint32_t)((R + 255) / 256), 256, 0, stream>>>((uint32_t)R, depth_bits + gid_bits, gid_bits, drop,
                                                                                      bin.entries, nullptr, bin.entries, bin.keys_in, bin.ranges);
            else
                lg_tile_ranges<false><<<(uint32_t)((R + 255) / 256), 256, 0, stream>>>((uint32_t)R, 32, 0, 0, bin.keys_tmp, bin.vals_out,
                                                                                       bin.entries, nullptr, bin.ranges);
        }
        KCHECK("lg_tile_ranges");
    }
    // (count / score accumulators of the count variant were cleared by lg_preprocess)
    {
        ProfScope ps(prof, count ? "blend_fwd_count" : "blend_fwd", stream);
        dim3 grid(ntiles_pad), block(256);
#define LAUNCH_FWD(CNT, FS, EX)                                                                                                      \
    lg_blend_fwd<CNT, FS, EX><<<grid, block, 0, stream>>>(W, H, gx, ntiles, ntiles_pad, bin.ranges, bin.entries, gid_mask, geo.rec, v->bg,     \
                                                         out_color, img.final_T, img.n_contrib, out_count, out_score, weight_policy)
        const bool fs = count && (weight_policy == LG_WEIGHT_ALPHA || weight_policy == LG_WEIGHT_ALPHA_T);
        const bool nocolor = count && !fast && (v->flags & LG_FLAG_SKIP_COLOR);   // significance-only pass: no colour, no per-pixel outputs
        if (!count) { if (fast) LAUNCH_FWD(false, false, false); else LAUNCH_FWD(false, false, true); }
        else if (nocolor) {
            if (fs) lg_blend_fwd<true, true, true, false><<<grid, block, 0, stream>>>(W, H, gx, ntiles, ntiles_pad, bin.ranges, bin.entries, gid_mask, geo.rec, v->bg, out_color, img.final_T, img.n_contrib, out_count, out_score, weight_policy);
            else lg_blend_fwd<true, false, true, false><<<grid, block, 0, stream>>>(W, H, gx, ntiles, ntiles_pad, bin.ranges, bin.entries, gid_mask, geo.rec, v->bg, out_color, img.final_T, img.n_contrib, out_count, out_score, weight_policy);
        }
        else if (!fs) { if (fast) LAUNCH_FWD(true, false, false); else LAUNCH_FWD(true, false, true); }
        else { if (fast) LAUNCH_FWD(true, true, false); else LAUNCH_FWD(true, true, true); }
#undef LAUNCH_FWD
    }
    KCHECK("lg_blend_fwd");
    if (count && N > 0 && (weight_policy == LG_WEIGHT_ONE || weight_policy == LG_WEIGHT_OPACITY)) {
        ProfScope ps(prof, "score", stream);
        lg_score_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, out_count, weight_policy == LG_WEIGHT_OPACITY ? g->opacities : nullptr, out_score);
        KCHECK("lg_score_kernel");
    }
    return LG_OK;
}

extern "C" int lg_forward(const lg_view* view, const lg_gaussians* g, void* geom, void* img, lg_alloc_fn alloc, void* alloc_user,
                          float* out_color, int32_t* out_radii, void** binning_out, int64_t* num_rendered, void* stream)
{
    return forward_impl(view, g, geom, img, alloc, alloc_user, LG_WEIGHT_OPACITY, out_color, out_radii, nullptr, nullptr, binning_out,
                        num_rendered, stream);
}

extern "C" int lg_forward_count(const lg_view* view, const lg_gaussians* g, void* geom, void* img, lg_alloc_fn alloc, void* alloc_user,
                                int32_t weight_policy, float* out_color, int32_t* out_radii, int32_t* out_count, float* out_score,
                                void** binning_out, int64_t* num_rendered, void* stream)
{
    if (g && g->N > 0 && (!out_count || !out_score)) return fail(LG_ERR_INVALID_ARGUMENT, "count/score outputs required");
    return forward_impl(view, g, geom, img, alloc, alloc_user, weight_policy, out_color, out_radii, out_count, out_score, binning_out,
                        num_rendered, stream);
}

extern "C" int lg_backward(const lg_view* v, const lg_gaussians* g, const int32_t* radii, const void* geom_p, const void* bin_p,
                           const void* img_p, int64_t R, const float* dL_dcolor, float* dL_dmeans2D, float* dL_dmeans3D,
                           float* dL_dshs, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                           float* dL_dcov3D, float* dL_dshs_rest, void* scratch, void* stream_p)
{
    int rc = check_args(v, g);
    if (rc != LG_OK) return rc;
    if (g->shs_rest && !dL_dshs_rest) return fail(LG_ERR_INVALID_ARGUMENT, "missing gradient output for shs_rest");
    if (!radii || !geom_p || !bin_p || !img_p || !dL_dcolor || !dL_dmeans2D || !dL_dmeans3D || !dL_dopacity || !scratch)
        return fail(LG_ERR_INVALID_ARGUMENT, "missing buffer");
    if ((g->shs && !dL_dshs) || (g->colors_precomp && !dL_dcolors) || (g->scales && (!dL_dscales || !dL_drotations)) ||
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
    BinView bin = carve_bin(const_cast<void*>(bin_p), R, W, H, true); // only the format-independent prefix is used
    const int gid_bits = bits_for((uint32_t)(N > 1 ? N : 2));          // same field width as the forward used
    const uint32_t gid_mask = gid_bits >= 32 ? 0xFFFFFFFFu : ((1u << gid_bits) - 1u);
    float* rows = (float*)scratch; // [R][12] gradient rows, every row written by lg_blend_bwd
    if (R > 0) {
        // dispatch order of the per-tile backward (longest lists first).  Computed here, not in the forward, so that
        // forward-only renders and the significance pass do not pay for it; it lands in the array reserved for it inside
        // the binning buffer (the one write the backward makes to saved state; idempotent).
        ProfScope ps(prof, "tile_order", stream);
        lg_tile_order<<<1, 1024, 0, stream>>>(ntiles, bin.ranges, bin.tile_order);
    }
    KCHECK("lg_tile_order");
    if (R > 0) {
        ProfScope ps(prof, "blend_bwd", stream);
        if (fast)
            lg_blend_bwd<false><<<ntiles, 64, 0, stream>>>(W, H, gx, ntiles, bin.tile_order, bin.ranges, bin.entries, gid_mask, geo.tinfo, geo.rec, v->bg,
                                                                 img.final_T, img.n_contrib, dL_dcolor, rows);
        else
            lg_blend_bwd<true><<<ntiles, 64, 0, stream>>>(W, H, gx, ntiles, bin.tile_order, bin.ranges, bin.entries, gid_mask, geo.tinfo, geo.rec, v->bg,
                                                                img.final_T, img.n_contrib, dL_dcolor, rows);
    }
    KCHECK("lg_blend_bwd");
    {
        ProfScope ps(prof, "preprocess_bwd", stream);
#define LAUNCH_PPB(RAWP)                                                                                                             \
    lg_preprocess_bwd<RAWP><<<(N + LG_PP - 1) / LG_PP, LG_PP, 0, stream>>>(                                                           \
        N, g->M, v->sh_degree, W, H, v->tanfovx, v->tanfovy, v->scale_modifier, v->viewmatrix, v->projmatrix, v->campos, g->means3D,  \
        g->shs, g->shs_rest, g->colors_precomp, g->opacities, g->scales, g->rotations, g->cov3D_precomp, radii, geo.aux, geo.touched,  \
        geo.offsets, reinterpret_cast<const float4*>(rows), dL_dmeans2D, dL_dmeans3D, dL_dshs, dL_dshs_rest, dL_dcolors, dL_dopacity,   \
        dL_dscales, dL_drotations, dL_dcov3D)
        if (v->flags & LG_FLAG_RAW_PARAMS) LAUNCH_PPB(true); else LAUNCH_PPB(false);
#undef LAUNCH_PPB
    }
    KCHECK("lg_preprocess_bwd");
    return LG_OK;
}

extern "C" int lg_score_from_count(int32_t N, const int32_t* count, const float* weight, float* score, void* stream_p)
{
    if (N < 0 || (N > 0 && (!count || !score))) return fail(LG_ERR_INVALID_ARGUMENT, "bad arguments");
    if (N == 0) return LG_OK;
    hipStream_t stream = (hipStream_t)stream_p;
    lg_score_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, count, weight, score);
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
        lg_knn_cells<<<nb, 256, 0, stream>>>(P, kv.cap, level, points, kv.box, kv.keys_in, kv.vals_in);
        KCHECK("lg_knn_cells");
        size_t tb = kv.sort_temp_bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(kv.sort_temp, tb, kv.keys_in, kv.keys_out, kv.vals_in, kv.vals_out, P, 0, key_bits, stream));
        HIP_TRY(hipMemsetAsync(kv.cell_start, 0, (size_t)kv.cap * 4, stream));
        HIP_TRY(hipMemsetAsync(kv.cell_end, 0, (size_t)kv.cap * 4, stream));
        lg_knn_ranges<<<nb, 256, 0, stream>>>(P, points, kv.keys_out, kv.vals_out, kv.cell_start, kv.cell_end, kv.sorted);
        KCHECK("lg_knn_ranges");
        uint32_t* open_in = (level & 1) ? kv.open_a : kv.open_b;
        uint32_t* open_out = (level & 1) ? kv.open_b : kv.open_a;
        const int max_rings = level == LG_KNN_LEVELS - 1 ? (1 << 30) : LG_KNN_RINGS;
        lg_knn_query<<<nb, 256, 0, stream>>>(P, kv.cap, level, max_rings, points, kv.box, kv.sorted, kv.cell_start, kv.cell_end, open_in,
                                             kv.box + 8 + (level > 0 ? level - 1 : 0), open_out, kv.box + 8 + level, mean_dist2);
        KCHECK("lg_knn_query");
    }
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
    dim3 grid((W + LG_LOSS_TILE - 1) / LG_LOSS_TILE, (H + LG_LOSS_TILE - 1) / LG_LOSS_TILE, C);
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
        lg_loss_fwd<<<grid, 256, 0, stream>>>(H, W, img, gt, lv.dmu1, lv.dsig1, lv.dsig12, lv.partials);
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
    dim3 grid((W + LG_LOSS_TILE - 1) / LG_LOSS_TILE, (H + LG_LOSS_TILE - 1) / LG_LOSS_TILE, C);
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
        lg_loss_bwd<<<grid, 256, 0, stream>>>(H, W, img, gt, lv.dmu1, lv.dsig1, lv.dsig12, dL_dl1, scale_l1, dL_dssim, scale_ssim,
                                              (float)(1.0 / ((double)C * H * W)), dL_dimg);
        KCHECK("lg_loss_bwd");
    }
    return LG_OK;
}

extern "C" int lg_debug_reduce9(const float* in_64x9, float* out_9, void* stream_p)
{
    lg_debug_reduce9_kernel<<<1, 64, 0, (hipStream_t)stream_p>>>(in_64x9, out_9);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LG_ERR_DEVICE, "lg_debug_reduce9 launch", e);
    return LG_OK;
}

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

