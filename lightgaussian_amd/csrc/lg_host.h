// lg_host.h -- host-side plumbing: error strings, optional per-kernel hipEvent profiler, scratch-buffer carving (GeomView / ImgView / BinView)
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/lightgaussian.h"
#include "../../include/lightgaussian_debug.h"
#include "lg_math.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
static thread_local std::string g_err;
static thread_local lg_stats g_stats = {0, 0};

static int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) return fail(LG_ERR_DEVICE, #expr, _e);      \
    } while (0)

// Clear of device words as a KERNEL node.  The forward / backward use this instead of hipMemsetAsync / device-to-device
// hipMemcpyAsync: inside a captured HIP graph (ROCm 7.2) memset / memcpy nodes between kernel nodes did not keep the kernels
// behind them ordered after the kernels before them -- replays with a moved camera blended the previous replay's lists
// (tools/reuse_probe.py graph).  With kernel nodes only, the captured forward / step replays bit-identically.
__global__ void __launch_bounds__(256) lg_zero_words(uint32_t* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline hipError_t lg_zero_async(void* p, size_t bytes, hipStream_t stream)     // bytes % 4 == 0, p 4-byte aligned
{
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    lg_zero_words<<<(unsigned)std::min<size_t>((n + 255) / 256, 2048), 256, 0, stream>>>((uint32_t*)p, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// optional per-kernel event timing (LG_FLAG_PROFILE)
struct ProfEntry { std::string name; double ms = 0; int64_t n = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };
// process-wide (autograd runs backward on its own thread), guarded by a mutex
static std::vector<ProfEntry> g_prof;
static std::mutex g_prof_mu;

static ProfEntry& prof_entry(const char* name)
{
    for (auto& p : g_prof) if (p.name == name) return p;
    g_prof.emplace_back();
    g_prof.back().name = name;
    return g_prof.back();
}
struct ProfScope {
    hipEvent_t a = nullptr, b = nullptr; hipStream_t s; const char* name; bool on;
    ProfScope(bool on_, const char* n, hipStream_t st) : s(st), name(n), on(on_)
    {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            prof_entry(name).pending.emplace_back(a, b);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// scratch carving (all sub-buffers 256-byte aligned)
static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// float4s per blend record.  3 are used.  A stride of 4 (every record inside one 64-byte sector instead of straddling
// two lines 37 % of the time) was measured: blend kernels unchanged (VALU-bound), preprocess 0.236 -> 0.274 ms for the
// wider strided store -- so the packed 48-byte record stays.
#ifndef LG_REC_F4
#define LG_REC_F4 3
#endif

struct GeomView {
    float4* rec;        // [N][LG_REC_F4]  blend record {x, y, ha, nb} {hc, opacity, r, g} {b, hx, hy, id | SH clamp flags << 29}
    uint4* tinfo;       // [N]     binning record: x = tx0 | ty0<<16, y = tx1 | ty1<<16 (tight tile rect), z = depth bits
    uint32_t* touched;  // [N]  instance count per Gaussian (K1)
    uint32_t* offsets;  // [N]  inclusive scan of touched (written by K3; K9 derives the slot base from it)
    float* shjac;       // [N][9] d rgb / d (unit view direction) of the SH expansion, written by K1 when the view carries
                        //      LG_FLAG_SAVE_SH_JACOBIAN (a forward whose backward will follow); K9 then needs no SH coefficients.
                        //      counters[9] = LG_SHJAC_MAGIC says the rows of this view are there
    uint8_t* visible;   // [N]  1 = radius > 0 (K1): the render package's visibility_filter, read in place through lg_geom_visible_offset()
    uint32_t* counters; // [16] per-view device words: 0 = abort flags, 1 = prefiltered violation, 2 = largest depth bit
                        //      pattern, 3 = instance count R (all written by lg_scan_blocks); 8 = arrival counter of
                        //      lg_scan_blocks (zeroed by K1, left at zero by the scan)
    uint32_t* blk_dmax; // [ceil(N/64)] per-K1-workgroup largest depth bit pattern (bit 31: prefiltered violation)
    uint32_t* blk_sum;  // [ceil(N/64)] per-K1-workgroup instance count
    uint32_t* blk_off;  // [ceil(N/64)] its exclusive scan inside each part of 1024 words (lg_scan_blocks)
    uint32_t *part_sum, *part_dmax, *part_prefix;   // [ceil(N/65536)] per part: instance count, depth maximum, exclusive scan
    size_t total;
};

#define LG_SHJAC_MAGIC 0x4A414353u
static GeomView carve_geom(void* base, int N)
{
    GeomView g;
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    size_t n = (size_t)(N > 0 ? N : 1);
    g.rec = (float4*)take(n * 16 * LG_REC_F4);
    g.tinfo = (uint4*)take(n * 16);
    g.touched = (uint32_t*)take(n * 4);
    g.offsets = (uint32_t*)take(n * 4);
    g.visible = (uint8_t*)take(n);
    g.counters = (uint32_t*)take(64);
    g.shjac = (float*)take(n * 36);
    g.blk_dmax = (uint32_t*)take(((n + 63) / 64) * 4);
    g.blk_sum = (uint32_t*)take(((n + 63) / 64) * 4);
    g.blk_off = (uint32_t*)take(((n + 63) / 64) * 4);
    g.part_sum = (uint32_t*)take(((n + 65535) / 65536) * 4);
    g.part_dmax = (uint32_t*)take(((n + 65535) / 65536) * 4);
    g.part_prefix = (uint32_t*)take(((n + 65535) / 65536) * 4);
    g.total = off;
    return g;
}

// K4: keys-only radix sort of the packed 64-bit keys: our own onesweep kernels (lg_sort.h).  Tile shape by a measured
// sweep on MI355X at C3 (~4 M keys), -DLG_SORT_BLOCK / -DLG_SORT_ITEMS to change it.
#include "lg_sort.h"

struct ImgView { float* final_T; uint32_t* n_contrib; size_t total; };
static ImgView carve_img(void* base, int W, int H)
{
    ImgView v; size_t off = 0; char* p = (char*)base; size_t P = (size_t)W * H;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    v.final_T = (float*)take(P * 4);
    v.n_contrib = (uint32_t*)take(P * 4);
    v.total = off;
    return v;
}

// Binning buffer.  There is no separate id list and no slot list: the id is the low field of the sorted key, and the pre-sort
// slot (row address of the backward) is recomputed from the Gaussian's tile rectangle (tinfo).
// Long per-tile lists (real captures: tens of thousands of entries on a few tiles) are cut into SEGMENTS of S entries
// (lg_view.segment_length, default LG_DEFAULT_SEGMENT): the forward leaves one checkpoint record per pixel at the end of every
// segment of such a tile, and the backward runs one wave per (tile, segment) instead of one wave per tile -- a 20 000-entry
// tile becomes ten independent work items instead of one 4 ms serial chain.  Tiles with at most one segment never
// touch the checkpoints.  S is part of the VIEW (the caller passes the same lg_view to the
// forward and to its backward; the forward also stores it in meta[2] and the backward kernels refuse to run on a mismatch):
// the library keeps no state of its own.  512 by measurement (the sweep over S and scenes: EXPERIMENTS.md, "segment length");
// 64 / 128 exercise the machinery on small scenes (tests).
#define LG_DEFAULT_SEGMENT 512
static inline int lg_segment_of(const lg_view* v) { return v->segment_length > 0 ? v->segment_length : LG_DEFAULT_SEGMENT; }

struct BinView {
    uint2* ranges;                  // [tiles]
    uint2* work;                    // [tiles + R / S + 1] work items {tile, segment} of the backward blend, longest first
    uint2* par_work;                // [tiles + R / S + 1] the items of the tiles whose list goes through the parallel long-tile forward
    uint32_t* par_arrived;          // [tiles] per long tile: segments that finished pass 1 (the last one to arrive scans the tile)
    uint32_t* long_tiles;           // [1 + tiles] second sort stage: [0] = number of lists beyond a workgroup's LDS capacity, then their tiles (lg_tile_sort -> lg_tile_sort_long)
    uint32_t* meta;                 // [16] 0 = number of work items, 1 = longest list of the view, 2 = S, 3 = par_min of the view
                                    //      (0 = none), 4 = number of par_work items (all written by lg_work_order_body)
    float4* ckpt;                   // [2 (R / S + 1)][256] checkpoint records {T, segment colour} of long tiles (lg_blend_fwd)
    uint32_t* ckpt_last;            // [2 (R / S + 1)][256] last contributing list position per (segment, pixel): pass 1 -> join of
                                    //     the parallel long-tile forward (lg_blend_fwd_seg / _scan / _rewalk)
    uint64_t* entries;              // [R] sorted list entries = the sorted keys (tile | depth | id); the low bits_for(N) bits are the Gaussian id
    uint64_t* keys_in;              // [R] radix-sort input; free after the sort: ping-pong buffer of lg_tile_sort_long / of lg_tile_ranges' long runs,
                                    //     then -- per-hit weight policies -- the per-instance {count | weight} words of lg_blend_fwd<COUNT, FSCORE>
    void* sort_temp; size_t sort_temp_bytes; size_t total;
};
static int bits_for(uint32_t n) // smallest b with 2^b >= n
{
    int b = 0;
    while (b < 32 && (1ull << b) < n) b++;
    return b;
}
static BinView carve_bin(void* base, int64_t R, int W, int H, int seg)
{
    BinView v; memset(&v, 0, sizeof(v)); size_t off = 0; char* p = (char*)base;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    size_t n = (size_t)(R > 0 ? R : 1);
    const int gx = (W + LG_TILE - 1) / LG_TILE, gy = (H + LG_TILE - 1) / LG_TILE;
    const size_t S = (size_t)(seg > 0 ? seg : LG_DEFAULT_SEGMENT);
    v.ranges = (uint2*)take((size_t)gx * gy * 8);
    v.meta = (uint32_t*)take(64);                 // before anything whose size depends on S: the backward finds meta[2] (the forward's S) whatever S it was handed
    v.work = (uint2*)take(((size_t)gx * gy + n / S + 1) * 8);
    v.par_work = (uint2*)take(((size_t)gx * gy + n / S + 1) * 8);
    v.par_arrived = (uint32_t*)take((size_t)gx * gy * 4);
    v.long_tiles = (uint32_t*)take(((size_t)gx * gy + 1) * 4);
    v.ckpt = (float4*)take(2 * (n / S + 1) * 256 * 16);
    v.ckpt_last = (uint32_t*)take(2 * (n / S + 1) * 256 * 4);
    v.entries = (uint64_t*)take(n * 8);
    v.keys_in = (uint64_t*)take(n * 8);
    const size_t tb = lg_sort_layout(n).total;
    v.sort_temp_bytes = tb;
    v.sort_temp = take(tb);
    v.total = off;
    return v;
}

extern "C" size_t lg_geom_bytes(int32_t N) { return carve_geom(nullptr, N).total; }
extern "C" size_t lg_geom_visible_offset(int32_t N) { return (size_t)((char*)carve_geom((void*)256, N).visible - (char*)256); }
extern "C" size_t lg_img_bytes(int32_t W, int32_t H) { return carve_img(nullptr, W, H).total; }
extern "C" size_t lg_binning_bytes(int64_t R, int32_t W, int32_t H, int32_t segment_length)
{
    if (segment_length != 0 && (segment_length < 64 || segment_length % 64 != 0)) return 0;   // (lg_forward rejects such a view)
    return carve_bin(nullptr, R, W, H, segment_length).total;
}
extern "C" size_t lg_backward_scratch_bytes(int32_t N, int64_t R)
{
    (void)N;
    return align_up((size_t)(R > 0 ? R : 1) * 12 * sizeof(float)); // one 48-byte gradient row per (tile, Gaussian) instance
}

