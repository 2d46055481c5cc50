// lg_kernels.hip -- gfx950 (CDNA4, wave64) kernels + C ABI of the LightGaussian rasterizer.
//
// Pipeline (one view):
//   K1 lg_preprocess      per Gaussian : project, EWA splat, SH->RGB, exact footprint culling, tile count
//   K2 inclusive scan     (rocPRIM)    : instance offsets, R = total instances
//   K3 lg_duplicate       per Gaussian : (tile<<32 | depth bits, id) for every tile of the tight rectangle
//   K4 radix sort         (rocPRIM)    : stable, by (tile, depth); ties keep Gaussian-id order
//   K5 lg_finalize_bins   per instance : [start,end) of every tile, sorted Gaussian ids, slot->position map
//   K6 lg_blend_fwd       per tile     : 4 autonomous waves (8x8 pixels each); each wave compacts the
//                                        Gaussians overlapping ITS 8x8 block into an LDS queue and blends
//                                        front-to-back; wave-ballot early termination; count variant
//                                        does one int atomic per (wave, Gaussian) from a popcount
//   K7 lg_blend_bwd       per tile     : same traversal back-to-front; the 9 per-pair partials are
//                                        reduced across the wave with DPP adds, one atomic set per wave
//   K8+K9 lg_preprocess_bwd per Gaussian: conic/mean2D/colour grads -> means3D, SH, scale, rotation
//
// Written for wave64 / 256-thread workgroups; no CUDA compatibility paths.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lightgaussian.h"
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

struct GeomView {
    float4* rec;        // [N][3]  blend record {x, y, ha, nb} {hc, opacity, r, g} {b, hx, hy, id-bits}
    float4* aux;        // [N][2]  backward record {cov3D[0..3]} {cov3D[4], cov3D[5], clamp-bits, -}
    uint4* tinfo;       // [N]     binning record: x = tx0 | ty0<<16, y = tx1 | ty1<<16 (tight tile rect), z = depth bits
    uint32_t* touched;  // [N]
    uint32_t* offsets;  // [N] inclusive scan of touched
    uint32_t* counters; // [16]: 1 = prefiltered violation, 2 = largest depth bit pattern
    uint32_t* blk_dmax; // [ceil(N/64)] per-workgroup largest depth bit pattern (reduced by lg_reduce_dmax)
    void* scan_temp; size_t scan_temp_bytes;
    size_t total;
};

static size_t scan_temp_bytes_for(int N)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, N);
    return bytes;
}

static GeomView carve_geom(void* base, int N)
{
    GeomView g;
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    size_t n = (size_t)(N > 0 ? N : 1);
    g.rec = (float4*)take(n * 48);
    g.aux = (float4*)take(n * 32);
    g.tinfo = (uint4*)take(n * 16);
    g.touched = (uint32_t*)take(n * 4);
    g.offsets = (uint32_t*)take(n * 4);
    g.counters = (uint32_t*)take(64);
    g.blk_dmax = (uint32_t*)take(((n + 63) / 64) * 4);
    g.scan_temp_bytes = scan_temp_bytes_for((int)n);
    g.scan_temp = take(g.scan_temp_bytes);
    g.total = off;
    return g;
}

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

// Binning buffer.  The first three arrays are what the blend kernels and the backward read; they sit at the same
// offsets for both key formats.
struct BinView {
    uint2* ranges;                  // [tiles]
    uint32_t* point_list;           // [R] Gaussian ids in (tile, depth, id) order
    uint32_t* slot_out;             // [R] pre-sort slot of every sorted position (row address of the backward)
    uint64_t *keys_in, *keys_out;   // [R] radix-sort double buffer
    uint32_t* slot_in;              // [R] (pairs format only) iota values carried through the sort
    uint32_t* gid_slot;             // [R] (pairs format only) Gaussian id of every pre-sort slot
    void* sort_temp; size_t sort_temp_bytes; size_t total;
};
static int bits_for(uint32_t n) // smallest b with 2^b >= n
{
    int b = 0;
    while (b < 32 && (1ull << b) < n) b++;
    return b;
}
static BinView carve_bin(void* base, int64_t R, int W, int H, bool packed)
{
    BinView v; memset(&v, 0, sizeof(v)); size_t off = 0; char* p = (char*)base;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    size_t n = (size_t)(R > 0 ? R : 1);
    const int gx = (W + LG_TILE - 1) / LG_TILE, gy = (H + LG_TILE - 1) / LG_TILE;
    v.ranges = (uint2*)take((size_t)gx * gy * 8);
    v.point_list = (uint32_t*)take(n * 4);
    v.slot_out = (uint32_t*)take(n * 4);
    v.keys_in = (uint64_t*)take(n * 8);
    v.keys_out = (uint64_t*)take(n * 8);
    size_t tb = 0;
    if (packed) {
        (void)hipcub::DeviceRadixSort::SortKeys(nullptr, tb, (uint64_t*)nullptr, (uint64_t*)nullptr, (int)n, 0, 64);
    } else {
        v.slot_in = (uint32_t*)take(n * 4);
        v.gid_slot = (uint32_t*)take(n * 4);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                                 (uint32_t*)nullptr, (int)n, 0, 64);
    }
    v.sort_temp_bytes = tb;
    v.sort_temp = take(tb);
    v.total = off;
    return v;
}

extern "C" size_t lg_geom_bytes(int32_t N) { return carve_geom(nullptr, N).total; }
extern "C" size_t lg_img_bytes(int32_t W, int32_t H) { return carve_img(nullptr, W, H).total; }
extern "C" size_t lg_binning_bytes(int64_t R, int32_t W, int32_t H)
{
    const size_t a = carve_bin(nullptr, R, W, H, true).total, b = carve_bin(nullptr, R, W, H, false).total;
    return a > b ? a : b; // upper bound over both key formats
}
extern "C" size_t lg_backward_scratch_bytes(int32_t N, int64_t R)
{
    (void)N;
    return align_up((size_t)(R > 0 ? R : 1) * 12 * sizeof(float)); // one 48-byte gradient row per (tile, Gaussian) instance
}

// ------------------------------------------------------------------------------------------------
// wave64 helpers
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t prefix_popc(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// DPP wave reduction: after the call lane 63 holds the sum over all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111, 0xf>(v); // row_shr:1
    v = dpp_add<0x112, 0xf>(v); // row_shr:2
    v = dpp_add<0x114, 0xf>(v); // row_shr:4
    v = dpp_add<0x118, 0xf>(v); // row_shr:8   -> lane 15 of every row holds the row total
    v = dpp_add<0x142, 0xa>(v); // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v); // row_bcast:31 into rows 2,3 -> lane 63 = total
    return v;
}

// ------------------------------------------------------------------------------------------------
// K1: preprocess.  One wave per workgroup, 64 consecutive Gaussians.  The SH rows of the wave (64 x 12M
// bytes, contiguous in memory) are fetched with coalesced 16-byte loads into LDS -- skipping rows of
// culled Gaussians -- instead of 48 strided dword loads per lane.
#define LG_PP 64
#define LG_SH_MAXF 48 // floats per SH row at M = 16
#define LG_COOP_ROWS 48u // K9: splats with more tile instances than this are gathered by the whole wave

// cooperative copy of the wave's SH rows into LDS (flat layout, row stride = rowf floats)
__device__ __forceinline__ void stage_sh_rows(const float* __restrict__ shs, int i0, int rows, int rowf, uint64_t need_mask,
                                              float* lds, uint32_t lane)
{
    const float* src = shs + (size_t)i0 * rowf;
    const int nfl = rows * rowf;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    if (vec_ok) {
        const int nvec = nfl >> 2;
        for (int q = (int)lane; q < nvec; q += LG_PP) {
            const int f = q << 2;
            // rows are skipped only when a float4 never straddles two rows
            if ((rowf & 3) == 0 && !((need_mask >> (f / rowf)) & 1ull)) continue;
            *reinterpret_cast<float4*>(lds + f) = *reinterpret_cast<const float4*>(src + f);
        }
        for (int f = (nvec << 2) + (int)lane; f < nfl; f += LG_PP) lds[f] = src[f];
    } else {
        for (int f = (int)lane; f < nfl; f += LG_PP) lds[f] = src[f];
    }
}

// RAW (section 8f row 1, "fused getters"): the inputs are GaussianModel's raw parameters -- log-scales, unnormalised
// quaternions, opacity logits, and the SH coefficients as the two tensors _features_dc [N,1,3] / _features_rest
// [N,M-1,3] -- and the activations (scene/gaussian_model.py:98-118) are evaluated here instead of by torch.
__device__ __forceinline__ float lg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <bool RAW>
__global__ void __launch_bounds__(LG_PP)
lg_preprocess(int N, int M, int D, int W, int H, float tanfovx, float tanfovy, float mod, int prefiltered,
              const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
              const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest,
              const float* __restrict__ colors_precomp,
              const float* __restrict__ opacities, const float* __restrict__ scales, const float* __restrict__ rotations,
              const float* __restrict__ cov3D_precomp, GeomView g, int32_t* __restrict__ radii)
{
    __shared__ __attribute__((aligned(16))) float sh_rows[LG_PP * LG_SH_MAXF];
    const uint32_t lane = threadIdx.x;
    const int i0 = blockIdx.x * LG_PP;
    const int i = i0 + (int)lane;
    float vm[16], pm[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = viewmatrix[k]; pm[k] = projmatrix[k]; }
    cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
    bool vis = false;
    float px = 0, py = 0, pz = 0, op = 0;
    float cov[6] = {0, 0, 0, 0, 0, 0};
    LgSplat sp;
    if (i < N) {
        px = means3D[3 * (size_t)i]; py = means3D[3 * (size_t)i + 1]; pz = means3D[3 * (size_t)i + 2];
        // near-plane test first so culled Gaussians cost 12 bytes of reads
        const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        if (vz > 0.2f) {
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) cov[k] = cov3D_precomp[6 * (size_t)i + k];
            } else {
                float sc[3] = { scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2] };
                const float4 q4 = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
                float q[4] = { q4.x, q4.y, q4.z, q4.w };
                if (RAW) {
                    sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]);
                    const float inv = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f); // F.normalize
                    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
                }
                lg_cov3d(sc, mod, q, cov);
            }
            op = RAW ? lg_sigmoid(opacities[i]) : opacities[i];
            vis = lg_project(vm, pm, px, py, pz, cov, op, W, H, tanfovx, tanfovy, sp);
        } else if (prefiltered) {
            g.counters[1] = 1u;
        }
    }
    const uint64_t vmask = __ballot(vis);
    const bool split = RAW && shs_rest != nullptr;        // dc and rest are separate tensors
    const int rowf = split ? 3 * (M - 1) : 3 * M;          // floats per LDS-staged row
    if (shs && vmask && rowf > 0) {
        stage_sh_rows(split ? shs_rest : shs, i0, min(LG_PP, N - i0), rowf, vmask, sh_rows, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (i < N) {
        uint32_t touched = 0;
        int radius = 0;
        if (vis) {
            radius = sp.radius;
            touched = (uint32_t)((sp.tx1 - sp.tx0) * (sp.ty1 - sp.ty0));
            float rgb[3];
            uint32_t cb = 0;
            if (colors_precomp) {
                rgb[0] = colors_precomp[3 * (size_t)i]; rgb[1] = colors_precomp[3 * (size_t)i + 1]; rgb[2] = colors_precomp[3 * (size_t)i + 2];
            } else {
                float sh[LG_SH_MAXF];
                const float* row = sh_rows + lane * rowf;
                const int nact = (D + 1) * (D + 1) * 3;
                if (split) {
                    sh[0] = shs[3 * (size_t)i]; sh[1] = shs[3 * (size_t)i + 1]; sh[2] = shs[3 * (size_t)i + 2];
#pragma unroll
                    for (int k = 3; k < LG_SH_MAXF; k++) sh[k] = (k < nact) ? row[k - 3] : 0.0f;
                } else if ((rowf & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < LG_SH_MAXF / 4; q++) {
                        float4 v4 = make_float4(0, 0, 0, 0);
                        if (q * 4 < nact) v4 = reinterpret_cast<const float4*>(row)[q];
                        sh[4 * q] = v4.x; sh[4 * q + 1] = v4.y; sh[4 * q + 2] = v4.z; sh[4 * q + 3] = v4.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < LG_SH_MAXF; k++) sh[k] = (k < nact) ? row[k] : 0.0f;
                }
                lg_sh_to_rgb(D, sh, px, py, pz, cp, rgb, cb);
            }
            g.rec[3 * (size_t)i + 0] = make_float4(sp.x, sp.y, sp.ha, sp.nb);
            g.rec[3 * (size_t)i + 1] = make_float4(sp.hc, op, rgb[0], rgb[1]);
            g.rec[3 * (size_t)i + 2] = make_float4(rgb[2], sp.hx, sp.hy, __uint_as_float((uint32_t)i));
            g.aux[2 * (size_t)i + 0] = make_float4(cov[0], cov[1], cov[2], cov[3]);
            g.aux[2 * (size_t)i + 1] = make_float4(cov[4], cov[5], __uint_as_float(cb), 0.0f);
            g.tinfo[i] = make_uint4((uint32_t)sp.tx0 | ((uint32_t)sp.ty0 << 16), (uint32_t)sp.tx1 | ((uint32_t)sp.ty1 << 16),
                                    __float_as_uint(sp.depth), 0u);
        }
        radii[i] = radius;
        g.touched[i] = touched;
    }
    // (no global visible-counter: 47k same-address atomics serialise at ~11 ns each -- more than the whole kernel)
    // largest depth of the workgroup (bit pattern; positive floats order like integers), for the packed sort key.
    // Written per workgroup and reduced by a one-block kernel: a shared atomicMax serialises the first ~3k waves.
    uint32_t dmax = vis ? __float_as_uint(sp.depth) : 0u;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, sh));
    if (lane == 0) g.blk_dmax[blockIdx.x] = dmax;
}

// ------------------------------------------------------------------------------------------------
// max over the per-workgroup depth maxima -> counters[2]
__global__ void __launch_bounds__(1024)
lg_reduce_dmax(int nblk, const uint32_t* __restrict__ blk_dmax, uint32_t* __restrict__ counters)
{
    __shared__ uint32_t wmax[16];
    uint32_t m = 0;
    for (int i = threadIdx.x; i < nblk; i += 1024) m = max(m, blk_dmax[i]);
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, sh));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) m = max(m, wmax[w]);
        counters[2] = m;
    }
}

// K3: duplicate with keys
// Key formats.  PACKED: tile | (depth bits - bias) | Gaussian id in one u64, sorted keys-only on the tile+depth
// bits: the stable radix sort keeps the emission (= id) order among equal depths, and the id rides along for free
// (5 passes x 16 B instead of 6 x 24 B).  PAIRS (fallback when the fields do not fit 64 bits): tile<<32 | depth
// with the pre-sort slot as value.
#define LG_DEPTH_BIAS (124u << 23) // bit pattern of 0.125f < the 0.2 near plane

template <bool PACKED>
__global__ void __launch_bounds__(256)
lg_duplicate(int N, int gx, int depth_bits, int gid_bits, const uint32_t* __restrict__ touched, const uint32_t* __restrict__ offsets,
             uint4* __restrict__ tinfo, uint64_t* __restrict__ keys, uint32_t* __restrict__ slots, uint32_t* __restrict__ gid_slot)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t t = touched[i];
    if (t == 0) return;
    uint32_t off = offsets[i] - t;
    const uint4 r = tinfo[i];
    const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF, y1 = r.y >> 16;
    if (PACKED) {
        tinfo[i].w = off; // slot base, read back by lg_finalize_bins
        const uint64_t low = ((uint64_t)(r.z - LG_DEPTH_BIAS) << gid_bits) | (uint32_t)i;
        const int sh = depth_bits + gid_bits;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) keys[off++] = ((uint64_t)(uint32_t)(y * gx + x) << sh) | low;
    } else {
        const uint64_t d = (uint64_t)r.z;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | d;
                slots[off] = off;
                gid_slot[off] = (uint32_t)i;
                off++;
            }
    }
}

// K5: per sorted position: tile ranges, Gaussian id (point_list) and pre-sort slot
template <bool PACKED>
__global__ void __launch_bounds__(256)
lg_finalize_bins(uint32_t R, int gx, int depth_bits, int gid_bits, const uint64_t* __restrict__ keys, const uint4* __restrict__ tinfo,
                 const uint32_t* __restrict__ slot_sorted, const uint32_t* __restrict__ gid_slot, uint32_t* __restrict__ point_list,
                 uint32_t* __restrict__ slot_out, uint2* __restrict__ ranges)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int tsh = PACKED ? depth_bits + gid_bits : 32;
    const uint64_t key = keys[i];
    const uint32_t t = (uint32_t)(key >> tsh);
    if (PACKED) {
        const uint32_t gid = (uint32_t)(key & ((1ull << gid_bits) - 1ull));
        const uint4 r = tinfo[gid];
        const int x0 = r.x & 0xFFFF, y0 = r.x >> 16, x1 = r.y & 0xFFFF;
        const int tx = (int)(t % (uint32_t)gx), ty = (int)(t / (uint32_t)gx);
        point_list[i] = gid;
        slot_out[i] = r.w + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
    } else {
        point_list[i] = gid_slot[slot_sorted[i]]; // slot_out was written by the sort itself
    }
    if (i == 0) ranges[t].x = 0;
    else {
        const uint32_t tp = (uint32_t)(keys[i - 1] >> tsh);
        if (t != tp) { ranges[tp].y = i; ranges[t].x = i; }
    }
    if (i == R - 1) ranges[t].y = R;
}

// ------------------------------------------------------------------------------------------------
// tile <-> workgroup mapping.  Workgroup b runs on XCD b % 8 (observed dispatch order); give every
// XCD a contiguous band of tiles so that neighbouring tiles -- which share Gaussians -- hit the same L2.
__device__ __forceinline__ int xcd_tile(int b, int ntiles_pad8)
{
    const int per = ntiles_pad8 >> 3;
    return (b & 7) * per + (b >> 3);
}

#define LG_Q 64 // LDS queue depth per wave = one batch

// Select-based (no divergent control flow) front-to-back step.  Same canonical operations as lg_blend_pair on
// every lane that contributes, so results are bit-identical; rejected lanes compute and discard.
template <bool EXACT>
__device__ __forceinline__ bool fwd_pair(const float4& a, const float4& b, const float4& c, bool live, float pxf, float pyf, float& T,
                                         float& C0, float& C1, float& C2, bool& done, uint32_t& last, uint32_t rel, float& alpha_out)
{
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy);
    const float pe = fminf(power, 0.0f);
    const float ex = EXACT ? lg_exp(pe) : __expf(pe);
    const float alpha = fminf(LG_ALPHA_MAX, b.y * ex);
    const bool ok = live && (power <= 0.0f) && (alpha >= LG_ALPHA_MIN);
    const float test_T = T * (1.0f - alpha);
    const bool sat = ok && (test_T < LG_T_MIN);
    const bool contrib = ok && !sat;
    const float w = alpha * T;
    const float n0 = fmaf(b.z, w, C0), n1 = fmaf(b.w, w, C1), n2 = fmaf(c.x, w, C2);
    C0 = contrib ? n0 : C0; C1 = contrib ? n1 : C1; C2 = contrib ? n2 : C2;
    T = contrib ? test_T : T;
    last = contrib ? rel : last;
    done = done || sat;
    alpha_out = alpha;
    return contrib;
}

// K6 / K6c: forward blend
template <bool COUNT, bool FSCORE, bool EXACT>
__global__ void __launch_bounds__(256)
lg_blend_fwd(int W, int H, int gx, int ntiles, int ntiles_pad8, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, const float* __restrict__ bg,
             float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             int32_t* __restrict__ count, float* __restrict__ fscore, int weight_policy)
{
    __shared__ float4 q0[4][LG_Q], q1[4][LG_Q], q2[4][LG_Q];
    const int tile = xcd_tile(blockIdx.x, ntiles_pad8);
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

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t base = range.x; base < range.y; base += LG_Q) {
        if (__ballot(!done) == 0) break; // every pixel of this wave is saturated or outside
        const uint32_t idx = base + lane;
        bool hit = false;
        float4 r0, r1, r2;
        if (idx < range.y) {
            const uint32_t id = point_list[idx];
            r0 = rec[3 * (size_t)id]; r1 = rec[3 * (size_t)id + 1]; r2 = rec[3 * (size_t)id + 2];
            // footprint box (x +- hx, y +- hy) vs this wave's 8x8 pixel block; hx = inf when culling is off
            hit = (r0.x + r2.y >= bx0) && (r0.x - r2.y <= bx1) && (r0.y + r2.z >= by0) && (r0.y - r2.z <= by1);
        }
        uint64_t mask = __ballot(hit);
        if (mask == 0) continue;
        if (hit) {
            const uint32_t pos = prefix_popc(mask);
            q0[wave][pos] = r0; q1[wave][pos] = r1; q2[wave][pos] = r2;
        }
        __builtin_amdgcn_wave_barrier();
        int mycnt = 0;
        float myf = 0.0f;
        uint32_t j = 0;
        const uint32_t rel = base - range.x + 1; // contributor index of source lane 0
        while (mask) {
            const uint32_t src = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1;
            const float4 a = q0[wave][j], b = q1[wave][j], c = q2[wave][j];
            float alpha = 0.0f, Tprev = T;
            const int res = fwd_pair<EXACT>(a, b, c, !done, pxf, pyf, T, C0, C1, C2, done, last, rel + src, alpha) ? 1 : 0;
            if (COUNT) {
                const uint64_t cm = __ballot(res == 1);
                if (lane == j) mycnt = (int)__popcll(cm);
                if (FSCORE) {
                    float wv = (res == 1) ? (weight_policy == LG_W_ALPHA ? alpha : alpha * Tprev) : 0.0f;
                    wv = wave_sum_to_lane63(wv);
                    const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), 63));
                    if (lane == j) myf = tot;
                }
            }
            j++;
        }
        if (COUNT) {
            // lane j owns compacted entry j: one atomic per (wave, Gaussian), issued 64-wide
            if (lane < j && mycnt > 0) {
                const uint32_t id = __float_as_uint(q2[wave][lane].w);
                atomicAdd(&count[id], mycnt);
                if (FSCORE) atomicAdd(&fscore[id], myf);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (inside) {
        const size_t pid = (size_t)pyi * W + pxi, HW = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = fmaf(T, bg[0], C0);
        out_color[HW + pid] = fmaf(T, bg[1], C1);
        out_color[2 * HW + pid] = fmaf(T, bg[2], C2);
    }
}

// per-view score from the exact integer count (ONE / OPACITY weights)
__global__ void __launch_bounds__(256)
lg_score_kernel(int N, const int32_t* __restrict__ count, const float* __restrict__ weight, float* __restrict__ score)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int c = count[i];
    score[i] = c > 0 ? lg_seqsum32(weight ? weight[i] : 1.0f, (uint32_t)c) : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// K7: backward blend.  One wave per 16x16 tile, FOUR pixels per lane (the four 8x8 sub-blocks), so a
// (tile, Gaussian) instance is reduced across lanes once, not once per 8x8 block.  Per batch of 64 list
// entries: lane l gathers entry l and computes its 4-bit sub-block overlap mask; the wave then walks the
// batch back to front, evaluating an entry only on the sub-blocks it overlaps (scalar branches on the
// mask).  The 9 partials are reduced with permlane32/16 swaps + row DPP adds (8 values packed into two
// registers: ~20 instructions instead of 54), parked in LDS, and flushed once per batch with 64-wide
// atomics (lane j owns entry j).  acc: [N][12] floats (9 used): dmean2D px x,y | dA dB dC | dopacity | drgb
typedef unsigned lg_u2v __attribute__((ext_vector_type(2)));

// combine two registers into one: lower 32 lanes = 32-lane partial sums of a, upper 32 lanes = of b
__device__ __forceinline__ float fold32(float a, float b)
{
    lg_u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// rows (16 lanes) of the result: (a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3)
__device__ __forceinline__ float fold16(float a, float b)
{
    lg_u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float row_sum_to_lane15(float v)
{
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    return v;
}
// Sums p[0..8] over the wave and writes the 9 totals to dst[0..8] (LDS).  Which 16-lane row ends up
// with which value is fixed by the two folds: rows of w0 = (p0, p2, p1, p3), rows of w1 = (p4, p6, p5, p7).
__device__ __forceinline__ void wave_reduce9_to_lds(const float (&p)[9], float* dst, uint32_t lane)
{
    const float u0 = fold32(p[0], p[1]), u1 = fold32(p[2], p[3]), u2 = fold32(p[4], p[5]), u3 = fold32(p[6], p[7]);
    float w0 = fold16(u0, u1), w1 = fold16(u2, u3);
    w0 = row_sum_to_lane15(w0);
    w1 = row_sum_to_lane15(w1);
    const float w8 = wave_sum_to_lane63(p[8]);
    if ((lane & 15u) == 15u) {
        const uint32_t r = lane >> 4;
        const uint32_t k = ((r & 1u) << 1) | (r >> 1); // row -> value index inside the group of four
        dst[k] = w0;
        dst[4 + k] = w1;
    }
    if (lane == 63u) dst[8] = w8;
}

// One (pixel, Gaussian) step of the back-to-front replay.  EXACT = canonical arithmetic (same sequence as
// the oracle); otherwise hardware exp / rcp and contraction allowed (training path, 1e-4 contract).
template <bool EXACT>
__device__ __forceinline__ bool bwd_pair(const float4& a, const float4& b, const float4& c, float pxf, float pyf, float& T, float T_final,
                                         float g0, float g1, float g2, float bg_dot, float& a0, float& a1, float& a2, float& last_alpha,
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
        const float A = -2.0f * a.z, B = -a.w, Cc = -2.0f * b.x;
        T = T / (1.0f - alpha);
        const float dch = alpha * T;
        const float c0 = b.z, c1 = b.w, c2 = c.x;
        a0 = last_alpha * lc0 + (1.0f - last_alpha) * a0;
        a1 = last_alpha * lc1 + (1.0f - last_alpha) * a1;
        a2 = last_alpha * lc2 + (1.0f - last_alpha) * a2;
        lc0 = c0; lc1 = c1; lc2 = c2;
        float dL_dalpha = (c0 - a0) * g0 + (c1 - a1) * g1 + (c2 - a2) * g2;
        dL_dalpha = dL_dalpha * T;
        last_alpha = alpha;
        dL_dalpha = dL_dalpha + (-T_final / (1.0f - alpha)) * bg_dot;
        const float dL_dG = op * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        p[0] += dL_dG * (-gdx * A - gdy * B);
        p[1] += dL_dG * (-gdy * Cc - gdx * B);
        p[2] += -0.5f * gdx * dx * dL_dG;
        p[3] += -gdx * dy * dL_dG;
        p[4] += -0.5f * gdy * dy * dL_dG;
        p[5] += G * dL_dalpha;
        p[6] += dch * g0; p[7] += dch * g1; p[8] += dch * g2;
        return true;
    }
    return false;
}

// Training-path variant (hardware exp / rcp, contraction allowed), written BRANCH-FREE: every lane runs
// the whole sequence and invalid lanes are neutralised by zeroing dL/dalpha and the colour weight and by
// selecting the old state.  (A branchy version makes hipcc copy the 9 accumulators at every nesting level.)
__device__ __forceinline__ bool bwd_pair_fast(const float4& a, const float4& b, const float4& c, bool live, float pxf, float pyf,
                                              float& T, float T_final, float g0, float g1, float g2, float bg_dot, float& a0, float& a1,
                                              float& a2, float (&p)[9])
{
#pragma clang fp contract(fast)
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy); // identical to the forward's expression
    const float G = __expf(fminf(power, 0.0f));
    const float op = b.y;
    const float alpha = fminf(LG_ALPHA_MAX, op * G);
    const bool ok = live && (power <= 0.0f) && (alpha >= LG_ALPHA_MIN);
    // am = alpha on valid lanes, 0 elsewhere: with am = 0 the colour recurrence below is the identity (0*c + 1*a = a)
    // and dch = 0, so four of the seven selects of a naive branch-free form disappear (v_cndmask / v_cmp / v_min cost
    // ~1.7x an fma on gfx950, tools/ubench/valu_rate2.hip).  T keeps its select: rcp(1.0) need not be exactly 1.
    const float am = ok ? alpha : 0.0f;
    const float om = 1.0f - am;
    const float inv = __builtin_amdgcn_rcpf(om);
    const float Tn = T * inv;
    const float c0 = b.z, c1 = b.w, c2 = c.x;
    // a0..a2 = colour accumulated behind this entry (eager form of the published last_alpha/last_color recurrence)
    const float dch = am * Tn;
    float dL_dalpha = ((c0 - a0) * g0 + (c1 - a1) * g1 + (c2 - a2) * g2) * Tn - (T_final * inv) * bg_dot;
    dL_dalpha = ok ? dL_dalpha : 0.0f;
    T = ok ? Tn : T;
    a0 = am * c0 + om * a0;
    a1 = am * c1 + om * a1;
    a2 = am * c2 + om * a2;
    const float dL_dG = op * dL_dalpha;
    const float gdx = G * dx, gdy = G * dy;
    // with ha = -A/2, nb = -B, hc = -C/2:  -gdx*A - gdy*B = 2*ha*gdx + nb*gdy
    p[0] += dL_dG * (2.0f * a.z * gdx + a.w * gdy);
    p[1] += dL_dG * (2.0f * b.x * gdy + a.w * gdx);
    const float hg = -0.5f * dL_dG;
    p[2] += hg * gdx * dx;
    p[3] -= dL_dG * gdx * dy;
    p[4] += hg * gdy * dy;
    p[5] += G * dL_dalpha;
    p[6] += dch * g0; p[7] += dch * g1; p[8] += dch * g2;
    return ok;
}

template <bool EXACT, int ABL = 0>
__global__ void __launch_bounds__(64)
lg_blend_bwd(int W, int H, int gx, int ntiles, int ntiles_pad8, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ slot_sorted, const float4* __restrict__ rec,
             const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
             const float* __restrict__ dL_dpix, float* __restrict__ part)
{
    __shared__ float4 q0[LG_Q], q1[LG_Q], q2[LG_Q];
    __shared__ float stage[LG_Q * 9];
    const int tile = xcd_tile(blockIdx.x, ntiles_pad8);
    if (tile >= ntiles) return;
    const uint32_t lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)H * W;
    const float bgr = bg[0], bgg = bg[1], bgb = bg[2];

    float pxf[4], pyf[4], T[4], Tfin[4], g0[4], g1[4], g2[4], bgd[4], a0[4], a1[4], a2[4], la[4], lc0[4], lc1[4], lc2[4];
    uint32_t last[4];
    uint32_t wmax = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int pxi = tx * LG_TILE + (s & 1) * 8 + (int)(lane & 7), pyi = ty * LG_TILE + (s >> 1) * 8 + (int)(lane >> 3);
        const bool inside = pxi < W && pyi < H;
        const size_t pid = (size_t)pyi * W + pxi;
        pxf[s] = (float)pxi; pyf[s] = (float)pyi;
        Tfin[s] = inside ? final_T[pid] : 0.0f;
        T[s] = Tfin[s];
        last[s] = inside ? n_contrib[pid] : 0u;
        g0[s] = inside ? dL_dpix[pid] : 0.0f;
        g1[s] = inside ? dL_dpix[HW + pid] : 0.0f;
        g2[s] = inside ? dL_dpix[2 * HW + pid] : 0.0f;
        bgd[s] = bgr * g0[s] + bgg * g1[s] + bgb * g2[s];
        a0[s] = a1[s] = a2[s] = la[s] = lc0[s] = lc1[s] = lc2[s] = 0.0f;
        wmax = max(wmax, last[s]);
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, sh));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    const uint32_t n_list = range.y - range.x;
    if (n_list == 0) return;
    if (wmax > n_list) wmax = n_list;
    const float tbx = (float)(tx * LG_TILE), tby = (float)(ty * LG_TILE);
    float4* rows = reinterpret_cast<float4*>(part);

    // every list entry of the tile writes exactly one 48-byte row (zeros when nothing contributed) at its
    // PRE-SORT slot, where the rows of one Gaussian are contiguous: no zero-fill pass, no atomics, and K9
    // reads its rows sequentially and sums them in a fixed order (deterministic gradients)
    for (int k = (int)((n_list - 1) / LG_Q); k >= 0; k--) {
        const uint32_t base = range.x + (uint32_t)k * LG_Q;
        const uint32_t nbt = min((uint32_t)LG_Q, n_list - (uint32_t)k * LG_Q);                             // entries of this batch
        const uint32_t nb = wmax > (uint32_t)k * LG_Q ? min((uint32_t)LG_Q, wmax - (uint32_t)k * LG_Q) : 0u; // ... that any pixel reached
        uint64_t hitmask = 0;
        if (nb > 0) {
            float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
            if (lane < nb) {
                const uint32_t id = point_list[base + lane];
                r0 = rec[3 * (size_t)id]; r1 = rec[3 * (size_t)id + 1]; r2 = rec[3 * (size_t)id + 2];
                uint32_t m = 0;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const float bx0 = tbx + (float)((s & 1) * 8), by0 = tby + (float)((s >> 1) * 8);
                    const bool hit = (r0.x + r2.y >= bx0) && (r0.x - r2.y <= bx0 + 7.0f) && (r0.y + r2.z >= by0) && (r0.y - r2.z <= by0 + 7.0f);
                    m |= (hit ? 1u : 0u) << s;
                }
                q0[lane] = r0; q1[lane] = r1; q2[lane] = make_float4(r2.x, r2.y, r2.z, __uint_as_float(m));
            }
            __builtin_amdgcn_wave_barrier();
            for (int j = (int)nb - 1; j >= 0; j--) {
                const float4 c = q2[j];
                const uint32_t m = __builtin_amdgcn_readfirstlane(__float_as_uint(c.w));
                if (m == 0) continue;
                const float4 a = q0[j], b = q1[j];
                const uint32_t rel = (uint32_t)k * LG_Q + (uint32_t)j + 1u;
                float p[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                bool contrib = false;
                if (ABL == 4) { asm volatile("" ::"v"(a.x), "v"(b.x)); continue; }
                if (ABL == 3) {
                    contrib = rel <= last[0];
                    p[0] = a.x; p[1] = a.y; p[2] = b.x; p[8] = c.x;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        if (m & (1u << s)) {
                            if (EXACT) {
                                if (rel <= last[s])
                                    contrib |= bwd_pair<true>(a, b, c, pxf[s], pyf[s], T[s], Tfin[s], g0[s], g1[s], g2[s], bgd[s], a0[s], a1[s], a2[s],
                                                              la[s], lc0[s], lc1[s], lc2[s], p);
                            } else {
                                contrib |= bwd_pair_fast(a, b, c, rel <= last[s], pxf[s], pyf[s], T[s], Tfin[s], g0[s], g1[s], g2[s], bgd[s], a0[s],
                                                         a1[s], a2[s], p);
                            }
                        }
                    }
                }
                if (__ballot(contrib) == 0) continue;
                if (ABL == 2) {
                    asm volatile("" ::"v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]));
                    continue;
                }
                wave_reduce9_to_lds(p, stage + j * 9, lane);
                hitmask |= 1ull << j;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (ABL != 1 && lane < nbt) {
            float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = o0;
            if ((hitmask >> lane) & 1ull) {
                const float* src = stage + lane * 9;
                o0 = make_float4(src[0], src[1], src[2], src[3]);
                o1 = make_float4(src[4], src[5], src[6], src[7]);
                o2 = make_float4(src[8], 0.0f, 0.0f, 0.0f);
            }
            float4* dst = rows + 3 * (size_t)slot_sorted[base + lane];
            dst[0] = o0; dst[1] = o1; dst[2] = o2;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// K7 (training variant): backward blend with the pixel reduction on the f32 MFMA pipe.
//
// For one (Gaussian j, 8x8 sub-block) the nine gradient sums are contractions over the 64 pixels:
//   colour     p6..8 = sum_px DCH(px,j) * g_c(px)                          DCH = alpha*T, g = dL/dpixel
//   geometry   p0..5 = linear combinations of the six moments sum_px W(px,j) * {1,u,v,u^2,uv,v^2}
//              with W = dL/dG * G and (u,v) the pixel offset from the sub-block centre (|u|,|v| <= 3.5);
//              dx = X - u, dy = Y - v with (X,Y) = Gaussian centre relative to that centre
// i.e. two small GEMMs  [16 x 64] . [64 x 16 entries].  The VALU stream only produces W and DCH per pixel
// (no nine partial products, no cross-lane reduction); v_mfma_f32_16x16x4_f32 (exact f32, separate pipe)
// does the sums for 16 pending entries at a time.  Order of evaluation inside a sub-block is still back to
// front, and sub-blocks are independent pixel sets, so the batch is walked once per sub-block.
typedef float lg_f4v __attribute__((ext_vector_type(4)));
#define LG_MF_STRIDE 66 // floats per pending-entry row in LDS: (66*j + 4t + k) is conflict-free for the operand reads

__device__ __forceinline__ bool bwd_pair_wd(const float4& a, const float4& b, const float4& c, bool live, float pxf, float pyf, float& T,
                                            float T_final, float g0, float g1, float g2, float bg_dot, float& a0, float& a1, float& a2,
                                            float& Wout, float& Dout)
{
#pragma clang fp contract(fast)
    const float dx = a.x - pxf, dy = a.y - pyf;
    const float power = fmaf(fmaf(a.z, dx, a.w * dy), dx, (b.x * dy) * dy); // identical to the forward's expression
    const float G = __expf(fminf(power, 0.0f));
    const float op = b.y;
    const float alpha = fminf(LG_ALPHA_MAX, op * G);
    const bool ok = live && (power <= 0.0f) && (alpha >= LG_ALPHA_MIN);
    const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
    const float Tn = T * inv;
    const float c0 = b.z, c1 = b.w, c2 = c.x;
    float dL_dalpha = ((c0 - a0) * g0 + (c1 - a1) * g1 + (c2 - a2) * g2) * Tn - (T_final * inv) * bg_dot;
    const float om = 1.0f - alpha;
    T = ok ? Tn : T;
    a0 = ok ? alpha * c0 + om * a0 : a0;
    a1 = ok ? alpha * c1 + om * a1 : a1;
    a2 = ok ? alpha * c2 + om * a2 : a2;
    Wout = ok ? (op * dL_dalpha) * G : 0.0f;
    Dout = ok ? alpha * Tn : 0.0f;
    return ok;
}

__global__ void __launch_bounds__(64)
lg_blend_bwd_mfma(int W, int H, int gx, int ntiles, int ntiles_pad8, const uint2* __restrict__ ranges,
                  const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ slot_sorted, const float4* __restrict__ rec,
                  const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                  const float* __restrict__ dL_dpix, float* __restrict__ part)
{
    __shared__ float4 q0[LG_Q], q1[LG_Q], q2[LG_Q];
    __shared__ float stage[LG_Q * 9];
    __shared__ float wbuf[16 * LG_MF_STRIDE], dbuf[16 * LG_MF_STRIDE];
    __shared__ float gT[4 * 3 * 64];
    const int tile = xcd_tile(blockIdx.x, ntiles_pad8);
    if (tile >= ntiles) return;
    const uint32_t lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)H * W;
    const float bgr = bg[0], bgg = bg[1], bgb = bg[2];

    float T[4], Tfin[4], g0[4], g1[4], g2[4], bgd[4], a0[4], a1[4], a2[4];
    uint32_t last[4];
    uint32_t wmax = 0;
    const float lxf = (float)(lane & 7), lyf = (float)(lane >> 3);
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int pxi = tx * LG_TILE + (s & 1) * 8 + (int)(lane & 7), pyi = ty * LG_TILE + (s >> 1) * 8 + (int)(lane >> 3);
        const bool inside = pxi < W && pyi < H;
        const size_t pid = (size_t)pyi * W + pxi;
        Tfin[s] = inside ? final_T[pid] : 0.0f;
        T[s] = Tfin[s];
        last[s] = inside ? n_contrib[pid] : 0u;
        g0[s] = inside ? dL_dpix[pid] : 0.0f;
        g1[s] = inside ? dL_dpix[HW + pid] : 0.0f;
        g2[s] = inside ? dL_dpix[2 * HW + pid] : 0.0f;
        bgd[s] = bgr * g0[s] + bgg * g1[s] + bgb * g2[s];
        a0[s] = a1[s] = a2[s] = 0.0f;
        wmax = max(wmax, last[s]);
        gT[(s * 3 + 0) * 64 + lane] = g0[s]; gT[(s * 3 + 1) * 64 + lane] = g1[s]; gT[(s * 3 + 2) * 64 + lane] = g2[s];
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, sh));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    const uint32_t n_list = range.y - range.x;
    if (n_list == 0) return;
    if (wmax > n_list) wmax = n_list;
    const float tbx = (float)(tx * LG_TILE), tby = (float)(ty * LG_TILE);
    float4* rows = reinterpret_cast<float4*>(part);

    // MFMA operand geometry of this lane: output row i = lane % 16, k = lane / 16 (pixel 4t + k of K-step t), entry column j = lane % 16
    const uint32_t mi = lane & 15u, mk = lane >> 4;
    // A operand of the moment chain: basis_i(u, v) of pixel 4t + k, i = 0..5 -> {1, u, v, u^2, uv, v^2}, else 0 (same for all sub-blocks)
    float Aw[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const uint32_t p = 4u * (uint32_t)t + mk;
        const float u = (float)(p & 7u) - 3.5f, v = (float)(p >> 3) - 3.5f;
        float bval = 0.0f;
        bval = mi == 0u ? 1.0f : bval; bval = mi == 1u ? u : bval; bval = mi == 2u ? v : bval;
        bval = mi == 3u ? u * u : bval; bval = mi == 4u ? u * v : bval; bval = mi == 5u ? v * v : bval;
        Aw[t] = bval;
    }
    __builtin_amdgcn_wave_barrier();

    for (int k = (int)((n_list - 1) / LG_Q); k >= 0; k--) {
        const uint32_t base = range.x + (uint32_t)k * LG_Q;
        const uint32_t nbt = min((uint32_t)LG_Q, n_list - (uint32_t)k * LG_Q);
        const uint32_t nb = wmax > (uint32_t)k * LG_Q ? min((uint32_t)LG_Q, wmax - (uint32_t)k * LG_Q) : 0u;
        uint64_t hitmask = 0;
        if (nb > 0) {
            if (lane < nb) {
                const uint32_t id = point_list[base + lane];
                const float4 r0 = rec[3 * (size_t)id], r1 = rec[3 * (size_t)id + 1], r2 = rec[3 * (size_t)id + 2];
                uint32_t m = 0;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const float bx0 = tbx + (float)((s & 1) * 8), by0 = tby + (float)((s >> 1) * 8);
                    const bool hit = (r0.x + r2.y >= bx0) && (r0.x - r2.y <= bx0 + 7.0f) && (r0.y + r2.z >= by0) && (r0.y - r2.z <= by0 + 7.0f);
                    m |= (hit ? 1u : 0u) << s;
                }
                q0[lane] = r0; q1[lane] = r1; q2[lane] = make_float4(r2.x, r2.y, r2.z, __uint_as_float(m));
            }
#pragma unroll
            for (int c9 = 0; c9 < 9; c9++) stage[lane * 9 + c9] = 0.0f;
            __builtin_amdgcn_wave_barrier();

#pragma unroll
            for (int s = 0; s < 4; s++) {
                // A operand of the colour chain for this sub-block: g_i(pixel 4t + k), i < 3
                float Ag[16];
#pragma unroll
                for (int t = 0; t < 16; t++) Ag[t] = mi < 3u ? gT[(s * 3 + (int)mi) * 64 + 4 * t + (int)mk] : 0.0f;
                const float pxf = tbx + (float)((s & 1) * 8) + lxf, pyf = tby + (float)((s >> 1) * 8) + lyf;
                const float cxs = tbx + (float)((s & 1) * 8) + 3.5f, cys = tby + (float)((s >> 1) * 8) + 3.5f;
                uint32_t cnt = 0;        // pending entries of this sub-block (uniform)
                uint32_t myentry = 0;    // lane e < cnt: batch index of pending entry e
                int j = (int)nb - 1;
                while (true) {
                    // ---- evaluate entries back to front until 16 are pending or the batch is exhausted ----
                    for (; j >= 0 && cnt < 16u; j--) {
                        const float4 c = q2[j];
                        const uint32_t m = __builtin_amdgcn_readfirstlane(__float_as_uint(c.w));
                        if (!(m & (1u << s))) continue;
                        const float4 a = q0[j], b = q1[j];
                        const uint32_t rel = (uint32_t)k * LG_Q + (uint32_t)j + 1u;
                        float Wv, Dv;
                        const bool ok = bwd_pair_wd(a, b, c, rel <= last[s], pxf, pyf, T[s], Tfin[s], g0[s], g1[s], g2[s], bgd[s], a0[s], a1[s],
                                                    a2[s], Wv, Dv);
                        if (__ballot(ok) == 0) continue;
                        wbuf[cnt * LG_MF_STRIDE + lane] = Wv;
                        dbuf[cnt * LG_MF_STRIDE + lane] = Dv;
                        myentry = (lane == cnt) ? (uint32_t)j : myentry;
                        hitmask |= 1ull << j;
                        cnt++;
                    }
                    if (cnt == 0) break;
                    // ---- flush: sums over the 64 pixels for the pending entries on the matrix pipe ----
                    __builtin_amdgcn_wave_barrier();
                    lg_f4v accw = {0.0f, 0.0f, 0.0f, 0.0f}, accd = {0.0f, 0.0f, 0.0f, 0.0f};
                    const bool colok = mi < cnt;
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const float bw = colok ? wbuf[mi * LG_MF_STRIDE + 4 * t + mk] : 0.0f;
                        const float bd = colok ? dbuf[mi * LG_MF_STRIDE + 4 * t + mk] : 0.0f;
                        accw = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[t], bw, accw, 0, 0, 0);
                        accd = __builtin_amdgcn_mfma_f32_16x16x4f32(Ag[t], bd, accd, 0, 0, 0);
                    }
                    // lane l holds rows 4*(l/16)..+3 of column l%16: moments {1,u,v,uu} in lanes 0-15, {uv,vv} in lanes 16-31
                    const float Suv = __shfl(accw[0], (int)(lane + 16u) & 63), Svv = __shfl(accw[1], (int)(lane + 16u) & 63);
                    if (lane < cnt) {
                        const float4 ea = q0[myentry], eb = q1[myentry];
                        const float X = ea.x - cxs, Y = ea.y - cys;
                        const float S0 = accw[0], Su = accw[1], Sv = accw[2], Suu = accw[3];
                        const float Wdx = X * S0 - Su, Wdy = Y * S0 - Sv;
                        const float Wdxx = X * X * S0 - 2.0f * X * Su + Suu;
                        const float Wdxy = X * Y * S0 - X * Sv - Y * Su + Suv;
                        const float Wdyy = Y * Y * S0 - 2.0f * Y * Sv + Svv;
                        float* dst = stage + myentry * 9;
                        dst[0] += 2.0f * ea.z * Wdx + ea.w * Wdy;
                        dst[1] += 2.0f * eb.x * Wdy + ea.w * Wdx;
                        dst[2] += -0.5f * Wdxx;
                        dst[3] += -Wdxy;
                        dst[4] += -0.5f * Wdyy;
                        dst[5] += (eb.y != 0.0f) ? S0 / eb.y : 0.0f;
                        dst[6] += accd[0]; dst[7] += accd[1]; dst[8] += accd[2];
                    }
                    cnt = 0;
                    __builtin_amdgcn_wave_barrier();
                    if (j < 0) break;
                }
            }
        }
        if (lane < nbt) {
            float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = o0;
            if ((hitmask >> lane) & 1ull) {
                const float* src = stage + lane * 9;
                o0 = make_float4(src[0], src[1], src[2], src[3]);
                o1 = make_float4(src[4], src[5], src[6], src[7]);
                o2 = make_float4(src[8], 0.0f, 0.0f, 0.0f);
            }
            float4* dst = rows + 3 * (size_t)slot_sorted[base + lane];
            dst[0] = o0; dst[1] = o1; dst[2] = o2;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// diagnostics: wave_reduce9_to_lds on one wave (64 x 9 inputs -> 9 sums); used by tests/test_gpu_parity.py
__global__ void lg_debug_reduce9_kernel(const float* __restrict__ in, float* __restrict__ out)
{
    __shared__ float dst[9];
    float p[9];
    for (int c = 0; c < 9; c++) p[c] = in[threadIdx.x * 9 + c];
    wave_reduce9_to_lds(p, dst, threadIdx.x);
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    if (threadIdx.x < 9) out[threadIdx.x] = dst[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// K8 + K9 fused: per-Gaussian backward.  One wave per workgroup; SH rows in and dL/dSH rows out go
// through LDS so that global traffic is coalesced 16-byte accesses.
template <bool RAW>
__global__ void __launch_bounds__(LG_PP)
lg_preprocess_bwd(int N, int M, int D, int W, int H, float tanfovx, float tanfovy, float mod,
                  const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
                  const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest,
                  const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                  const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                  const int32_t* __restrict__ radii, const float4* __restrict__ aux, const uint32_t* __restrict__ touched,
                  const uint32_t* __restrict__ offsets, const float4* __restrict__ part,
                  float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dshs,
                  float* __restrict__ dL_dshs_rest, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
                  float* __restrict__ dL_dscales, float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D)
{
    __shared__ __attribute__((aligned(16))) float sh_rows[LG_PP * LG_SH_MAXF];
    const uint32_t lane = threadIdx.x;
    const int i0 = blockIdx.x * LG_PP;
    const int i = i0 + (int)lane;
    float vm[16], pm[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = viewmatrix[k]; pm[k] = projmatrix[k]; }
    cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
    const bool vis = (i < N) && radii[i] > 0;
    const uint64_t vmask = __ballot(vis);
    const bool split = RAW && shs_rest != nullptr;
    const int rowf = split ? 3 * (M - 1) : 3 * M;
    const int rows = min(LG_PP, N - i0);
    const bool use_sh = (shs != nullptr) && (dL_dshs != nullptr);
    if (use_sh && vmask && rowf > 0) {
        stage_sh_rows(split ? shs_rest : shs, i0, rows, rowf, vmask, sh_rows, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // Screen-filling splats own thousands of gradient rows; a single lane walking them would stall its wave for
    // milliseconds.  Such lanes are served one at a time by the whole wave: 64 rows per step, then a wave reduction.
    const uint32_t my_t = vis ? touched[i] : 0u;
    const uint32_t my_u0 = vis ? offsets[i] - my_t : 0u;
    float coop[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    {
        uint64_t big = __ballot(my_t > LG_COOP_ROWS);
        while (big) {
            const int src = (int)__builtin_ctzll(big);
            big &= big - 1;
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)my_t, src);
            const uint32_t u0 = (uint32_t)__builtin_amdgcn_readlane((int)my_u0, src);
            float acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t u = u0 + lane; u < u0 + t; u += LG_PP) {
                const float4* rp = part + 3 * (size_t)u;
                const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
                acc9[0] += v0.x; acc9[1] += v0.y; acc9[2] += v0.z; acc9[3] += v0.w; acc9[4] += v1.x; acc9[5] += v1.y; acc9[6] += v1.z;
                acc9[7] += v1.w; acc9[8] += v2.x;
            }
#pragma unroll
            for (int k9 = 0; k9 < 9; k9++) {
                const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(acc9[k9])), 63));
                if ((int)lane == src) coop[k9] = tot;
            }
        }
    }
    float m2[3] = {0, 0, 0}, m3[3] = {0, 0, 0}, dop = 0.0f, dsc[3] = {0, 0, 0}, drot[4] = {0, 0, 0, 0}, dcov[6] = {0, 0, 0, 0, 0, 0};
    float dcol[3] = {0, 0, 0};
    float dsh[LG_SH_MAXF];
#pragma unroll
    for (int k = 0; k < LG_SH_MAXF; k++) dsh[k] = 0.0f;
    if (vis) {
        // gather this Gaussian's gradient rows (one per tile instance) in slot order: deterministic, no atomics.
        // Splats with more than LG_COOP_ROWS instances were summed cooperatively by the whole wave (below).
        float a[9];
#pragma unroll
        for (int k9 = 0; k9 < 9; k9++) a[k9] = coop[k9];
        if (my_t <= LG_COOP_ROWS) {
            for (uint32_t u = my_u0; u < my_u0 + my_t; u++) {
                const float4* rp = part + 3 * (size_t)u;
                const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
                a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w; a[8] += v2.x;
            }
        }
        const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
        const float4 x0 = aux[2 * (size_t)i], x1 = aux[2 * (size_t)i + 1];
        float S[6] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y };
        LgGradOut go;
        lg_backward_geom(vm, pm, px, py, pz, S, a, W, H, tanfovx, tanfovy, go);
        m2[0] = go.mean2D[0]; m2[1] = go.mean2D[1];
        m3[0] = go.mean3D[0]; m3[1] = go.mean3D[1]; m3[2] = go.mean3D[2];
        dop = a[5];
        if (colors_precomp) {
            dcol[0] = a[6]; dcol[1] = a[7]; dcol[2] = a[8];
        } else if (use_sh) {
            const uint32_t cb = __float_as_uint(x1.z);
            float dRGB[3] = { (cb & 1u) ? 0.0f : a[6], (cb & 2u) ? 0.0f : a[7], (cb & 4u) ? 0.0f : a[8] };
            float sh[LG_SH_MAXF];
            const float* row = sh_rows + lane * rowf;
            const int nact = (D + 1) * (D + 1) * 3;
            if (split) {
                sh[0] = shs[3 * (size_t)i]; sh[1] = shs[3 * (size_t)i + 1]; sh[2] = shs[3 * (size_t)i + 2];
#pragma unroll
                for (int k = 3; k < LG_SH_MAXF; k++) sh[k] = (k < nact) ? row[k - 3] : 0.0f;
            } else if ((rowf & 3) == 0) {
#pragma unroll
                for (int q = 0; q < LG_SH_MAXF / 4; q++) {
                    float4 v4 = make_float4(0, 0, 0, 0);
                    if (q * 4 < nact) v4 = reinterpret_cast<const float4*>(row)[q];
                    sh[4 * q] = v4.x; sh[4 * q + 1] = v4.y; sh[4 * q + 2] = v4.z; sh[4 * q + 3] = v4.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < LG_SH_MAXF; k++) sh[k] = (k < nact) ? row[k] : 0.0f;
            }
            lg_backward_sh(D, sh, px, py, pz, cp, dRGB, m3, [&](int k, int c, float v) { dsh[k * 3 + c] = v; });
        }
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) dcov[k] = go.cov3D[k];
        } else {
            float sc[3] = { scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2] };
            const float4 q4 = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
            float q[4] = { q4.x, q4.y, q4.z, q4.w };
            float qn = 1.0f;
            if (RAW) {
                sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]);
                qn = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
                const float inv = 1.0f / qn;
                q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
            }
            lg_backward_cov3d(sc, mod, q, go.cov3D, dsc, drot);
            if (RAW) {
                // exp: d/draw = d/ds * s ; normalize: d/dr = (g - q (q.g)) / |r|
                dsc[0] *= sc[0]; dsc[1] *= sc[1]; dsc[2] *= sc[2];
                const float qg = q[0] * drot[0] + q[1] * drot[1] + q[2] * drot[2] + q[3] * drot[3];
                const float inv = 1.0f / qn;
#pragma unroll
                for (int k = 0; k < 4; k++) drot[k] = (drot[k] - q[k] * qg) * inv;
            }
        }
        if (RAW) { // sigmoid: d/dlogit = d/dsigma * sigma (1 - sigma)
            const float sg = lg_sigmoid(opacities[i]);
            dop = dop * sg * (1.0f - sg);
        }
    }
    if (use_sh) {
        // every lane has read its input row: reuse the LDS rows for the gradient rows, then store coalesced
        __builtin_amdgcn_wave_barrier();
        float* row = sh_rows + lane * rowf;
        if (split) {
            if (i < N) { dL_dshs[3 * (size_t)i] = dsh[0]; dL_dshs[3 * (size_t)i + 1] = dsh[1]; dL_dshs[3 * (size_t)i + 2] = dsh[2]; }
#pragma unroll
            for (int k = 3; k < LG_SH_MAXF; k++)
                if (k - 3 < rowf) row[k - 3] = dsh[k];
        } else if ((rowf & 3) == 0) {
#pragma unroll
            for (int q = 0; q < LG_SH_MAXF / 4; q++)
                if (q * 4 < rowf) reinterpret_cast<float4*>(row)[q] = make_float4(dsh[4 * q], dsh[4 * q + 1], dsh[4 * q + 2], dsh[4 * q + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < LG_SH_MAXF; k++)
                if (k < rowf) row[k] = dsh[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* dst = (split ? dL_dshs_rest : dL_dshs) + (size_t)i0 * rowf;
        const int nfl = rows * rowf;
        if (nfl > 0) {
            if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
                const int nvec = nfl >> 2;
                for (int q = (int)lane; q < nvec; q += LG_PP) reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(sh_rows)[q];
                for (int f = (nvec << 2) + (int)lane; f < nfl; f += LG_PP) dst[f] = sh_rows[f];
            } else {
                for (int f = (int)lane; f < nfl; f += LG_PP) dst[f] = sh_rows[f];
            }
        }
    }
    if (i >= N) return;
    dL_dmeans2D[3 * (size_t)i] = m2[0]; dL_dmeans2D[3 * (size_t)i + 1] = m2[1]; dL_dmeans2D[3 * (size_t)i + 2] = 0.0f;
    dL_dmeans3D[3 * (size_t)i] = m3[0]; dL_dmeans3D[3 * (size_t)i + 1] = m3[1]; dL_dmeans3D[3 * (size_t)i + 2] = m3[2];
    dL_dopacity[i] = dop;
    if (dL_dcolors) { dL_dcolors[3 * (size_t)i] = dcol[0]; dL_dcolors[3 * (size_t)i + 1] = dcol[1]; dL_dcolors[3 * (size_t)i + 2] = dcol[2]; }
    if (dL_dscales) { dL_dscales[3 * (size_t)i] = dsc[0]; dL_dscales[3 * (size_t)i + 1] = dsc[1]; dL_dscales[3 * (size_t)i + 2] = dsc[2]; }
    if (dL_drots) *reinterpret_cast<float4*>(dL_drots + 4 * (size_t)i) = make_float4(drot[0], drot[1], drot[2], drot[3]);
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = dcov[k];
    }
}

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
    const int ntiles_pad8 = (ntiles + 7) / 8 * 8;
    GeomView geo = carve_geom(geom_p, N);
    ImgView img = carve_img(img_p, W, H);
    const size_t HW = (size_t)W * H;

    if (binning_out) *binning_out = nullptr;
    if (num_rendered) *num_rendered = 0;
    uint32_t h_counters[3] = {0, 0, 0}, h_R = 0;
    if (N > 0) {
        HIP_TRY(hipMemsetAsync(geo.counters, 0, 64, stream));
        {
            ProfScope ps(prof, "preprocess", stream);
#define LAUNCH_PP(RAWP)                                                                                                              \
    lg_preprocess<RAWP><<<(N + LG_PP - 1) / LG_PP, LG_PP, 0, stream>>>(N, g->M, v->sh_degree, W, H, v->tanfovx, v->tanfovy,           \
                                                                      v->scale_modifier, v->prefiltered, v->viewmatrix, v->projmatrix, \
                                                                      v->campos, g->means3D, g->shs, g->shs_rest, g->colors_precomp,   \
                                                                      g->opacities, g->scales, g->rotations, g->cov3D_precomp, geo, out_radii)
            if (v->flags & LG_FLAG_RAW_PARAMS) LAUNCH_PP(true); else LAUNCH_PP(false);
#undef LAUNCH_PP
        }
        KCHECK("lg_preprocess");
        lg_reduce_dmax<<<1, 1024, 0, stream>>>((N + LG_PP - 1) / LG_PP, geo.blk_dmax, geo.counters);
        KCHECK("lg_reduce_dmax");
        {
            ProfScope ps(prof, "scan", stream);
            size_t tb = geo.scan_temp_bytes;
            HIP_TRY(hipcub::DeviceScan::InclusiveSum(geo.scan_temp, tb, geo.touched, geo.offsets, N, stream));
        }
        HIP_TRY(hipMemcpyAsync(&h_R, geo.offsets + (N - 1), 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(h_counters, geo.counters, 12, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (v->prefiltered && h_counters[1]) return fail(LG_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
    }
    const int64_t R = h_R;
    g_stats.num_rendered = R;
    g_stats.num_visible = -1; // not tracked on the device (see lg_preprocess); callers count radii > 0
    if (num_rendered) *num_rendered = R;

    // key format: packed single-u64 keys when tile | depth | id fit 64 bits (they do for every BASELINE config)
    const int tile_bits = bits_for((uint32_t)ntiles), gid_bits = bits_for((uint32_t)(N > 1 ? N : 2));
    const uint32_t dspan = h_counters[2] > LG_DEPTH_BIAS ? h_counters[2] - LG_DEPTH_BIAS : 0u;
    const int depth_bits = bits_for(dspan + 1u) > 0 ? bits_for(dspan + 1u) : 1;
    const bool packed = (tile_bits + depth_bits + gid_bits <= 64) && (getenv("LG_FORCE_PAIR_SORT") == nullptr);

    void* bin_p = alloc(alloc_user, carve_bin(nullptr, R, W, H, packed).total);
    if (!bin_p) return fail(LG_ERR_ALLOC, "binning allocator returned NULL");
    if (binning_out) *binning_out = bin_p;
    BinView bin = carve_bin(bin_p, R, W, H, packed);
    HIP_TRY(hipMemsetAsync(bin.ranges, 0, (size_t)ntiles * 8, stream));
    const uint32_t* point_list = bin.point_list;
    if (R > 0) {
        {
            ProfScope ps(prof, "duplicate", stream);
            if (packed)
                lg_duplicate<true><<<(N + 255) / 256, 256, 0, stream>>>(N, gx, depth_bits, gid_bits, geo.touched, geo.offsets, geo.tinfo,
                                                                        bin.keys_in, nullptr, nullptr);
            else
                lg_duplicate<false><<<(N + 255) / 256, 256, 0, stream>>>(N, gx, 0, 0, geo.touched, geo.offsets, geo.tinfo, bin.keys_in,
                                                                         bin.slot_in, bin.gid_slot);
        }
        KCHECK("lg_duplicate");
        {
            ProfScope ps(prof, "sort", stream);
            size_t tb = bin.sort_temp_bytes;
            if (packed)
                HIP_TRY(hipcub::DeviceRadixSort::SortKeys(bin.sort_temp, tb, bin.keys_in, bin.keys_out, (int)R, gid_bits,
                                                          gid_bits + depth_bits + tile_bits, stream));
            else
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(bin.sort_temp, tb, bin.keys_in, bin.keys_out, bin.slot_in, bin.slot_out, (int)R, 0,
                                                           32 + tile_bits, stream));
        }
        {
            ProfScope ps(prof, "finalize_bins", stream);
            if (packed)
                lg_finalize_bins<true><<<(uint32_t)((R + 255) / 256), 256, 0, stream>>>((uint32_t)R, gx, depth_bits, gid_bits, bin.keys_out,
                                                                                        geo.tinfo, nullptr, nullptr, bin.point_list,
                                                                                        bin.slot_out, bin.ranges);
            else
                lg_finalize_bins<false><<<(uint32_t)((R + 255) / 256), 256, 0, stream>>>((uint32_t)R, gx, 0, 0, bin.keys_out, geo.tinfo,
                                                                                         bin.slot_out, bin.gid_slot, bin.point_list,
                                                                                         bin.slot_out, bin.ranges);
        }
        KCHECK("lg_finalize_bins");
    }
    if (count && N > 0) {
        HIP_TRY(hipMemsetAsync(out_count, 0, (size_t)N * 4, stream));
        HIP_TRY(hipMemsetAsync(out_score, 0, (size_t)N * 4, stream));
    }
    {
        ProfScope ps(prof, count ? "blend_fwd_count" : "blend_fwd", stream);
        dim3 grid(ntiles_pad8), block(256);
#define LAUNCH_FWD(CNT, FS, EX)                                                                                                      \
    lg_blend_fwd<CNT, FS, EX><<<grid, block, 0, stream>>>(W, H, gx, ntiles, ntiles_pad8, bin.ranges, point_list, geo.rec, v->bg,     \
                                                         out_color, img.final_T, img.n_contrib, out_count, out_score, weight_policy)
        const bool fs = count && (weight_policy == LG_WEIGHT_ALPHA || weight_policy == LG_WEIGHT_ALPHA_T);
        if (!count) { if (fast) LAUNCH_FWD(false, false, false); else LAUNCH_FWD(false, false, true); }
        else if (!fs) { if (fast) LAUNCH_FWD(true, false, false); else LAUNCH_FWD(true, false, true); }
        else { if (fast) LAUNCH_FWD(true, true, false); else LAUNCH_FWD(true, true, true); }
#undef LAUNCH_FWD
    }
    KCHECK("lg_blend_fwd");
    (void)HW;
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
    const int ntiles_pad8 = (ntiles + 7) / 8 * 8;
    GeomView geo = carve_geom(const_cast<void*>(geom_p), N);
    ImgView img = carve_img(const_cast<void*>(img_p), W, H);
    BinView bin = carve_bin(const_cast<void*>(bin_p), R, W, H, true); // only the format-independent prefix is used
    float* acc = (float*)scratch; // [R][12] gradient rows, every row written by lg_blend_bwd
    if (R > 0) {
        ProfScope ps(prof, "blend_bwd", stream);
        const char* abl_s = getenv("LG_ABLATE");
        const int abl = abl_s ? atoi(abl_s) : 0;
#define LAUNCH_BWD(EX, AB) lg_blend_bwd<EX, AB><<<ntiles_pad8, 64, 0, stream>>>(W, H, gx, ntiles, ntiles_pad8, bin.ranges, bin.point_list, bin.slot_out, geo.rec, v->bg, \
                                                                 img.final_T, img.n_contrib, dL_dcolor, acc)
        if (fast && abl == 1) LAUNCH_BWD(false, 1);
        else if (fast && abl == 2) LAUNCH_BWD(false, 2);
        else if (fast && abl == 3) LAUNCH_BWD(false, 3);
        else if (fast && abl == 4) LAUNCH_BWD(false, 4);
        else if (fast && getenv("LG_BWD_MFMA") != nullptr) // experiment, off by default: correct but 1.8x slower
            lg_blend_bwd_mfma<<<ntiles_pad8, 64, 0, stream>>>(W, H, gx, ntiles, ntiles_pad8, bin.ranges, bin.point_list, bin.slot_out, geo.rec,
                                                            v->bg, img.final_T, img.n_contrib, dL_dcolor, acc);
        else if (fast)
            lg_blend_bwd<false><<<ntiles_pad8, 64, 0, stream>>>(W, H, gx, ntiles, ntiles_pad8, bin.ranges, bin.point_list, bin.slot_out, geo.rec, v->bg,
                                                                 img.final_T, img.n_contrib, dL_dcolor, acc);
        else
            lg_blend_bwd<true><<<ntiles_pad8, 64, 0, stream>>>(W, H, gx, ntiles, ntiles_pad8, bin.ranges, bin.point_list, bin.slot_out, geo.rec, v->bg,
                                                                img.final_T, img.n_contrib, dL_dcolor, acc);
    }
    KCHECK("lg_blend_bwd");
    {
        ProfScope ps(prof, "preprocess_bwd", stream);
#define LAUNCH_PPB(RAWP)                                                                                                             \
    lg_preprocess_bwd<RAWP><<<(N + LG_PP - 1) / LG_PP, LG_PP, 0, stream>>>(                                                           \
        N, g->M, v->sh_degree, W, H, v->tanfovx, v->tanfovy, v->scale_modifier, v->viewmatrix, v->projmatrix, v->campos, g->means3D,  \
        g->shs, g->shs_rest, g->colors_precomp, g->opacities, g->scales, g->rotations, g->cov3D_precomp, radii, geo.aux, geo.touched,  \
        geo.offsets, reinterpret_cast<const float4*>(acc), dL_dmeans2D, dL_dmeans3D, dL_dshs, dL_dshs_rest, dL_dcolors, dL_dopacity,   \
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
