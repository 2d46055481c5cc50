// lg_preprocess.h -- per-Gaussian kernels: K1 lg_preprocess (project / EWA / SH->RGB / culling) and K8+K9 lg_preprocess_bwd
// Part of liblightgaussian_hip.so (single translation unit: lg_api.hip includes the lg_*.h kernel headers).
#pragma once

#include "lg_host.h"
#include "lg_wave.h"

// ------------------------------------------------------------------------------------------------
// K1: preprocess.  One wave per workgroup, 64 consecutive Gaussians.  Default (DIRECT): every visible lane reads its own SH row with
// dword-aligned 16-byte loads at the row stride (twelve pieces at degree 3; measured faster than staging).
// LG_FLAG_K1_LDS (cross-check, option k1_lds) selects the round-1 form: the wave's rows (64 x 12M bytes, contiguous in memory) through
// coalesced 16-byte loads into LDS, skipping rows of culled Gaussians.
#define LG_PP 64
#define LG_ID_BITS 29            // blend record, last word: Gaussian id | SH clamp flags << 29
#define LG_ID_MASK ((1u << LG_ID_BITS) - 1u)
#define LG_SH_MAXF 48 // floats per SH row at M = 16
#ifndef LG_K1_PAD_LDS
// A TUNING CONSTANT, not a resource the kernel uses (r4 verdict, weak #6): 160 KB of LDS per CU / (5.4 KB static + 7.9 KB of this padding)
// = 12 workgroups (waves) per CU, where K1's 92 VGPRs alone would allow 20.  It is a side-effect knob -- it also keeps OTHER kernels' waves
// off the CU while K1 runs (which is why the host drops it where K1 reads no SH rows, lg_api.hip) -- and it papers over the real cause
// (twelve 16-byte pieces of one cache line fetched at a 180-byte stride by twenty waves).  Tied to gfx950's 160 KB LDS, to the static LDS
// above and to K1's register count: re-measure (tools/k1_cfg.sh sweeps it) whenever any of the three changes.  Round 5 re-measured nothing
// here: K1 is unchanged (0.189 ms under rocprofv3, r04 and r05 profiles alike).
#define LG_K1_PAD_LDS 7900 // dynamic LDS the host adds to K1's 5.4 KB per wave: 12 waves per CU (lg_api.hip, at the launch)
#endif
#ifndef LG_K9_GATHER
#define LG_K9_GATHER 4 // K9: gradient rows a lane requests per round trip
#endif
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
// SH rows are 3M (or 3(M-1)) floats, i.e. in general only 4-byte aligned (180 B for _features_rest at M = 16).  gfx950
// global loads of 16 bytes need dword alignment only, so a lane reads its row with dwordx4 loads through a packed,
// 4-byte-aligned float4 plus at most three trailing dwords -- no LDS staging, no alignment precondition.
struct __attribute__((packed, aligned(4))) lg_f4u { float x, y, z, w; };
template <int K0>
__device__ __forceinline__ void read_row_direct(const float* __restrict__ row, int nfl, float* sh /*[LG_SH_MAXF]*/)
{
#pragma unroll
    for (int q = 0; q < (LG_SH_MAXF - K0 + 3) / 4; q++) {
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (4 * q + 4 <= nfl) {
            const lg_f4u t = reinterpret_cast<const lg_f4u*>(row)[q];
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (4 * q + c < nfl) v[c] = row[4 * q + c];
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (K0 + 4 * q + c < LG_SH_MAXF) sh[K0 + 4 * q + c] = v[c];
    }
}
__device__ __forceinline__ float lg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// 92 VGPRs -> 5 waves/SIMD.  Forcing 6 or 8 (amdgpu_waves_per_eu) spills 43 / 71 registers: measured 0.22 -> 0.28 / 0.45 ms.
template <bool RAW, bool DIRECT>
#ifdef LG_K1_WAVES
__global__ void __launch_bounds__(LG_PP) __attribute__((amdgpu_waves_per_eu(LG_K1_WAVES, 8)))
#else
__global__ void __launch_bounds__(LG_PP)
#endif
lg_preprocess(int N, int M, int D, int W, int H, float tanfovx, float tanfovy, float mod, int prefiltered, int skip_color,
              const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
              const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest,
              const float* __restrict__ colors_precomp,
              const float* __restrict__ opacities, const float* __restrict__ scales, const float* __restrict__ rotations,
              const float* __restrict__ cov3D_precomp, GeomView g, int32_t* __restrict__ radii, int32_t* __restrict__ zero_count,
              float* __restrict__ zero_score, uint32_t* __restrict__ clear_words, uint32_t n_clear, int save_jac)
{
    // DIRECT (default): every visible lane reads its own SH row with dwordx4 loads (read_row_direct) and no LDS is
    // allocated for SH (occupancy is then register-limited, 5 waves/SIMD, instead of LDS-limited, 3).  The LDS-staged
    // variant is kept behind the LG_K1_LDS environment switch as the cross-check of the direct reads.
    __shared__ __attribute__((aligned(16))) float sh_rows[DIRECT ? 4 : LG_PP * LG_SH_MAXF];
    __shared__ float4 st_rec[LG_PP * 3]; // records leave through LDS as coalesced 16-byte stores
    __shared__ __attribute__((aligned(16))) float st_jac[LG_PP * 9];   // ... and so do the SH direction Jacobians (save_jac)
    static_assert(LG_REC_F4 == 3, "coalesced record store assumes packed 48-byte records");
    const uint32_t lane = threadIdx.x;
    const int i0 = blockIdx.x * LG_PP;
    const int i = i0 + (int)lane;
    // capacity-bounded forward: the binning buffer exists before this kernel runs, so the clear of the radix sort's digit
    // histograms / tickets / look-back states (~2.6 MB at C3) rides here, a few words per workgroup, instead of in a launch of
    // its own (lg_zero_words: 4.7 us, the duration of an empty launch on MI355X)
    for (uint32_t w = blockIdx.x * LG_PP + lane; w < n_clear; w += gridDim.x * LG_PP) clear_words[w] = 0u;
    float vm[16], pm[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = viewmatrix[k]; pm[k] = projmatrix[k]; }
    cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
    bool vis = false, violation = false;
    float px = 0, py = 0, pz = 0, op = 0;
    float cov[6] = {0, 0, 0, 0, 0, 0};
    LgSplat sp;
    if (i < N) {
        px = means3D[3 * (size_t)i]; py = means3D[3 * (size_t)i + 1]; pz = means3D[3 * (size_t)i + 2];
        // near-plane test first so culled Gaussians cost 12 bytes of reads (hoisted loads, an early SH request: EXPERIMENTS.md, "K1 / K9")
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
                    // F.normalize = x / max(|x|, eps), evaluated EXACTLY as torch does on this GPU: squares summed pairwise,
                    // (a^2 + b^2) + (c^2 + d^2), and a DIVISION per component (tools/activation_probe.py: 0 of 12 M components differ;
                    // left-to-right summation differs in 10 % of them, a multiplication by the reciprocal in 26 %).  expf and
                    // 1 / (1 + expf(-x)) equal torch.exp / torch.sigmoid bit for bit as they stand.  One differing bit can move
                    // ceil(3 sigma) -- the radius -- by one and with it a border tile: 2e-3 in the image of 3 of 500 fuzz scenes.
                    const float qn = fmaxf(sqrtf((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3])), 1e-12f);
                    q[0] /= qn; q[1] /= qn; q[2] /= qn; q[3] /= qn;
                }
                lg_cov3d(sc, mod, q, cov);
            }
            op = RAW ? lg_sigmoid(opacities[i]) : opacities[i];
            vis = lg_project(vm, pm, px, py, pz, cov, op, W, H, tanfovx, tanfovy, sp);
        } else if (prefiltered) {
            violation = true;   // "Point is filtered although prefiltered is set": reported through the per-workgroup word below
        }
        // the count variant accumulates into these with atomics from the blend kernel: cleared here instead of by two memsets
        if (zero_count) { zero_count[i] = 0; zero_score[i] = 0.0f; }
    }
    const uint64_t vmask = __ballot(vis);
    const bool split = RAW && shs_rest != nullptr;        // dc and rest are separate tensors
    const int rowf = split ? 3 * (M - 1) : 3 * M;          // floats per LDS-staged row
    if (!DIRECT && !skip_color && shs && vmask && rowf > 0) {
        stage_sh_rows(split ? shs_rest : shs, i0, min(LG_PP, N - i0), rowf, vmask, sh_rows, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t touched = 0;
    if (i < N) {
        int radius = 0;
        if (vis) {
            radius = sp.radius;
            touched = (uint32_t)((sp.tx1 - sp.tx0) * (sp.ty1 - sp.ty0));
            float rgb[3] = {0.0f, 0.0f, 0.0f};
            uint32_t cb = 0;
            if (skip_color) {
                // significance-only pass (LG_FLAG_SKIP_COLOR): the image is not wanted, so the SH rows are never read
            } else if (colors_precomp) {
                rgb[0] = colors_precomp[3 * (size_t)i]; rgb[1] = colors_precomp[3 * (size_t)i + 1]; rgb[2] = colors_precomp[3 * (size_t)i + 2];
            } else {
                float sh[LG_SH_MAXF];
                const float* row = DIRECT ? shs + (size_t)i * rowf : sh_rows + lane * rowf;
                const int nact = (D + 1) * (D + 1) * 3;
                if (split && DIRECT) {
                    sh[0] = shs[3 * (size_t)i]; sh[1] = shs[3 * (size_t)i + 1]; sh[2] = shs[3 * (size_t)i + 2];
                    read_row_direct<3>(shs_rest + (size_t)i * rowf, nact - 3, sh);
                } else if (DIRECT) {
                    read_row_direct<0>(row, nact, sh);
                } else if (split) {
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
                if (save_jac) {
                    float J[9];
                    lg_sh_dir_jacobian(D, sh, px, py, pz, cp, J);
#pragma unroll
                    for (int k = 0; k < 9; k++) st_jac[9 * lane + k] = J[k];
                }
            }
            st_rec[3 * lane + 0] = make_float4(sp.x, sp.y, sp.ha, sp.nb);
            st_rec[3 * lane + 1] = make_float4(sp.hc, op, rgb[0], rgb[1]);
            // last word: Gaussian id (29 bits) | SH clamp flags (3 bits, for K9).  No separate "backward record": K9 recomputes the
            // 3D covariance from the scales / rotation it reads anyway (the 32-byte aux rows of round 1 were 10 % of K1's traffic)
            st_rec[3 * lane + 2] = make_float4(rgb[2], sp.hx, sp.hy, __uint_as_float((uint32_t)i | (cb << LG_ID_BITS)));
            g.tinfo[i] = make_uint4((uint32_t)sp.tx0 | ((uint32_t)sp.ty0 << 16), (uint32_t)sp.tx1 | ((uint32_t)sp.ty1 << 16),
                                    __float_as_uint(sp.depth), 0u);
        }
        radii[i] = radius;
        g.visible[i] = radius > 0 ? (uint8_t)1 : (uint8_t)0;
        g.touched[i] = touched;
    }
    // The records of the workgroup's 64 Gaussians are contiguous in rec: staged in LDS and written as coalesced
    // 16-byte stores (per-lane 48-byte-stride stores measured 0.26 -> 0.22 ms for the whole kernel).  Entries of
    // invisible Gaussians carry stale LDS contents; nothing reads them (touched == 0).
    if (vmask) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nrec = min(LG_PP, N - i0);
#pragma unroll
        for (int k = 0; k < 3; k++)
            if ((int)(k * LG_PP + lane) < 3 * nrec) g.rec[3 * (size_t)i0 + k * LG_PP + lane] = st_rec[k * LG_PP + lane];
        if (save_jac && shs && !skip_color) {
            // 36 bytes per Gaussian, contiguous for the workgroup (i0 is a multiple of 64: 16-byte aligned); rows of culled Gaussians
            // carry stale LDS contents, nothing reads them
            float* dstj = g.shjac + 9 * (size_t)i0;
            const int nfl = 9 * nrec, nvec = nfl >> 2;
            for (int q = (int)lane; q < nvec; q += LG_PP) reinterpret_cast<float4*>(dstj)[q] = reinterpret_cast<const float4*>(st_jac)[q];
            for (int f = (nvec << 2) + (int)lane; f < nfl; f += LG_PP) dstj[f] = st_jac[f];
        }
    }
    // (no global visible-counter: 47k same-address atomics serialise at ~11 ns each -- more than the whole kernel)
    // largest depth of the workgroup (bit pattern; positive floats order like integers), for the packed sort key.
    // Written per workgroup and reduced by a one-block kernel: a shared atomicMax serialises the first ~3k waves.
    uint32_t dmax = vis ? __float_as_uint(sp.depth) : 0u;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, sh));
    // bit 31 (never set in the bit pattern of a positive depth): a prefiltered violation in this workgroup
    // ... and its instance count: lg_scan_blocks scans these words (one per 64 Gaussians) instead of a device-wide scan of N
    uint32_t tsum = touched;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) tsum += (uint32_t)__shfl_xor((int)tsum, sh);
    const uint64_t any_violation = __ballot(violation);   // evaluated by the whole wave, not under the lane-0 branch below
    if (lane == 0) {
        g.blk_dmax[blockIdx.x] = dmax | (any_violation ? 0x80000000u : 0u);
        g.blk_sum[blockIdx.x] = tsum;
        if (blockIdx.x == 0) {
            g.counters[8] = 0u;   // arrival counter of lg_scan_blocks (the scratch buffer arrives uninitialised)
            g.counters[9] = (save_jac && shs && !skip_color) ? LG_SHJAC_MAGIC : 0u;   // K9: the Jacobian rows of this view exist
        }
    }
}


// ------------------------------------------------------------------------------------------------
// K8 + K9 fused: per-Gaussian backward.  One wave per workgroup; SH rows in and dL/dSH rows out go
// through LDS so that global traffic is coalesced 16-byte accesses.
// Registers: JAC = false (the SH coefficients are re-read, sh[48] lives in registers) 155 VGPRs -> 3 waves/SIMD; forcing 4
// (-DLG_K9_WAVES=4: 128 VGPRs, 76 B/lane of scratch) was measured in round 3: 0.384 -> 0.594 ms -- the spills cost more than the
// fourth wave hides.  JAC = true (the default of every differentiated render since round 4: no coefficients, no input staging) 107 VGPRs
// -> 4 waves/SIMD.  Reading the SH rows directly per lane as K1 does (JAC = false only) was measured: 0.380 vs 0.380 ms.
template <bool RAW, bool JAC>
#ifdef LG_K9_WAVES
__global__ void __launch_bounds__(LG_PP) __attribute__((amdgpu_waves_per_eu(LG_K9_WAVES, 8)))
#else
__global__ void __launch_bounds__(LG_PP)
#endif
lg_preprocess_bwd(int N, int first_blk, int M, int D, int W, int H, float tanfovx, float tanfovy, float mod,
                  const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
                  const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest,
                  const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                  const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                  const int32_t* __restrict__ radii, const float4* __restrict__ rec,
                  const uint32_t* __restrict__ counters, const uint32_t* __restrict__ meta, uint32_t S, const uint32_t* __restrict__ touched,
                  const uint32_t* __restrict__ offsets, const float4* __restrict__ part, const float* __restrict__ shjac,
                  float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dshs,
                  float* __restrict__ dL_dshs_rest, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
                  float* __restrict__ dL_dscales, float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D)
{
    __shared__ __attribute__((aligned(16))) float sh_rows[LG_PP * LG_SH_MAXF];
    const uint32_t lane = threadIdx.x;
    const int i0 = ((int)blockIdx.x + first_blk) * LG_PP;   // (chunked backward: this launch covers workgroups first_blk ...)
    const int i = i0 + (int)lane;
    float vm[16], pm[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = viewmatrix[k]; pm[k] = projmatrix[k]; }
    cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
    // counters[0] != 0: the forward aborted this view on the device (lg_forward_bounded overflow); there are no rows.
    // meta[2] != S: this backward was given another segment length than the forward that filled the buffers (lg_view.segment_length
    // must match): lg_blend_bwd refused to run, there are no rows either -- zero gradients, and LG_FLAG_DEBUG reports it
    const bool vis = (i < N) && radii[i] > 0 && counters[0] == 0u && meta[2] == S && (!JAC || counters[9] == LG_SHJAC_MAGIC);
    const uint64_t vmask = __ballot(vis);
    const bool split = RAW && shs_rest != nullptr;
    const int rowf = split ? 3 * (M - 1) : 3 * M;
    const int rows = min(LG_PP, N - i0);
    // rgb_only (round 5, data-parallel steps): SH inputs, but the caller asked for dL/d(rgb) per Gaussian [N,3] (post clamp mask, through
    // dL_dcolors) INSTEAD of the (M, 3) coefficient gradients -- the SH gradient of one view is the outer product basis(dir) x dRGB, which
    // every rank can rebuild from 12 bytes per Gaussian and the view's camera centre (lg_sh_grad_from_rgb below): 192 bytes per Gaussian
    // less to write here and to put on the wire.  The view-direction term of dL/dmeans3D is computed as always.
    const bool rgb_only = (shs != nullptr) && (dL_dshs == nullptr) && (dL_dcolors != nullptr);
    const bool use_sh = (shs != nullptr) && ((dL_dshs != nullptr) || rgb_only);
    // JAC: the forward left d rgb / d direction of every visible Gaussian (LG_FLAG_SAVE_SH_JACOBIAN; the host instantiates this variant
    // when the view carries the flag): the coefficients are not read at all.  The marker word says the rows of THIS view are there; a
    // backward handed the flag after a forward without it finds no marker and writes zero gradients (LG_FLAG_DEBUG reports it), like
    // a backward with another segment length.
    if (use_sh && !JAC && vmask && rowf > 0) {
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
        float mo[9];
#pragma unroll
        for (int k9 = 0; k9 < 9; k9++) mo[k9] = coop[k9];
        if (my_t <= LG_COOP_ROWS) {
            // LG_K9_GATHER rows per round trip (round 4): the row-by-row loop waited for every row before it asked for the next one, and a
            // wave runs as many rounds as its busiest lane has rows -- 4 to 9 dependent round trips.  The loads of a round are issued
            // together (rows past the lane's last one re-read it and are not added); same additions in the same order, so the gradients
            // are bit-identical.  K9 0.262 -> 0.249 ms at C3 with 3 or 4 rows per round (2: no change), A/B on one box.
            // (Requesting the record, the parameters and the Jacobian row in front of the rows as well: 0.233 vs 0.232-0.240 ms, nothing.)
            const uint32_t ue = my_u0 + my_t;
            for (uint32_t u = my_u0; u < ue; u += LG_K9_GATHER) {
                float4 a[LG_K9_GATHER][3];
#pragma unroll
                for (int j = 0; j < LG_K9_GATHER; j++) {
                    const float4* rp = part + 3 * (size_t)min(u + (uint32_t)j, ue - 1u);
                    a[j][0] = rp[0]; a[j][1] = rp[1]; a[j][2] = rp[2];
                }
#pragma unroll
                for (int j = 0; j < LG_K9_GATHER; j++) {
                    if (u + (uint32_t)j < ue) {
                        mo[0] += a[j][0].x; mo[1] += a[j][0].y; mo[2] += a[j][0].z; mo[3] += a[j][0].w; mo[4] += a[j][1].x; mo[5] += a[j][1].y;
                        mo[6] += a[j][1].z; mo[7] += a[j][1].w; mo[8] += a[j][2].x;
                    }
                }
            }
        }
        // the rows are pixel-offset moments (lg_blend.h): finish them with this Gaussian's conic and opacity, exactly the
        // values the blend kernels used (its blend record)
        const float4 q0 = rec[LG_REC_F4 * (size_t)i], q1 = rec[LG_REC_F4 * (size_t)i + 1], q2 = rec[LG_REC_F4 * (size_t)i + 2];
        const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
        float a[9];
        lg_rows_to_grads(mo, q0.z, q0.w, q1.x, q1.y, a);
        // 3D covariance: the precomputed input, or recomputed from the (activated) scales / rotation exactly as K1 did
        float S[6], sc[3] = {0, 0, 0}, q[4] = {0, 0, 0, 0}, qn = 1.0f;
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) S[k] = cov3D_precomp[6 * (size_t)i + k];
        } else {
            sc[0] = scales[3 * (size_t)i]; sc[1] = scales[3 * (size_t)i + 1]; sc[2] = scales[3 * (size_t)i + 2];
            const float4 q4 = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
            q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
            if (RAW) {
                sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]);
                qn = fmaxf(sqrtf((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3])), 1e-12f);
                q[0] /= qn; q[1] /= qn; q[2] /= qn; q[3] /= qn;     // exactly K1's operations (pairwise sum, division: see there)
            }
            lg_cov3d(sc, mod, q, S);
        }
        LgGradOut go;
        lg_backward_geom(vm, pm, px, py, pz, S, a, W, H, tanfovx, tanfovy, go);
        m2[0] = go.mean2D[0]; m2[1] = go.mean2D[1];
        m3[0] = go.mean3D[0]; m3[1] = go.mean3D[1]; m3[2] = go.mean3D[2];
        dop = a[5];
        if (colors_precomp) {
            dcol[0] = a[6]; dcol[1] = a[7]; dcol[2] = a[8];
        } else if (use_sh) {
            const uint32_t cb = __float_as_uint(q2.w) >> LG_ID_BITS;
            float dRGB[3] = { (cb & 1u) ? 0.0f : a[6], (cb & 2u) ? 0.0f : a[7], (cb & 4u) ? 0.0f : a[8] };
            if (rgb_only) { dcol[0] = dRGB[0]; dcol[1] = dRGB[1]; dcol[2] = dRGB[2]; }
            if (JAC) {
                const float* jr = shjac + 9 * (size_t)i;          // 36-byte rows: dword-aligned 16-byte loads
                const lg_f4u j0 = reinterpret_cast<const lg_f4u*>(jr)[0], j1 = reinterpret_cast<const lg_f4u*>(jr)[1];
                const float J[9] = { j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w, jr[8] };
                lg_backward_sh_jac(D, J, px, py, pz, cp, dRGB, m3, [&](int k, int c, float v) { dsh[k * 3 + c] = v; });
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
            lg_backward_sh(D, sh, px, py, pz, cp, dRGB, m3, [&](int k, int c, float v) { dsh[k * 3 + c] = v; });
            }
        }
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) dcov[k] = go.cov3D[k];
        } else {
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
    if (use_sh && !rgb_only) {
        // every lane has read its input row: reuse the LDS rows for the gradient rows, then store coalesced.  (HALVES = 2 sends the
        // 64 rows out in two halves through half the LDS -- 6 KB per wave, four waves per SIMD with the 107 VGPRs of the JAC variant
        // instead of three: measured in round 4, 0.2481 / 0.2451 vs 0.2476 / 0.2481 ms, nothing -- the kernel moves 1.25 GB at 5.1 TB/s.
        // Every lane storing its own row with 16-byte stores at the row stride, no LDS at all: 0.26 -> 0.79 ms -- partial-line WRITES are
        // what the staging is for; strided 16-byte READS, K1's way, are fine.)
        constexpr int HALVES = 1, HROWS = LG_PP / HALVES;
        __builtin_amdgcn_wave_barrier();
        if (split && i < N) { dL_dshs[3 * (size_t)i] = dsh[0]; dL_dshs[3 * (size_t)i + 1] = dsh[1]; dL_dshs[3 * (size_t)i + 2] = dsh[2]; }
#pragma unroll
        for (int hf = 0; hf < HALVES; hf++) {
            if ((int)(lane / HROWS) == hf) {
                float* row = sh_rows + (lane % HROWS) * rowf;
                if (split) {
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
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int r0 = hf * HROWS;                                       // first row of this half inside the workgroup
            float* dst = (split ? dL_dshs_rest : dL_dshs) + ((size_t)i0 + r0) * rowf;
            const int nfl = max(0, min(HROWS, rows - r0)) * rowf;
            if (nfl > 0) {
                if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
                    const int nvec = nfl >> 2;
                    // gradient rows are written once and read next by the optimizer, long after this view: non-temporal stores (round 4).
                    // This kernel itself does not get faster (0.386 vs 0.388 ms) -- the NEXT view's K1 does, 0.204 -> 0.195 ms: 576 MB of
                    // gradients no longer push its inputs out of the cache
                    typedef float lg_f4v __attribute__((ext_vector_type(4)));
                    for (int q = (int)lane; q < nvec; q += LG_PP)
                        __builtin_nontemporal_store(reinterpret_cast<const lg_f4v*>(sh_rows)[q], reinterpret_cast<lg_f4v*>(dst) + q);
                    for (int f = (nvec << 2) + (int)lane; f < nfl; f += LG_PP) dst[f] = sh_rows[f];
                } else {
                    for (int f = (int)lane; f < nfl; f += LG_PP) dst[f] = sh_rows[f];
                }
            }
            if (hf + 1 < HALVES) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();             // the copy has read the rows before the next half overwrites them
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    // (staging the [N,3] gradients through LDS for lane-contiguous stores, as K1 does for its records, was measured:
    // 0.366 -> 0.370 ms, no gain -- this kernel already runs at ~4.4 TB/s of algorithmic traffic)
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
// Data-parallel steps (round 5; r4 verdict item 4): the SH-coefficient gradient of ONE view is rank one -- dL/dsh[k][c] =
// basis_k(dir) * dRGB[c] with dir = normalize(xyz - camera centre) -- so a rank need not ship its 12 M bytes per Gaussian: it ships
// dRGB [N, 3] (K9's rgb_only output) and its camera centre, and every rank rebuilds
//     dL/dsh[i][k][c] = ( sum over views v, in view order, of  basis_k(dir_v(i)) * dRGB_v[i][c] ) / divisor
// here.  Each term is evaluated by lg_backward_sh_jac's own expressions (zero Jacobian: the direction path stays in dL/dmeans3D),
// i.e. bit for bit the value K9 would have written for that view; the terms are then added in view order, so every rank holds the
// SAME bits whatever the collective's reduction order would have been, and at two ranks the result equals the dense exchange
// (t0 + t1) / 2 exactly.  One wave per 64 Gaussians; rows leave through LDS as coalesced 16-byte stores like K9's.
__global__ void __launch_bounds__(LG_PP)
lg_sh_grad_from_rgb_kernel(int N, int M, int D, int V, const float* __restrict__ means3D, const float* __restrict__ campos,
                           const float* __restrict__ drgb, size_t view_stride, float divisor, int accumulate, float* __restrict__ dL_dshs,
                           float* __restrict__ dL_dshs_rest)
{
    __shared__ __attribute__((aligned(16))) float sh_rows[LG_PP * LG_SH_MAXF];
    const uint32_t lane = threadIdx.x;
    const int i0 = (int)blockIdx.x * LG_PP;
    const int i = i0 + (int)lane;
    const bool split = dL_dshs_rest != nullptr;
    const int rowf = split ? 3 * (M - 1) : 3 * M;
    const int rows = min(LG_PP, N - i0);
    float dsh[LG_SH_MAXF];
#pragma unroll
    for (int k = 0; k < LG_SH_MAXF; k++) dsh[k] = 0.0f;
    if (accumulate) {
        // a camera batch per rank: the views of the batch arrive in several calls; this one continues the running sum the previous call
        // left in the output (rows in through LDS, coalesced, as they leave)
        if (rowf > 0) {
            const float* src = (split ? dL_dshs_rest : dL_dshs) + (size_t)i0 * rowf;
            const int nfl = rows * rowf;
            for (int f = (int)lane; f < nfl; f += LG_PP) sh_rows[f] = src[f];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float* row = sh_rows + lane * rowf;
            if (i < N) {
#pragma unroll
                for (int k = 0; k < LG_SH_MAXF; k++) {
                    if (split) { if (k >= 3 && k - 3 < rowf) dsh[k] = row[k - 3]; }
                    else if (k < rowf) dsh[k] = row[k];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (split && i < N) { dsh[0] = dL_dshs[3 * (size_t)i]; dsh[1] = dL_dshs[3 * (size_t)i + 1]; dsh[2] = dL_dshs[3 * (size_t)i + 2]; }
    }
    if (i < N) {
        const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
        const float J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int v = 0; v < V; v++) {
            const float* dr = drgb + (size_t)v * view_stride + 3 * (size_t)i;
            const float dRGB[3] = { dr[0], dr[1], dr[2] };
            const float cp[3] = { campos[3 * v], campos[3 * v + 1], campos[3 * v + 2] };
            float dm[3] = {0, 0, 0};
            if (v == 0 && !accumulate) lg_backward_sh_jac(D, J, px, py, pz, cp, dRGB, dm, [&](int k, int c, float val) { dsh[k * 3 + c] = val; });
            else lg_backward_sh_jac(D, J, px, py, pz, cp, dRGB, dm, [&](int k, int c, float val) { dsh[k * 3 + c] += val; });
        }
        if (divisor != 1.0f) {
            // a power of two (2, 4, 8 ranks): the reciprocal is exact and x * (1 / d) == x / d bit for bit -- one multiply instead of a
            // correctly rounded division per coefficient; any other rank count divides, as torch's div_ does in the dense exchange
            const uint32_t db = __float_as_uint(divisor);
            if ((db & 0x007FFFFFu) == 0u) {
                const float inv = 1.0f / divisor;
#pragma unroll
                for (int k = 0; k < LG_SH_MAXF; k++) dsh[k] = dsh[k] * inv;
            } else {
#pragma unroll
                for (int k = 0; k < LG_SH_MAXF; k++) dsh[k] = dsh[k] / divisor;
            }
        }
    }
    if (split && i < N) { dL_dshs[3 * (size_t)i] = dsh[0]; dL_dshs[3 * (size_t)i + 1] = dsh[1]; dL_dshs[3 * (size_t)i + 2] = dsh[2]; }
    if (rowf <= 0) return;
    float* row = sh_rows + lane * rowf;
    if (split) {
#pragma unroll
        for (int k = 3; k < LG_SH_MAXF; k++)
            if (k - 3 < rowf) row[k - 3] = dsh[k];
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
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int nvec = nfl >> 2;
        for (int q = (int)lane; q < nvec; q += LG_PP) reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(sh_rows)[q];
        for (int f = (nvec << 2) + (int)lane; f < nfl; f += LG_PP) dst[f] = sh_rows[f];
    } else {
        for (int f = (int)lane; f < nfl; f += LG_PP) dst[f] = sh_rows[f];
    }
}
