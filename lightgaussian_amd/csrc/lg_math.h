// lg_math.h -- per-Gaussian / per-pair arithmetic of the MI355X rasterizer.
//
// Pure scalar float code, no wave intrinsics: the same functions are called from the gfx950
// kernels (lg_preprocess.h / lg_blend.h, compiled through lg_api.hip) and, compiled with g++ by tests/cpu_harness, from the CPU-side unit
// tests that pin them bit-for-bit against the oracle.  Everything that feeds a threshold test
// (power > 0, alpha < 1/255, T < 1e-4, tile rectangles) follows the CANONICAL OPERATION ORDER
// documented in DESIGN.md section 4: plain IEEE mul/add/div/sqrt with contraction disabled
// (-ffp-contract=off) and explicit fmaf() only where written.
//
// Replaces (reference interface): the un-vendored CUDA submodule
// submodules/compress-diff-gaussian-rasterization (.gitmodules:6-8); conventions anchored on
// utils/sh_utils.py:26-103, utils/general_utils.py:68-119, scene/cameras.py:70-85.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LG_HD __host__ __device__ __forceinline__
#else
#define LG_HD static inline
#endif

#define LG_TILE 16
#define LG_ALPHA_MIN (1.0f / 255.0f)
#define LG_T_MIN 0.0001f
#define LG_ALPHA_MAX 0.99f

enum { LG_W_ONE = 0, LG_W_OPACITY = 1, LG_W_ALPHA = 2, LG_W_ALPHA_T = 3 };

LG_HD float lg_bits2f(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
LG_HD uint32_t lg_f2bits(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

// ---------------------------------------------------------------------------------------------
// Canonical exp for x <= 0: Cody-Waite reduction + Cephes degree-5 polynomial.  Only exactly
// rounded primitives, so CPU and GPU agree bitwise.  (The hardware v_exp_f32 is ~1 ulp but not
// reproducible on a CPU; it is used only by the non-exact "fast" blend variant.)
LG_HD float lg_exp(float x)
{
    x = fmaxf(x, -87.0f);
    float t = x * 1.44269504088896341f;
    float n = rintf(t);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float r2 = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    return p * lg_bits2f((uint32_t)((int)n + 127) << 23);
}

// ---------------------------------------------------------------------------------------------
// Q24.40 fixed point of the per-hit significance weights (LG_W_ALPHA / LG_W_ALPHA_T).
// lg_fix40_bits(w): bit pattern of (double) w + 4096.  For 0 <= w < 4096 the sum lies in [2^12, 2^13), where a double's unit in the
// last place is 2^-40: the IEEE addition rounds w to the nearest multiple of 2^-40 (ties to even) and the low 40 bits of the
// mantissa field are that multiple.  LG_FIX_MAGIC = bits(4096.0); bits - LG_FIX_MAGIC = round(w 2^40).
// lg_fix40_score(q): the per-view sum as fp32 -- ONE rounding (nearest even) of the integer, then an exact scaling by 2^-40.
#define LG_FIX_FRAC_BITS 40
#define LG_FIX_MAGIC 0x40B0000000000000ull
LG_HD uint64_t lg_fix40_bits(float w) { union { double d; uint64_t u; } c; c.d = (double)w + 4096.0; return c.u; }
LG_HD uint64_t lg_fix40_quant(float w) { return lg_fix40_bits(w) - LG_FIX_MAGIC; }
LG_HD float lg_fix40_score(uint64_t q) { return (float)q * 0x1p-40f; }

// ---------------------------------------------------------------------------------------------
// seqsum32(w, c): the float obtained by c sequential additions of w starting from 0 -- i.e. what
// c atomicAdd(float*, w) calls produce (all addends equal, so order independent).  O(#binades)
// instead of O(c): inside one binade every step adds the same multiple of the ulp, so the run
// of identical steps is applied as one exact integer multiply.  SURVEY.md section 8a-note.
LG_HD float lg_seqsum32(float w, uint32_t c)
{
    float s = 0.0f;
    if (!(w > 0.0f)) { // w <= 0 or NaN: plain loop semantics are trivial for 0; keep exact loop otherwise
        for (uint32_t k = 0; k < c; k++) s = s + w;
        return s;
    }
    while (c > 0) {
        // three real steps a -> b -> d; if all in one binade the increment (d - b) is the steady one
        float a = s + w; c--;
        if (a == s) return s; // stalled: every further add is a no-op
        s = a;
        if (c == 0) break;
        float b = a + w; c--;
        s = b;
        if (c == 0) break;
        float d = b + w; c--;
        s = d;
        if (c == 0) break;
        uint32_t ua = lg_f2bits(a), ub = lg_f2bits(b), ud = lg_f2bits(d);
        uint32_t e = ud >> 23;
        if ((ua >> 23) != e || (ub >> 23) != e) continue;       // crossed a binade: keep stepping
        uint32_t inc = ud - ub;                                    // steady increment in ulps
        if (inc == 0) return s;
        uint32_t top = (e << 23) | 0x7FFFFFu;                      // largest value of this binade
        uint32_t room = (top - ud) / inc;                          // steps that stay inside it
        uint32_t k = room < c ? room : c;
        s = lg_bits2f(ud + k * inc);
        c -= k;
    }
    return s;
}

// ---------------------------------------------------------------------------------------------
// 3D covariance from (scale, quaternion): Sigma = (R S)(R S)^T packed (xx,xy,xz,yy,yz,zz).
LG_HD void lg_cov3d(const float s_in[3], float mod, const float q[4], float cov[6])
{
    float s0 = mod * s_in[0], s1 = mod * s_in[1], s2 = mod * s_in[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R00 = 1.0f - 2.0f * (y * y + z * z), R01 = 2.0f * (x * y - r * z), R02 = 2.0f * (x * z + r * y);
    float R10 = 2.0f * (x * y + r * z), R11 = 1.0f - 2.0f * (x * x + z * z), R12 = 2.0f * (y * z - r * x);
    float R20 = 2.0f * (x * z - r * y), R21 = 2.0f * (y * z + r * x), R22 = 1.0f - 2.0f * (x * x + y * y);
    float L00 = R00 * s0, L01 = R01 * s1, L02 = R02 * s2;
    float L10 = R10 * s0, L11 = R11 * s1, L12 = R12 * s2;
    float L20 = R20 * s0, L21 = R21 * s1, L22 = R22 * s2;
    cov[0] = L00 * L00 + L01 * L01 + L02 * L02;
    cov[1] = L00 * L10 + L01 * L11 + L02 * L12;
    cov[2] = L00 * L20 + L01 * L21 + L02 * L22;
    cov[3] = L10 * L10 + L11 * L11 + L12 * L12;
    cov[4] = L10 * L20 + L11 * L21 + L12 * L22;
    cov[5] = L20 * L20 + L21 * L21 + L22 * L22;
}

struct LgEwa {
    float T2[6];     // J * Wrot, 2x3 row-major
    float txc, tyc;  // clamped view-space x, y
    int xclamp, yclamp;
};

LG_HD void lg_ewa(const float* vm, float tx, float ty, float tz, float fx, float fy, float limx, float limy, LgEwa& e)
{
    float txtz = tx / tz, tytz = ty / tz;
    e.xclamp = (txtz < -limx || txtz > limx);
    e.yclamp = (tytz < -limy || tytz > limy);
    float cx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    float cy = fminf(limy, fmaxf(-limy, tytz)) * tz;
    e.txc = cx; e.tyc = cy;
    float J00 = fx / tz, J02 = -(fx * cx) / (tz * tz);
    float J11 = fy / tz, J12 = -(fy * cy) / (tz * tz);
    for (int k = 0; k < 3; k++) {
        e.T2[k] = J00 * vm[4 * k + 0] + J02 * vm[4 * k + 2];
        e.T2[3 + k] = J11 * vm[4 * k + 1] + J12 * vm[4 * k + 2];
    }
}

struct LgCov2D { float a, b, c, U[3], V[3]; };

LG_HD void lg_cov2d(const float* T2, const float* S, LgCov2D& o)
{
    o.U[0] = T2[0] * S[0] + T2[1] * S[1] + T2[2] * S[2];
    o.U[1] = T2[0] * S[1] + T2[1] * S[3] + T2[2] * S[4];
    o.U[2] = T2[0] * S[2] + T2[1] * S[4] + T2[2] * S[5];
    o.V[0] = T2[3] * S[0] + T2[4] * S[1] + T2[5] * S[2];
    o.V[1] = T2[3] * S[1] + T2[4] * S[3] + T2[5] * S[4];
    o.V[2] = T2[3] * S[2] + T2[4] * S[4] + T2[5] * S[5];
    float a = o.U[0] * T2[0] + o.U[1] * T2[1] + o.U[2] * T2[2];
    float b = o.U[0] * T2[3] + o.U[1] * T2[4] + o.U[2] * T2[5];
    float c = o.V[0] * T2[3] + o.V[1] * T2[4] + o.V[2] * T2[5];
    o.a = a + 0.3f; o.b = b; o.c = c + 0.3f;
}

// Result of projecting one Gaussian (K1).
struct LgSplat {
    float x, y, depth;       // pixel-space mean, view-space z
    float ha, nb, hc;        // -0.5*A, -B, -0.5*C of the conic (exact rescalings)
    float hx, hy;            // conservative half-extent of the alpha >= 1/255 footprint (inf = no cull)
    int radius;
    int rx0, ry0, rx1, ry1;  // reference tile rectangle (max exclusive)
    int tx0, ty0, tx1, ty1;  // tight tile rectangle actually emitted (subset of the reference one)
};

// Projection + EWA splat + radius + tile rectangles.  Returns false when the Gaussian is not
// rasterised (near-culled, singular covariance, empty rectangle) -- radii must then be 0.
LG_HD bool lg_project(const float* vm, const float* pm, float px, float py, float pz, const float* cov3D, float opacity,
                      int W, int H, float tanfovx, float tanfovy, LgSplat& o)
{
    float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    if (vz <= 0.2f) return false;
    float hxm = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
    float hym = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
    float hwm = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
    float p_w = 1.0f / (hwm + 0.0000001f);
    float ndcx = hxm * p_w, ndcy = hym * p_w;

    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    LgEwa e;
    lg_ewa(vm, vx, vy, vz, fx, fy, 1.3f * tanfovx, 1.3f * tanfovy, e);
    LgCov2D c2;
    lg_cov2d(e.T2, cov3D, c2);
    const float a = c2.a, b = c2.b, c = c2.c;
    float det = a * c - b * b;
    if (det == 0.0f) return false;
    float det_inv = 1.0f / det;
    float A = c * det_inv, B = -b * det_inv, C = a * det_inv;
    float mid = 0.5f * (a + c);
    float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    float l1 = mid + sq, l2 = mid - sq;
    float rad = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
    float ix = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
    float iy = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const int gx = (W + LG_TILE - 1) / LG_TILE, gy = (H + LG_TILE - 1) / LG_TILE;
    // the casts must not overflow for absurd coordinates: clamp the float first (no effect in range)
    const float big = 1.0e7f;
    int r0 = (int)fminf(big, fmaxf(-big, (ix - rad) / (float)LG_TILE));
    int r1 = (int)fminf(big, fmaxf(-big, (iy - rad) / (float)LG_TILE));
    int r2 = (int)fminf(big, fmaxf(-big, (ix + rad + (float)(LG_TILE - 1)) / (float)LG_TILE));
    int r3 = (int)fminf(big, fmaxf(-big, (iy + rad + (float)(LG_TILE - 1)) / (float)LG_TILE));
    o.rx0 = r0 < 0 ? 0 : (r0 > gx ? gx : r0);
    o.ry0 = r1 < 0 ? 0 : (r1 > gy ? gy : r1);
    o.rx1 = r2 < 0 ? 0 : (r2 > gx ? gx : r2);
    o.ry1 = r3 < 0 ? 0 : (r3 > gy ? gy : r3);
    if ((o.rx1 - o.rx0) * (o.ry1 - o.ry0) == 0) return false;
    o.x = ix; o.y = iy; o.depth = vz; o.radius = (int)rad;
    o.ha = -0.5f * A; o.nb = -B; o.hc = -0.5f * C;

    // ---- exact-conservative footprint culling (not part of the reference semantics: it only
    // removes (tile, Gaussian) instances every pixel of which would fail alpha >= 1/255) ----
    o.tx0 = o.rx0; o.ty0 = o.ry0; o.tx1 = o.rx1; o.ty1 = o.ry1;
    o.hx = INFINITY; o.hy = INFINITY;
    if (!(opacity >= LG_ALPHA_MIN)) {
        // alpha = min(0.99, opacity * exp(power<=0)) <= opacity < 1/255 everywhere: no instance needed
        if (opacity < LG_ALPHA_MIN) { o.tx1 = o.tx0; o.ty1 = o.ty0; o.hx = -1.0f; o.hy = -1.0f; }
        return true; // (NaN opacity: keep the full rectangle, behave like the reference)
    }
    float amp = (a * c) / det; // cancellation amplification of det, conic
    if (amp > 0.0f && amp < 100.0f && rad < 4096.0f) {
        float dmax = rad + (float)LG_TILE;
        float smax = (fabsf(o.ha) + fabsf(o.nb) + fabsf(o.hc)) * dmax * dmax;
        float tau = logf(opacity * 255.0f);
        float taup = (tau + 1.0e-6f * smax) * 1.00001f + 1.0e-4f;
        if (taup < 1.0e30f) {
            float hx = sqrtf(2.0f * taup * a) * 1.001f + 0.01f;
            float hy = sqrtf(2.0f * taup * c) * 1.001f + 0.01f;
            int t0 = (int)fminf(big, fmaxf(-big, ceilf((ix - hx - (float)(LG_TILE - 1)) / (float)LG_TILE)));
            int t1 = (int)fminf(big, fmaxf(-big, floorf((ix + hx) / (float)LG_TILE))) + 1;
            int u0 = (int)fminf(big, fmaxf(-big, ceilf((iy - hy - (float)(LG_TILE - 1)) / (float)LG_TILE)));
            int u1 = (int)fminf(big, fmaxf(-big, floorf((iy + hy) / (float)LG_TILE))) + 1;
            o.tx0 = t0 > o.rx0 ? t0 : o.rx0; o.tx1 = t1 < o.rx1 ? t1 : o.rx1;
            o.ty0 = u0 > o.ry0 ? u0 : o.ry0; o.ty1 = u1 < o.ry1 ? u1 : o.ry1;
            if (o.tx0 > o.rx1) o.tx0 = o.rx1; // footprint entirely beside the rectangle: empty, but keep it a sub-rectangle
            if (o.ty0 > o.ry1) o.ty0 = o.ry1;
            if (o.tx1 < o.tx0) o.tx1 = o.tx0;
            if (o.ty1 < o.ty0) o.ty1 = o.ty0;
            o.hx = hx; o.hy = hy;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// SH -> RGB (utils/sh_utils.py:57-103 polynomial; +0.5 and clamp as gaussian_renderer/__init__.py:99)
#define LG_SH_C0 0.28209479177387814f
#define LG_SH_C1 0.4886025119029199f
#define LG_SH_C2_0 1.0925484305920792f
#define LG_SH_C2_1 -1.0925484305920792f
#define LG_SH_C2_2 0.31539156525252005f
#define LG_SH_C2_3 -1.0925484305920792f
#define LG_SH_C2_4 0.5462742152960396f
#define LG_SH_C3_0 -0.5900435899266435f
#define LG_SH_C3_1 2.890611442640554f
#define LG_SH_C3_2 -0.4570457994644658f
#define LG_SH_C3_3 0.3731763325901154f
#define LG_SH_C3_4 -0.4570457994644658f
#define LG_SH_C3_5 1.445305721320277f
#define LG_SH_C3_6 -0.5900435899266435f

// sh: [M][3] for this Gaussian.  Writes rgb (clamped at 0) and the clamp bits (bit c = channel c clamped).
LG_HD void lg_sh_to_rgb(int deg, const float* sh, float px, float py, float pz, const float* campos, float rgb[3],
                        uint32_t& clamp_bits)
{
    float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx = dx / len; dy = dy / len; dz = dz / len;
    clamp_bits = 0;
    for (int c = 0; c < 3; c++) {
        float res = LG_SH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            res = res - LG_SH_C1 * dy * sh[1 * 3 + c] + LG_SH_C1 * dz * sh[2 * 3 + c] - LG_SH_C1 * dx * sh[3 * 3 + c];
            if (deg > 1) {
                float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                res = res + LG_SH_C2_0 * xy * sh[4 * 3 + c] + LG_SH_C2_1 * yz * sh[5 * 3 + c] +
                      LG_SH_C2_2 * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + LG_SH_C2_3 * xz * sh[7 * 3 + c] +
                      LG_SH_C2_4 * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    res = res + LG_SH_C3_0 * dy * (3.0f * xx - yy) * sh[9 * 3 + c] + LG_SH_C3_1 * xy * dz * sh[10 * 3 + c] +
                          LG_SH_C3_2 * dy * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
                          LG_SH_C3_3 * dz * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
                          LG_SH_C3_4 * dx * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
                          LG_SH_C3_5 * dz * (xx - yy) * sh[14 * 3 + c] + LG_SH_C3_6 * dx * (xx - 3.0f * yy) * sh[15 * 3 + c];
                }
            }
        }
        float v = res + 0.5f;
        if (v < 0.0f) clamp_bits |= (1u << c);
        rgb[c] = fmaxf(v, 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// One (pixel, Gaussian) evaluation of the forward blend, canonical order.
// Returns 0 = rejected, 1 = contributes (T, C updated), 2 = pixel saturated (done).
template <bool EXACT>
LG_HD int lg_blend_pair(float gx, float gy, float ha, float nb, float hc, float op, float r, float g, float b, float pxf,
                        float pyf, float& T, float& C0, float& C1, float& C2, float& alpha_out)
{
    const float dx = gx - pxf, dy = gy - pyf;
    const float power = fmaf(fmaf(ha, dx, nb * dy), dx, (hc * dy) * dy);
    if (power > 0.0f) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    const float ex = EXACT ? lg_exp(power) : __expf(power);
#else
    const float ex = lg_exp(power);
#endif
    const float alpha = fminf(LG_ALPHA_MAX, op * ex);
    if (alpha < LG_ALPHA_MIN) return 0;
    const float test_T = T * (1.0f - alpha);
    if (test_T < LG_T_MIN) return 2;
    const float w = alpha * T;
    C0 = fmaf(r, w, C0);
    C1 = fmaf(g, w, C1);
    C2 = fmaf(b, w, C2);
    T = test_T;
    alpha_out = alpha;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Backward of the per-Gaussian stage (K8 + K9 fused).  acc = the 9 blend-stage sums:
//   [0,1] d/d(mean2D pixel x,y)  [2,3,4] d/d(conic A,B,C) (B: full derivative)  [5] d/d(opacity)  [6..8] d/d(rgb)
struct LgGradOut {
    float mean2D[2];   // NDC units
    float mean3D[3];
    float cov3D[6];
    float scale[3];
    float rot[4];
};

// Gradient rows of the backward blend hold pixel-offset MOMENTS (lg_blend.h): m[0..4] = sum t {dx, dy, dx^2, dx dy, dy^2},
// m[5] = sum t, with t = G dL/dalpha and dx = x_g - px; m[6..8] = dL/drgb.  The published per-pair expressions are linear in
// them with per-Gaussian coefficients (ha = -A/2, nb = -B, hc = -C/2 of the conic, opacity op), applied here once per Gaussian:
// acc = {dL/dmean2D.x, .y (pixel units), dL/dA, dL/dB, dL/dC, dL/dopacity, dL/drgb}, the input of lg_backward_geom.
LG_HD void lg_rows_to_grads(const float* m, float ha, float nb, float hc, float op, float* acc)
{
    acc[0] = op * (2.0f * ha * m[0] + nb * m[1]);
    acc[1] = op * (2.0f * hc * m[1] + nb * m[0]);
    acc[2] = -0.5f * op * m[2];
    acc[3] = -op * m[3];
    acc[4] = -0.5f * op * m[4];
    acc[5] = m[5];
    acc[6] = m[6]; acc[7] = m[7]; acc[8] = m[8];
}

LG_HD void lg_backward_geom(const float* vm, const float* pm, float px, float py, float pz, const float* S /*cov3D*/,
                            const float* acc, int W, int H, float tanfovx, float tanfovy, LgGradOut& g)
{
    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    const float gndx = acc[0] * (0.5f * (float)W), gndy = acc[1] * (0.5f * (float)H);
    g.mean2D[0] = gndx; g.mean2D[1] = gndy;
    float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    LgEwa e;
    lg_ewa(vm, vx, vy, vz, fx, fy, 1.3f * tanfovx, 1.3f * tanfovy, e);
    LgCov2D c2;
    lg_cov2d(e.T2, S, c2);
    const float a = c2.a, b = c2.b, c = c2.c;
    const float* T2 = e.T2;
    const float gA = acc[2], gB = acc[3], gC = acc[4];
    float denom = a * c - b * b;
    float d2inv = 1.0f / (denom * denom + 0.0000001f);
    float dL_da = 0.0f, dL_db = 0.0f, dL_dc = 0.0f;
    if (d2inv != 0.0f) {
        dL_da = d2inv * (-c * c * gA + b * c * gB + (denom - a * c) * gC);
        dL_dc = d2inv * (-a * a * gC + a * b * gB + (denom - a * c) * gA);
        dL_db = d2inv * (2.0f * b * c * gA - (denom + 2.0f * b * b) * gB + 2.0f * a * b * gC);
    }
    g.cov3D[0] = T2[0] * T2[0] * dL_da + T2[0] * T2[3] * dL_db + T2[3] * T2[3] * dL_dc;
    g.cov3D[3] = T2[1] * T2[1] * dL_da + T2[1] * T2[4] * dL_db + T2[4] * T2[4] * dL_dc;
    g.cov3D[5] = T2[2] * T2[2] * dL_da + T2[2] * T2[5] * dL_db + T2[5] * T2[5] * dL_dc;
    g.cov3D[1] = 2.0f * T2[0] * T2[1] * dL_da + (T2[0] * T2[4] + T2[1] * T2[3]) * dL_db + 2.0f * T2[3] * T2[4] * dL_dc;
    g.cov3D[2] = 2.0f * T2[0] * T2[2] * dL_da + (T2[0] * T2[5] + T2[2] * T2[3]) * dL_db + 2.0f * T2[3] * T2[5] * dL_dc;
    g.cov3D[4] = 2.0f * T2[1] * T2[2] * dL_da + (T2[1] * T2[5] + T2[2] * T2[4]) * dL_db + 2.0f * T2[4] * T2[5] * dL_dc;
    const float* U = c2.U; const float* V = c2.V;
    float dT00 = 2.0f * dL_da * U[0] + dL_db * V[0], dT01 = 2.0f * dL_da * U[1] + dL_db * V[1], dT02 = 2.0f * dL_da * U[2] + dL_db * V[2];
    float dT10 = dL_db * U[0] + 2.0f * dL_dc * V[0], dT11 = dL_db * U[1] + 2.0f * dL_dc * V[1], dT12 = dL_db * U[2] + 2.0f * dL_dc * V[2];
    float dJ00 = dT00 * vm[0] + dT01 * vm[4] + dT02 * vm[8];
    float dJ02 = dT00 * vm[2] + dT01 * vm[6] + dT02 * vm[10];
    float dJ11 = dT10 * vm[1] + dT11 * vm[5] + dT12 * vm[9];
    float dJ12 = dT10 * vm[2] + dT11 * vm[6] + dT12 * vm[10];
    float tz = 1.0f / vz, tz2 = tz * tz, tz3 = tz2 * tz;
    float dtx = (e.xclamp ? 0.0f : 1.0f) * (-fx * tz2 * dJ02);
    float dty = (e.yclamp ? 0.0f : 1.0f) * (-fy * tz2 * dJ12);
    float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0f * fx * e.txc) * tz3 * dJ02 + (2.0f * fy * e.tyc) * tz3 * dJ12;
    float dmx = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
    float dmy = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
    float dmz = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
    {
        float hxm = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
        float hym = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
        float hwm = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
        float m_w = 1.0f / (hwm + 0.0000001f);
        float mul1 = hxm * m_w * m_w, mul2 = hym * m_w * m_w;
        dmx += (pm[0] * m_w - pm[3] * mul1) * gndx + (pm[1] * m_w - pm[3] * mul2) * gndy;
        dmy += (pm[4] * m_w - pm[7] * mul1) * gndx + (pm[5] * m_w - pm[7] * mul2) * gndy;
        dmz += (pm[8] * m_w - pm[11] * mul1) * gndx + (pm[9] * m_w - pm[11] * mul2) * gndy;
    }
    g.mean3D[0] = dmx; g.mean3D[1] = dmy; g.mean3D[2] = dmz;
}

// dL/dSigma(packed) -> dL/dscale, dL/dquaternion.  Like the published implementation the
// scale gradient carries no scale_modifier factor.
LG_HD void lg_backward_cov3d(const float sc[3], float mod, const float q[4], const float dS[6], float dscale[3], float drot[4])
{
    float sv[3] = { mod * sc[0], mod * sc[1], mod * sc[2] };
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float Rm[9] = { 1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y),
                    2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x),
                    2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y) };
    float L[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[3 * i + j] = Rm[3 * i + j] * sv[j];
    float Gs[9] = { dS[0], 0.5f * dS[1], 0.5f * dS[2], 0.5f * dS[1], dS[3], 0.5f * dS[4], 0.5f * dS[2], 0.5f * dS[4], dS[5] };
    float dLm[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        dLm[3 * i + j] = 2.0f * (Gs[3 * i] * L[j] + Gs[3 * i + 1] * L[3 + j] + Gs[3 * i + 2] * L[6 + j]);
    for (int j = 0; j < 3; j++) dscale[j] = dLm[j] * Rm[j] + dLm[3 + j] * Rm[3 + j] + dLm[6 + j] * Rm[6 + j];
    float g[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) g[3 * i + j] = dLm[3 * i + j] * sv[j];
    drot[0] = 2.0f * (-z * g[1] + y * g[2] + z * g[3] - x * g[5] - y * g[6] + x * g[7]);
    drot[1] = 2.0f * (y * g[1] + z * g[2] + y * g[3] - 2.0f * x * g[4] - r * g[5] + z * g[6] + r * g[7] - 2.0f * x * g[8]);
    drot[2] = 2.0f * (-2.0f * y * g[0] + x * g[1] + r * g[2] + x * g[3] + z * g[5] - r * g[6] + z * g[7] - 2.0f * y * g[8]);
    drot[3] = 2.0f * (-2.0f * z * g[0] - r * g[1] + x * g[2] + r * g[3] - 2.0f * z * g[4] + y * g[5] + x * g[6] + y * g[7]);
}

// SH backward: writes dsh[M][3] through `store(k, c, value)` and accumulates the direction path
// into dmean[3].  dRGB already has clamped channels zeroed.
template <typename StoreFn>
LG_HD void lg_backward_sh(int deg, const float* sh, float px, float py, float pz, const float* campos, const float dRGB[3],
                          float dmean[3], StoreFn store)
{
    float ox = px - campos[0], oy = py - campos[1], oz = pz - campos[2];
    float len = sqrtf(ox * ox + oy * oy + oz * oz);
    float x = ox / len, y = oy / len, z = oz / len;
    float ddx = 0.0f, ddy = 0.0f, ddz = 0.0f;
    for (int c = 0; c < 3; c++) {
        const float d = dRGB[c];
        float dRdx = 0.0f, dRdy = 0.0f, dRdz = 0.0f;
        store(0, c, LG_SH_C0 * d);
        if (deg > 0) {
            store(1, c, -LG_SH_C1 * y * d);
            store(2, c, LG_SH_C1 * z * d);
            store(3, c, -LG_SH_C1 * x * d);
            dRdx = -LG_SH_C1 * sh[3 * 3 + c];
            dRdy = -LG_SH_C1 * sh[1 * 3 + c];
            dRdz = LG_SH_C1 * sh[2 * 3 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                store(4, c, LG_SH_C2_0 * xy * d);
                store(5, c, LG_SH_C2_1 * yz * d);
                store(6, c, LG_SH_C2_2 * (2.0f * zz - xx - yy) * d);
                store(7, c, LG_SH_C2_3 * xz * d);
                store(8, c, LG_SH_C2_4 * (xx - yy) * d);
                dRdx += LG_SH_C2_0 * y * sh[4 * 3 + c] + LG_SH_C2_2 * 2.0f * -x * sh[6 * 3 + c] + LG_SH_C2_3 * z * sh[7 * 3 + c] +
                        LG_SH_C2_4 * 2.0f * x * sh[8 * 3 + c];
                dRdy += LG_SH_C2_0 * x * sh[4 * 3 + c] + LG_SH_C2_1 * z * sh[5 * 3 + c] + LG_SH_C2_2 * 2.0f * -y * sh[6 * 3 + c] +
                        LG_SH_C2_4 * 2.0f * -y * sh[8 * 3 + c];
                dRdz += LG_SH_C2_1 * y * sh[5 * 3 + c] + LG_SH_C2_2 * 2.0f * 2.0f * z * sh[6 * 3 + c] + LG_SH_C2_3 * x * sh[7 * 3 + c];
                if (deg > 2) {
                    store(9, c, LG_SH_C3_0 * y * (3.0f * xx - yy) * d);
                    store(10, c, LG_SH_C3_1 * xy * z * d);
                    store(11, c, LG_SH_C3_2 * y * (4.0f * zz - xx - yy) * d);
                    store(12, c, LG_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * d);
                    store(13, c, LG_SH_C3_4 * x * (4.0f * zz - xx - yy) * d);
                    store(14, c, LG_SH_C3_5 * z * (xx - yy) * d);
                    store(15, c, LG_SH_C3_6 * x * (xx - 3.0f * yy) * d);
                    dRdx += LG_SH_C3_0 * sh[9 * 3 + c] * 3.0f * 2.0f * xy + LG_SH_C3_1 * sh[10 * 3 + c] * yz +
                            LG_SH_C3_2 * sh[11 * 3 + c] * -2.0f * xy + LG_SH_C3_3 * sh[12 * 3 + c] * -3.0f * 2.0f * xz +
                            LG_SH_C3_4 * sh[13 * 3 + c] * (-3.0f * xx + 4.0f * zz - yy) + LG_SH_C3_5 * sh[14 * 3 + c] * 2.0f * xz +
                            LG_SH_C3_6 * sh[15 * 3 + c] * 3.0f * (xx - yy);
                    dRdy += LG_SH_C3_0 * sh[9 * 3 + c] * 3.0f * (xx - yy) + LG_SH_C3_1 * sh[10 * 3 + c] * xz +
                            LG_SH_C3_2 * sh[11 * 3 + c] * (-3.0f * yy + 4.0f * zz - xx) +
                            LG_SH_C3_3 * sh[12 * 3 + c] * -3.0f * 2.0f * yz + LG_SH_C3_4 * sh[13 * 3 + c] * -2.0f * xy +
                            LG_SH_C3_5 * sh[14 * 3 + c] * -2.0f * yz + LG_SH_C3_6 * sh[15 * 3 + c] * -3.0f * 2.0f * xy;
                    dRdz += LG_SH_C3_1 * sh[10 * 3 + c] * xy + LG_SH_C3_2 * sh[11 * 3 + c] * 4.0f * 2.0f * yz +
                            LG_SH_C3_3 * sh[12 * 3 + c] * 3.0f * (2.0f * zz - xx - yy) +
                            LG_SH_C3_4 * sh[13 * 3 + c] * 4.0f * 2.0f * xz + LG_SH_C3_5 * sh[14 * 3 + c] * (xx - yy);
                }
            }
        }
        ddx += dRdx * d; ddy += dRdy * d; ddz += dRdz * d;
    }
    float sum2 = ox * ox + oy * oy + oz * oz;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
    dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
    dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
}


// The part of lg_backward_sh that needs the SH coefficients -- d rgb_c / d (unit view direction), J[3 c + {0,1,2}] = {dRdx, dRdy, dRdz}
// of channel c -- depends on forward-time quantities only.  K1 evaluates it next to the colours (the coefficients are in registers
// there) and leaves 36 bytes per visible Gaussian; K9 then never reads the 12 M bytes of SH coefficients again (round 4: 388 MB per view at
// C3).  The expressions are those of lg_backward_sh, term for term, so the values -- and the gradients -- are bit-identical.
LG_HD void lg_sh_dir_jacobian(int deg, const float* sh, float px, float py, float pz, const float* campos, float J[9])
{
    float ox = px - campos[0], oy = py - campos[1], oz = pz - campos[2];
    float len = sqrtf(ox * ox + oy * oy + oz * oz);
    float x = ox / len, y = oy / len, z = oz / len;
    for (int c = 0; c < 3; c++) {
        float dRdx = 0.0f, dRdy = 0.0f, dRdz = 0.0f;
        if (deg > 0) {
            dRdx = -LG_SH_C1 * sh[3 * 3 + c];
            dRdy = -LG_SH_C1 * sh[1 * 3 + c];
            dRdz = LG_SH_C1 * sh[2 * 3 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dRdx += LG_SH_C2_0 * y * sh[4 * 3 + c] + LG_SH_C2_2 * 2.0f * -x * sh[6 * 3 + c] + LG_SH_C2_3 * z * sh[7 * 3 + c] +
                        LG_SH_C2_4 * 2.0f * x * sh[8 * 3 + c];
                dRdy += LG_SH_C2_0 * x * sh[4 * 3 + c] + LG_SH_C2_1 * z * sh[5 * 3 + c] + LG_SH_C2_2 * 2.0f * -y * sh[6 * 3 + c] +
                        LG_SH_C2_4 * 2.0f * -y * sh[8 * 3 + c];
                dRdz += LG_SH_C2_1 * y * sh[5 * 3 + c] + LG_SH_C2_2 * 2.0f * 2.0f * z * sh[6 * 3 + c] + LG_SH_C2_3 * x * sh[7 * 3 + c];
                if (deg > 2) {
                    dRdx += LG_SH_C3_0 * sh[9 * 3 + c] * 3.0f * 2.0f * xy + LG_SH_C3_1 * sh[10 * 3 + c] * yz +
                            LG_SH_C3_2 * sh[11 * 3 + c] * -2.0f * xy + LG_SH_C3_3 * sh[12 * 3 + c] * -3.0f * 2.0f * xz +
                            LG_SH_C3_4 * sh[13 * 3 + c] * (-3.0f * xx + 4.0f * zz - yy) + LG_SH_C3_5 * sh[14 * 3 + c] * 2.0f * xz +
                            LG_SH_C3_6 * sh[15 * 3 + c] * 3.0f * (xx - yy);
                    dRdy += LG_SH_C3_0 * sh[9 * 3 + c] * 3.0f * (xx - yy) + LG_SH_C3_1 * sh[10 * 3 + c] * xz +
                            LG_SH_C3_2 * sh[11 * 3 + c] * (-3.0f * yy + 4.0f * zz - xx) +
                            LG_SH_C3_3 * sh[12 * 3 + c] * -3.0f * 2.0f * yz + LG_SH_C3_4 * sh[13 * 3 + c] * -2.0f * xy +
                            LG_SH_C3_5 * sh[14 * 3 + c] * -2.0f * yz + LG_SH_C3_6 * sh[15 * 3 + c] * -3.0f * 2.0f * xy;
                    dRdz += LG_SH_C3_1 * sh[10 * 3 + c] * xy + LG_SH_C3_2 * sh[11 * 3 + c] * 4.0f * 2.0f * yz +
                            LG_SH_C3_3 * sh[12 * 3 + c] * 3.0f * (2.0f * zz - xx - yy) +
                            LG_SH_C3_4 * sh[13 * 3 + c] * 4.0f * 2.0f * xz + LG_SH_C3_5 * sh[14 * 3 + c] * (xx - yy);
                }
            }
        }
        J[3 * c] = dRdx; J[3 * c + 1] = dRdy; J[3 * c + 2] = dRdz;
    }
}

// lg_backward_sh with the direction Jacobian handed in (lg_sh_dir_jacobian) instead of the coefficients: dL/dSH through store(k, c, value),
// the view-direction term into dmean[3].  Same operations in the same order as lg_backward_sh.
template <typename StoreFn>
LG_HD void lg_backward_sh_jac(int deg, const float J[9], float px, float py, float pz, const float* campos, const float dRGB[3],
                              float dmean[3], StoreFn store)
{
    float ox = px - campos[0], oy = py - campos[1], oz = pz - campos[2];
    float len = sqrtf(ox * ox + oy * oy + oz * oz);
    float x = ox / len, y = oy / len, z = oz / len;
    float ddx = 0.0f, ddy = 0.0f, ddz = 0.0f;
    for (int c = 0; c < 3; c++) {
        const float d = dRGB[c];
        store(0, c, LG_SH_C0 * d);
        if (deg > 0) {
            store(1, c, -LG_SH_C1 * y * d);
            store(2, c, LG_SH_C1 * z * d);
            store(3, c, -LG_SH_C1 * x * d);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                store(4, c, LG_SH_C2_0 * xy * d);
                store(5, c, LG_SH_C2_1 * yz * d);
                store(6, c, LG_SH_C2_2 * (2.0f * zz - xx - yy) * d);
                store(7, c, LG_SH_C2_3 * xz * d);
                store(8, c, LG_SH_C2_4 * (xx - yy) * d);
                if (deg > 2) {
                    store(9, c, LG_SH_C3_0 * y * (3.0f * xx - yy) * d);
                    store(10, c, LG_SH_C3_1 * xy * z * d);
                    store(11, c, LG_SH_C3_2 * y * (4.0f * zz - xx - yy) * d);
                    store(12, c, LG_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * d);
                    store(13, c, LG_SH_C3_4 * x * (4.0f * zz - xx - yy) * d);
                    store(14, c, LG_SH_C3_5 * z * (xx - yy) * d);
                    store(15, c, LG_SH_C3_6 * x * (xx - 3.0f * yy) * d);
                }
            }
        }
        ddx += J[3 * c] * d; ddy += J[3 * c + 1] * d; ddz += J[3 * c + 2] * d;
    }
    float sum2 = ox * ox + oy * oy + oz * oz;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
    dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
    dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
}
