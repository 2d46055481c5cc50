/*
 * lg_oracle.c -- CPU ORACLE for the LightGaussian differentiable-render hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (lightgaussian_amd/)
 * never links, imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the un-vendored git submodule
 * submodules/compress-diff-gaussian-rasterization (/root/reference/.gitmodules:6-8, empty
 * directory, no pinned SHA) and the reference ships no tests/golden vectors (SURVEY.md
 * section 0).  This file therefore restates the PUBLISHED algorithm of
 * graphdeco-inria/diff-gaussian-rasterization + the LightGaussian count/score addition
 * (paper Eq. GS_j = sum_rays 1(hit) * sigma_j * gamma(Sigma_j)), anchored on the reference's
 * own call sites:
 *   - call contract, tensor layouts, output order ........ gaussian_renderer/__init__.py:52-68,106-115,156-172,209-218
 *   - SH basis constants / polynomial order .............. utils/sh_utils.py:26-54,74-103   (pinned by tests/golden)
 *   - colour post-op  max(sh + 0.5, 0) .................... gaussian_renderer/__init__.py:99
 *   - quaternion (r,x,y,z) -> R, Sigma = (R S)(R S)^T,
 *     6-packing (xx,xy,xz,yy,yz,zz) ...................... utils/general_utils.py:68-119, scene/gaussian_model.py:29-33 (pinned by tests/golden)
 *   - row-vector matrix convention (W2C^T, full_proj) .... scene/cameras.py:70-85, utils/graphics_utils.py:42-76
 * Constants of the un-vendored rasterizer: 16x16 tiles, near cull
 * z_view <= 0.2, low-pass +0.3, alpha = min(0.99, sigma*exp(power)), skip alpha < 1/255,
 * stop when T*(1-alpha) < 1e-4, radius = ceil(3*sqrt(lambda_max)).
 *
 * Built twice by oracle/Makefile:  liblg_oracle_f32.so (float; canonical operation order --
 * the HIP kernels are written to reproduce these float results bit for bit in "exact" mode)
 * and liblg_oracle_f64.so (double twin, -DLG_F64, used for gradient checks).
 *
 * Canonical arithmetic rules (float build): compiled with -ffp-contract=off; every fused
 * multiply-add is an explicit fmaf(); exp() on the blend path is lg_exp() below (Cody-Waite
 * reduction + Cephes degree-5 polynomial, exactly-rounded primitive ops only) so that CPU
 * and GPU agree bitwise on every threshold test.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef LG_F64
typedef double real;
#define RC(x) x
#define R_SQRT sqrt
#define R_CEIL ceil
#define R_FMA fma
#define R_FMIN fmin
#define R_FMAX fmax
#else
typedef float real;
#define RC(x) x##f
#define R_SQRT sqrtf
#define R_CEIL ceilf
#define R_FMA fmaf
#define R_FMIN fminf
#define R_FMAX fmaxf
#endif

#define TILE 16

/* weight policies for the significance score */
enum { LG_W_ONE = 0, LG_W_OPACITY = 1, LG_W_ALPHA = 2, LG_W_ALPHA_T = 3 };

/* ------------------------------------------------------------------------------------------ */
/* deterministic exp for x <= 0 (blend path).  Spec shared with the HIP kernels.  */
static inline real lg_exp(real x)
{
#ifdef LG_F64
    return exp(x);
#else
    union { uint32_t u; float f; } sc;
    x = fmaxf(x, -87.0f);
    float t = x * 1.44269504088896341f;
    float n = rintf(t); /* round-to-nearest-even */
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
    sc.u = (uint32_t)((int)n + 127) << 23;
    return p * sc.f;
#endif
}

/* exported so tests can pin the GPU-side exp against it */
real lgo_exp(real x) { return lg_exp(x); }

/* SH constants: utils/sh_utils.py:26-54 */
static const real SH_C0 = RC(0.28209479177387814);
static const real SH_C1 = RC(0.4886025119029199);
static const real SH_C2[5] = { RC(1.0925484305920792), RC(-1.0925484305920792), RC(0.31539156525252005),
                               RC(-1.0925484305920792), RC(0.5462742152960396) };
static const real SH_C3[7] = { RC(-0.5900435899266435), RC(2.890611442640554), RC(-0.4570457994644658),
                               RC(0.3731763325901154), RC(-0.4570457994644658), RC(1.445305721320277),
                               RC(-0.5900435899266435) };

typedef struct {
    int N, W, H, gx, gy;
    uint64_t R;            /* number of (tile, gaussian) instances */
    uint32_t *point_list;  /* [R] gaussian ids sorted by (tile, depth, id) */
    uint32_t *range_lo;    /* [gx*gy] */
    uint32_t *range_hi;    /* [gx*gy] */
} lgo_ctx;

/* ------------------------------------------------------------------------------------------ */
/* K1: per-Gaussian projection.  Returns 1 if the Gaussian is rasterised.                      */
typedef struct {
    real xy[2], depth, conic[3], rgb[3], cov3D[6];
    int radius, rect[4]; /* xmin, ymin, xmax, ymax (tiles, max exclusive) */
    unsigned char clamped[3];
} splat_t;

static void cov3d_from_scale_rot(const real *scale, real mod, const real *q, real *cov)
{
    /* Sigma = (R S)(R S)^T, quaternion (r,x,y,z) NOT re-normalised (getter normalises) */
    real s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    real r = q[0], x = q[1], y = q[2], z = q[3];
    real R00 = RC(1.0) - RC(2.0) * (y * y + z * z), R01 = RC(2.0) * (x * y - r * z), R02 = RC(2.0) * (x * z + r * y);
    real R10 = RC(2.0) * (x * y + r * z), R11 = RC(1.0) - RC(2.0) * (x * x + z * z), R12 = RC(2.0) * (y * z - r * x);
    real R20 = RC(2.0) * (x * z - r * y), R21 = RC(2.0) * (y * z + r * x), R22 = RC(1.0) - RC(2.0) * (x * x + y * y);
    real L00 = R00 * s0, L01 = R01 * s1, L02 = R02 * s2;
    real L10 = R10 * s0, L11 = R11 * s1, L12 = R12 * s2;
    real L20 = R20 * s0, L21 = R21 * s1, L22 = R22 * s2;
    cov[0] = L00 * L00 + L01 * L01 + L02 * L02;
    cov[1] = L00 * L10 + L01 * L11 + L02 * L12;
    cov[2] = L00 * L20 + L01 * L21 + L02 * L22;
    cov[3] = L10 * L10 + L11 * L11 + L12 * L12;
    cov[4] = L10 * L20 + L11 * L21 + L12 * L22;
    cov[5] = L20 * L20 + L21 * L21 + L22 * L22;
}

/* T2 = J * Wrot (2x3), shared by forward and backward */
static void ewa_T(const real *vm, real tx, real ty, real tz, real fx, real fy, real limx, real limy,
                  real *T2 /*6*/, real *txc, real *tyc, int *xclamp, int *yclamp)
{
    real txtz = tx / tz, tytz = ty / tz;
    *xclamp = (txtz < -limx || txtz > limx);
    *yclamp = (tytz < -limy || tytz > limy);
    real cx = R_FMIN(limx, R_FMAX(-limx, txtz)) * tz;
    real cy = R_FMIN(limy, R_FMAX(-limy, tytz)) * tz;
    *txc = cx; *tyc = cy;
    real J00 = fx / tz, J02 = -(fx * cx) / (tz * tz);
    real J11 = fy / tz, J12 = -(fy * cy) / (tz * tz);
    /* Wm[c][k] = vm[4k + c] */
    for (int k = 0; k < 3; k++) {
        T2[k]     = J00 * vm[4 * k + 0] + J02 * vm[4 * k + 2];
        T2[3 + k] = J11 * vm[4 * k + 1] + J12 * vm[4 * k + 2];
    }
}

static void sh_basis_eval(int deg, int M, const real *sh /*[M][3]*/, real dx, real dy, real dz, real *out)
{
    (void)M;
    for (int c = 0; c < 3; c++) {
        real res = SH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            res = res - SH_C1 * dy * sh[1 * 3 + c] + SH_C1 * dz * sh[2 * 3 + c] - SH_C1 * dx * sh[3 * 3 + c];
            if (deg > 1) {
                real xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                res = res + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
                      SH_C2[2] * (RC(2.0) * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
                      SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    res = res + SH_C3[0] * dy * (RC(3.0) * xx - yy) * sh[9 * 3 + c] +
                          SH_C3[1] * xy * dz * sh[10 * 3 + c] +
                          SH_C3[2] * dy * (RC(4.0) * zz - xx - yy) * sh[11 * 3 + c] +
                          SH_C3[3] * dz * (RC(2.0) * zz - RC(3.0) * xx - RC(3.0) * yy) * sh[12 * 3 + c] +
                          SH_C3[4] * dx * (RC(4.0) * zz - xx - yy) * sh[13 * 3 + c] +
                          SH_C3[5] * dz * (xx - yy) * sh[14 * 3 + c] +
                          SH_C3[6] * dx * (xx - RC(3.0) * yy) * sh[15 * 3 + c];
                }
            }
        }
        out[c] = res;
    }
}

static int project_one(int i, int M, int D, int W, int H, const real *means3D, const real *shs,
                       const real *colors_precomp, const real *scales, real mod, const real *rotations,
                       const real *cov3D_precomp, const real *vm, const real *pm, const real *campos,
                       real tanfovx, real tanfovy, splat_t *o)
{
    const real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    real vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    real vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    real vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    if (vz <= RC(0.2)) return 0;
    real hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
    real hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
    real hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
    real p_w = RC(1.0) / (hw + RC(0.0000001));
    real ndcx = hx * p_w, ndcy = hy * p_w;

    if (cov3D_precomp) memcpy(o->cov3D, cov3D_precomp + 6 * i, 6 * sizeof(real));
    else cov3d_from_scale_rot(scales + 3 * i, mod, rotations + 4 * i, o->cov3D);
    const real *S = o->cov3D;

    const real fx = (real)W / (RC(2.0) * tanfovx), fy = (real)H / (RC(2.0) * tanfovy);
    real T2[6], txc, tyc; int xc, yc;
    ewa_T(vm, vx, vy, vz, fx, fy, RC(1.3) * tanfovx, RC(1.3) * tanfovy, T2, &txc, &tyc, &xc, &yc);
    /* U = T2 * Sigma (2x3) */
    real U0 = T2[0] * S[0] + T2[1] * S[1] + T2[2] * S[2];
    real U1 = T2[0] * S[1] + T2[1] * S[3] + T2[2] * S[4];
    real U2 = T2[0] * S[2] + T2[1] * S[4] + T2[2] * S[5];
    real V0 = T2[3] * S[0] + T2[4] * S[1] + T2[5] * S[2];
    real V1 = T2[3] * S[1] + T2[4] * S[3] + T2[5] * S[4];
    real V2 = T2[3] * S[2] + T2[4] * S[4] + T2[5] * S[5];
    real a = U0 * T2[0] + U1 * T2[1] + U2 * T2[2];
    real b = U0 * T2[3] + U1 * T2[4] + U2 * T2[5];
    real c = V0 * T2[3] + V1 * T2[4] + V2 * T2[5];
    a = a + RC(0.3);
    c = c + RC(0.3);
    real det = a * c - b * b;
    if (det == RC(0.0)) return 0;
    real det_inv = RC(1.0) / det;
    o->conic[0] = c * det_inv; o->conic[1] = -b * det_inv; o->conic[2] = a * det_inv;
    real mid = RC(0.5) * (a + c);
    real sq = R_SQRT(R_FMAX(RC(0.1), mid * mid - det));
    real l1 = mid + sq, l2 = mid - sq;
    real rad = R_CEIL(RC(3.0) * R_SQRT(R_FMAX(l1, l2)));
    real ix = ((ndcx + RC(1.0)) * (real)W - RC(1.0)) * RC(0.5);
    real iy = ((ndcy + RC(1.0)) * (real)H - RC(1.0)) * RC(0.5);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int r0 = (int)((ix - rad) / (real)TILE), r1 = (int)((iy - rad) / (real)TILE);
    int r2 = (int)((ix + rad + (real)(TILE - 1)) / (real)TILE), r3 = (int)((iy + rad + (real)(TILE - 1)) / (real)TILE);
    o->rect[0] = r0 < 0 ? 0 : (r0 > gx ? gx : r0);
    o->rect[1] = r1 < 0 ? 0 : (r1 > gy ? gy : r1);
    o->rect[2] = r2 < 0 ? 0 : (r2 > gx ? gx : r2);
    o->rect[3] = r3 < 0 ? 0 : (r3 > gy ? gy : r3);
    if ((o->rect[2] - o->rect[0]) * (o->rect[3] - o->rect[1]) == 0) return 0;

    o->clamped[0] = o->clamped[1] = o->clamped[2] = 0;
    if (colors_precomp) {
        for (int ch = 0; ch < 3; ch++) o->rgb[ch] = colors_precomp[3 * i + ch];
    } else {
        real dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
        real len = R_SQRT(dx * dx + dy * dy + dz * dz);
        dx = dx / len; dy = dy / len; dz = dz / len;
        real res[3];
        sh_basis_eval(D, M, shs + (size_t)i * M * 3, dx, dy, dz, res);
        for (int ch = 0; ch < 3; ch++) {
            real v = res[ch] + RC(0.5);
            o->clamped[ch] = (v < RC(0.0));
            o->rgb[ch] = R_FMAX(v, RC(0.0));
        }
    }
    o->xy[0] = ix; o->xy[1] = iy; o->depth = vz; o->radius = (int)rad;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* sorting of (tile, depth bits, id): stable LSD radix, 16-bit digits                          */
typedef struct { uint64_t key; uint64_t dkey; uint32_t id; } inst_t;

#ifdef LG_F64
static int cmp_inst(const void *a, const void *b)
{
    const inst_t *x = (const inst_t *)a, *y = (const inst_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->dkey != y->dkey) return x->dkey < y->dkey ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}
#endif

static void sort_instances(inst_t *a, size_t n, uint64_t ntiles)
{
#ifdef LG_F64
    (void)ntiles;
    qsort(a, n, sizeof(inst_t), cmp_inst);
#else
    /* key = tile<<32 | depth_bits.  3 passes over the low 48 bits while tile ids fit 16 bits, 4 beyond (round 3: a
     * 4112 x 4096 image has 65 792 tiles). */
    const int passes = ntiles > 65536u ? 4 : 3;
    inst_t *orig = a;
    inst_t *b = (inst_t *)malloc(n * sizeof(inst_t));
    inst_t *tmp = b;
    size_t *hist = (size_t *)malloc(65536 * sizeof(size_t));
    for (int pass = 0; pass < passes; pass++) {
        int sh = 16 * pass;
        memset(hist, 0, 65536 * sizeof(size_t));
        for (size_t i = 0; i < n; i++) hist[(a[i].key >> sh) & 0xFFFF]++;
        size_t acc = 0;
        for (int d = 0; d < 65536; d++) { size_t t = hist[d]; hist[d] = acc; acc += t; }
        for (size_t i = 0; i < n; i++) b[hist[(a[i].key >> sh) & 0xFFFF]++] = a[i];
        inst_t *t = a; a = b; b = t;
    }
    /* the result is in the buffer currently named 'a': the temporary after an odd number of passes */
    if (a != orig) memcpy(orig, a, n * sizeof(inst_t));
    free(tmp);
    free(hist);
#endif
}

/* ------------------------------------------------------------------------------------------ */
void lgo_free(lgo_ctx *ctx)
{
    if (!ctx) return;
    free(ctx->point_list); free(ctx->range_lo); free(ctx->range_hi); free(ctx);
}

uint64_t lgo_num_rendered(const lgo_ctx *ctx) { return ctx ? ctx->R : 0; }

/* Gaussian id of the last contributor of every pixel (0xFFFFFFFF: none), from the per-pixel contributor index the forward
 * left: n_contrib itself is a position in THIS implementation's tile list (the oracle enumerates the reference's full
 * rectangles, the HIP path culls instances that cannot contribute), the id it points at is what two implementations share. */
void lgo_last_contributor_ids(const lgo_ctx *ctx, const int *n_contrib, uint32_t *out_ids)
{
    const int W = ctx->W, H = ctx->H, gx = ctx->gx;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t pid = (size_t)y * W + x;
            const int t = (y / TILE) * gx + (x / TILE);
            const int n = n_contrib[pid];
            out_ids[pid] = n > 0 ? ctx->point_list[ctx->range_lo[t] + (uint32_t)n - 1u] : 0xFFFFFFFFu;
        }
}

/* per-view score from an integer hit count: c sequential additions of w starting from 0 in
 * the accumulator precision -- exactly what c atomicAdd(score, w) calls produce (all addends
 * equal => order independent).  SURVEY.md section 8a-note. */
/* Q24.40 restatement of the per-hit weight sums (ALPHA / ALPHA_T policies; see lgo_forward): quantisation of one weight and the
 * fp32 score of a sum. */
uint64_t lgo_fix40_quant(float w) { return (uint64_t)llrint(ldexp((double)w, 40)); }
float lgo_fix40_score(uint64_t q) { return (float)q * 0x1p-40f; }

real lgo_seqsum(real w, int c)
{
    real s = RC(0.0);
    for (int k = 0; k < c; k++) s = s + w;
    return s;
}

/*
 * Forward (+ optional count/score).  All tensors are caller-allocated, contiguous.
 * Saved state for backward: xy[N*2], depth[N], conic_opacity[N*4], rgb[N*3], cov3D[N*6],
 * clamped[N*3], final_T[P], n_contrib[P] + the returned ctx (sorted lists, tile ranges).
 */
lgo_ctx *lgo_forward(int N, int M, int D, int W, int H, const real *bg, const real *means3D,
                     const real *shs, const real *colors_precomp, const real *opacities,
                     const real *scales, real scale_modifier, const real *rotations,
                     const real *cov3D_precomp, const real *viewmatrix, const real *projmatrix,
                     const real *campos, real tanfovx, real tanfovy, int weight_policy,
                     real *out_color, int *radii, int *count, real *score, real *xy, real *depth,
                     real *conic_opacity, real *rgb, real *cov3D, unsigned char *clamped,
                     real *final_T, int *n_contrib)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    lgo_ctx *ctx = (lgo_ctx *)calloc(1, sizeof(lgo_ctx));
    ctx->N = N; ctx->W = W; ctx->H = H; ctx->gx = gx; ctx->gy = gy;
    int *rects = (int *)malloc((size_t)N * 4 * sizeof(int));
    uint32_t *touched = (uint32_t *)calloc((size_t)N + 1, sizeof(uint32_t));

#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        splat_t s;
        radii[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = RC(0.0); depth[i] = RC(0.0);
        for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = RC(0.0);
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = RC(0.0); clamped[3 * i + k] = 0; }
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = RC(0.0);
        for (int k = 0; k < 4; k++) rects[4 * i + k] = 0;
        if (!project_one(i, M, D, W, H, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, campos, tanfovx, tanfovy, &s)) {
            /* cov3D is still produced for culled-by-rect Gaussians upstream; irrelevant to outputs */
            continue;
        }
        radii[i] = s.radius;
        xy[2 * i] = s.xy[0]; xy[2 * i + 1] = s.xy[1]; depth[i] = s.depth;
        conic_opacity[4 * i] = s.conic[0]; conic_opacity[4 * i + 1] = s.conic[1];
        conic_opacity[4 * i + 2] = s.conic[2]; conic_opacity[4 * i + 3] = opacities[i];
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = s.rgb[k]; clamped[3 * i + k] = s.clamped[k]; }
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = s.cov3D[k];
        for (int k = 0; k < 4; k++) rects[4 * i + k] = s.rect[k];
        touched[i] = (uint32_t)((s.rect[2] - s.rect[0]) * (s.rect[3] - s.rect[1]));
    }

    /* exclusive scan -> offsets; duplicate with keys */
    uint64_t R = 0;
    uint64_t *offs = (uint64_t *)malloc(((size_t)N + 1) * sizeof(uint64_t));
    for (int i = 0; i < N; i++) { offs[i] = R; R += touched[i]; }
    offs[N] = R;
    ctx->R = R;
    inst_t *inst = (inst_t *)malloc((R ? R : 1) * sizeof(inst_t));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        if (!touched[i]) continue;
        uint64_t o = offs[i];
        uint64_t dk;
#ifdef LG_F64
        { union { double d; uint64_t u; } cv; cv.d = depth[i]; dk = cv.u; }
#else
        { union { float f; uint32_t u; } cv; cv.f = depth[i]; dk = cv.u; }
#endif
        for (int ty = rects[4 * i + 1]; ty < rects[4 * i + 3]; ty++)
            for (int tx = rects[4 * i]; tx < rects[4 * i + 2]; tx++) {
                uint64_t tile = (uint64_t)ty * gx + tx;
#ifdef LG_F64
                inst[o].key = tile; inst[o].dkey = dk;
#else
                inst[o].key = (tile << 32) | dk; inst[o].dkey = 0;
#endif
                inst[o].id = (uint32_t)i;
                o++;
            }
    }
    sort_instances(inst, R, (uint64_t)gx * (uint64_t)gy);
    ctx->point_list = (uint32_t *)malloc((R ? R : 1) * sizeof(uint32_t));
    ctx->range_lo = (uint32_t *)calloc((size_t)gx * gy, sizeof(uint32_t));
    ctx->range_hi = (uint32_t *)calloc((size_t)gx * gy, sizeof(uint32_t));
    for (uint64_t k = 0; k < R; k++) {
        ctx->point_list[k] = inst[k].id;
#ifdef LG_F64
        uint32_t t = (uint32_t)inst[k].key;
        uint32_t tp = k ? (uint32_t)inst[k - 1].key : 0xFFFFFFFFu;
#else
        uint32_t t = (uint32_t)(inst[k].key >> 32);
        uint32_t tp = k ? (uint32_t)(inst[k - 1].key >> 32) : 0xFFFFFFFFu;
#endif
        if (k == 0) ctx->range_lo[t] = 0;
        else if (t != tp) { ctx->range_hi[tp] = (uint32_t)k; ctx->range_lo[t] = (uint32_t)k; }
        if (k == R - 1) ctx->range_hi[t] = (uint32_t)R;
    }
    free(inst); free(offs); free(touched); free(rects);

    if (count) memset(count, 0, (size_t)N * sizeof(int));
    /* ALPHA / ALPHA_T policies: the weight differs from hit to hit, so a float sum would depend on the order of the hits.  The
     * path's definition: every hit's weight, AS THE fp32 VALUE THE BLEND COMPUTED, is rounded to the nearest
     * multiple of 2^-40 (ties to even) and the multiples are added as 64-bit integers (Q24.40) -- associative, order-free -- and the
     * per-view score is that integer rounded once to fp32 (nearest even) times 2^-40.  Independent restatement: llrint(ldexp(w, 40))
     * under the default rounding mode; the HIP kernel gets the same integer out of the mantissa of (double) w + 4096. */
    uint64_t *fix = NULL;
    if (score) {
        for (int i = 0; i < N; i++) score[i] = RC(0.0);
        if (weight_policy == LG_W_ALPHA || weight_policy == LG_W_ALPHA_T) fix = (uint64_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(uint64_t));
    }

    /* K6 / K6c: per-tile front-to-back blend */
    const int ntiles = gx * gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < ntiles; t++) {
        const int tx0 = (t % gx) * TILE, ty0 = (t / gx) * TILE;
        const uint32_t lo = ctx->range_lo[t], hi = ctx->range_hi[t];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int pxi = tx0 + lx, pyi = ty0 + ly;
                if (pxi >= W || pyi >= H) continue;
                const real pxf = (real)pxi, pyf = (real)pyi;
                real T = RC(1.0), C0 = RC(0.0), C1 = RC(0.0), C2 = RC(0.0);
                uint32_t contributor = 0, last = 0;
                for (uint32_t k = lo; k < hi; k++) {
                    contributor++;
                    const uint32_t g = ctx->point_list[k];
                    const real dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
                    const real ha = RC(-0.5) * conic_opacity[4 * g], nb = -conic_opacity[4 * g + 1],
                               hc = RC(-0.5) * conic_opacity[4 * g + 2], op = conic_opacity[4 * g + 3];
                    const real power = R_FMA(R_FMA(ha, dx, nb * dy), dx, (hc * dy) * dy);
                    if (power > RC(0.0)) continue;
                    const real alpha = R_FMIN(RC(0.99), op * lg_exp(power));
                    if (alpha < RC(1.0) / RC(255.0)) continue;
                    const real test_T = T * (RC(1.0) - alpha);
                    if (test_T < RC(0.0001)) break; /* done */
                    const real w = alpha * T;
                    C0 = R_FMA(rgb[3 * g], w, C0);
                    C1 = R_FMA(rgb[3 * g + 1], w, C1);
                    C2 = R_FMA(rgb[3 * g + 2], w, C2);
                    if (count) {
#pragma omp atomic
                        count[g] += 1;
                    }
                    if (fix) {
                        const float wv = (float)((weight_policy == LG_W_ALPHA) ? alpha : w);
                        const uint64_t q = lgo_fix40_quant(wv);
#pragma omp atomic
                        fix[g] += q;
                    }
                    T = test_T;
                    last = contributor;
                }
                const size_t pid = (size_t)pyi * W + pxi;
                final_T[pid] = T;
                n_contrib[pid] = (int)last;
                out_color[0 * (size_t)H * W + pid] = R_FMA(T, bg[0], C0);
                out_color[1 * (size_t)H * W + pid] = R_FMA(T, bg[1], C1);
                out_color[2 * (size_t)H * W + pid] = R_FMA(T, bg[2], C2);
            }
    }
    if (score && count && (weight_policy == LG_W_ONE || weight_policy == LG_W_OPACITY)) {
#pragma omp parallel for schedule(dynamic, 1024)
        for (int i = 0; i < N; i++)
            score[i] = lgo_seqsum(weight_policy == LG_W_ONE ? RC(1.0) : opacities[i], count[i]);
    }
    if (fix) {
        for (int i = 0; i < N; i++) score[i] = (real)lgo_fix40_score(fix[i]);
        free(fix);
    }
    return ctx;
}

/* ------------------------------------------------------------------------------------------ */
/* Backward.  dL_dcolor [3,H,W] -> dense grads (zero for non-rasterised Gaussians).            */
/* dL_dmeans2D is in NDC units, [N,3] with z = 0 (consumed by add_densification_stats,         */
/* scene/gaussian_model.py:784-788).                                                           */
void lgo_backward(const lgo_ctx *ctx, int N, int M, int D, int W, int H, const real *bg,
                  const real *means3D, const real *shs, const real *colors_precomp,
                  const real *scales, real scale_modifier, const real *rotations,
                  const real *cov3D_precomp, const real *viewmatrix, const real *projmatrix,
                  const real *campos, real tanfovx, real tanfovy, const int *radii, const real *xy,
                  const real *conic_opacity, const real *rgb, const real *cov3D,
                  const unsigned char *clamped, const real *final_T, const int *n_contrib,
                  const real *dL_dpix, real *dL_dmeans2D, real *dL_dmeans3D, real *dL_dshs,
                  real *dL_dcolors, real *dL_dopacity, real *dL_dscales, real *dL_drots,
                  real *dL_dcov3D)
{
    const int gx = ctx->gx, gy = ctx->gy;
    const uint64_t R = ctx->R;
    const size_t HW = (size_t)H * W;
    /* per-instance partials: [R][9] = d(mean2D pixel x,y), dconic A,B,C, dopacity, drgb[3] */
    real *part = (real *)calloc((R ? R : 1) * 9, sizeof(real));
    const int ntiles = gx * gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < ntiles; t++) {
        const int tx0 = (t % gx) * TILE, ty0 = (t / gx) * TILE;
        const uint32_t lo = ctx->range_lo[t];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int pxi = tx0 + lx, pyi = ty0 + ly;
                if (pxi >= W || pyi >= H) continue;
                const size_t pid = (size_t)pyi * W + pxi;
                const real pxf = (real)pxi, pyf = (real)pyi;
                const real T_final = final_T[pid];
                real T = T_final;
                const uint32_t last = (uint32_t)n_contrib[pid];
                const real g0 = dL_dpix[pid], g1 = dL_dpix[HW + pid], g2 = dL_dpix[2 * HW + pid];
                const real bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
                real acc0 = RC(0.0), acc1 = RC(0.0), acc2 = RC(0.0);
                real last_alpha = RC(0.0), lc0 = RC(0.0), lc1 = RC(0.0), lc2 = RC(0.0);
                for (uint32_t k = lo + last; k-- > lo;) {
                    const uint32_t g = ctx->point_list[k];
                    const real dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
                    const real A = conic_opacity[4 * g], B = conic_opacity[4 * g + 1], Cc = conic_opacity[4 * g + 2],
                               op = conic_opacity[4 * g + 3];
                    const real ha = RC(-0.5) * A, nb = -B, hc = RC(-0.5) * Cc;
                    const real power = R_FMA(R_FMA(ha, dx, nb * dy), dx, (hc * dy) * dy);
                    if (power > RC(0.0)) continue;
                    const real G = lg_exp(power);
                    const real alpha = R_FMIN(RC(0.99), op * G);
                    if (alpha < RC(1.0) / RC(255.0)) continue;
                    T = T / (RC(1.0) - alpha);
                    const real dch = alpha * T;
                    const real c0 = rgb[3 * g], c1 = rgb[3 * g + 1], c2 = rgb[3 * g + 2];
                    acc0 = last_alpha * lc0 + (RC(1.0) - last_alpha) * acc0;
                    acc1 = last_alpha * lc1 + (RC(1.0) - last_alpha) * acc1;
                    acc2 = last_alpha * lc2 + (RC(1.0) - last_alpha) * acc2;
                    lc0 = c0; lc1 = c1; lc2 = c2;
                    real dL_dalpha = (c0 - acc0) * g0 + (c1 - acc1) * g1 + (c2 - acc2) * g2;
                    dL_dalpha = dL_dalpha * T;
                    last_alpha = alpha;
                    dL_dalpha = dL_dalpha + (-T_final / (RC(1.0) - alpha)) * bg_dot;
                    const real dL_dG = op * dL_dalpha;
                    const real gdx = G * dx, gdy = G * dy;
                    const real dG_ddelx = -gdx * A - gdy * B;
                    const real dG_ddely = -gdy * Cc - gdx * B;
                    real *p = part + (size_t)k * 9;
                    p[0] += dL_dG * dG_ddelx;           /* d/d(mean2D pixel x) */
                    p[1] += dL_dG * dG_ddely;
                    p[2] += RC(-0.5) * gdx * dx * dL_dG; /* dA */
                    p[3] += -gdx * dy * dL_dG;           /* dB (full derivative) */
                    p[4] += RC(-0.5) * gdy * dy * dL_dG; /* dC */
                    p[5] += G * dL_dalpha;               /* dopacity */
                    p[6] += dch * g0; p[7] += dch * g1; p[8] += dch * g2;
                }
            }
    }
    /* deterministic gather: instances in sorted order */
    real *acc = (real *)calloc((size_t)N * 9, sizeof(real));
    for (uint64_t k = 0; k < R; k++) {
        const uint32_t g = ctx->point_list[k];
        for (int j = 0; j < 9; j++) acc[(size_t)g * 9 + j] += part[k * 9 + j];
    }
    free(part);

    const real fx = (real)W / (RC(2.0) * tanfovx), fy = (real)H / (RC(2.0) * tanfovy);
    const real *vm = viewmatrix, *pm = projmatrix;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 3; k++) { dL_dmeans2D[3 * i + k] = RC(0.0); dL_dmeans3D[3 * i + k] = RC(0.0); }
        dL_dopacity[i] = RC(0.0);
        if (dL_dcolors) for (int k = 0; k < 3; k++) dL_dcolors[3 * i + k] = RC(0.0);
        if (dL_dshs) for (int k = 0; k < 3 * M; k++) dL_dshs[(size_t)i * 3 * M + k] = RC(0.0);
        if (dL_dscales) for (int k = 0; k < 3; k++) dL_dscales[3 * i + k] = RC(0.0);
        if (dL_drots) for (int k = 0; k < 4; k++) dL_drots[4 * i + k] = RC(0.0);
        if (dL_dcov3D) for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = RC(0.0);
        if (!(radii[i] > 0)) continue;
        const real *a9 = acc + (size_t)i * 9;
        /* mean2D gradient in NDC units: pixel = ((ndc+1)*S-1)/2 */
        const real gndx = a9[0] * (RC(0.5) * (real)W), gndy = a9[1] * (RC(0.5) * (real)H);
        dL_dmeans2D[3 * i] = gndx; dL_dmeans2D[3 * i + 1] = gndy;
        dL_dopacity[i] = a9[5];
        const real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        real dmx = RC(0.0), dmy = RC(0.0), dmz = RC(0.0);

        /* ---- conic -> cov2D -> (cov3D, t) ---- */
        real vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        real vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        real vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        real T2[6], txc, tyc; int xc, yc;
        ewa_T(vm, vx, vy, vz, fx, fy, RC(1.3) * tanfovx, RC(1.3) * tanfovy, T2, &txc, &tyc, &xc, &yc);
        const real *S = cov3D + 6 * i;
        real U0 = T2[0] * S[0] + T2[1] * S[1] + T2[2] * S[2];
        real U1 = T2[0] * S[1] + T2[1] * S[3] + T2[2] * S[4];
        real U2 = T2[0] * S[2] + T2[1] * S[4] + T2[2] * S[5];
        real V0 = T2[3] * S[0] + T2[4] * S[1] + T2[5] * S[2];
        real V1 = T2[3] * S[1] + T2[4] * S[3] + T2[5] * S[4];
        real V2 = T2[3] * S[2] + T2[4] * S[4] + T2[5] * S[5];
        real a = U0 * T2[0] + U1 * T2[1] + U2 * T2[2] + RC(0.3);
        real b = U0 * T2[3] + U1 * T2[4] + U2 * T2[5];
        real c = V0 * T2[3] + V1 * T2[4] + V2 * T2[5] + RC(0.3);
        const real gA = a9[2], gB = a9[3], gC = a9[4];
        real denom = a * c - b * b;
        real d2inv = RC(1.0) / (denom * denom + RC(0.0000001));
        real dL_da = RC(0.0), dL_db = RC(0.0), dL_dc = RC(0.0);
        if (d2inv != RC(0.0)) {
            dL_da = d2inv * (-c * c * gA + b * c * gB + (denom - a * c) * gC);
            dL_dc = d2inv * (-a * a * gC + a * b * gB + (denom - a * c) * gA);
            dL_db = d2inv * (RC(2.0) * b * c * gA - (denom + RC(2.0) * b * b) * gB + RC(2.0) * a * b * gC);
        }
        /* dL/dSigma packed (off-diagonals appear twice) */
        real dS[6];
        dS[0] = T2[0] * T2[0] * dL_da + T2[0] * T2[3] * dL_db + T2[3] * T2[3] * dL_dc;
        dS[3] = T2[1] * T2[1] * dL_da + T2[1] * T2[4] * dL_db + T2[4] * T2[4] * dL_dc;
        dS[5] = T2[2] * T2[2] * dL_da + T2[2] * T2[5] * dL_db + T2[5] * T2[5] * dL_dc;
        dS[1] = RC(2.0) * T2[0] * T2[1] * dL_da + (T2[0] * T2[4] + T2[1] * T2[3]) * dL_db + RC(2.0) * T2[3] * T2[4] * dL_dc;
        dS[2] = RC(2.0) * T2[0] * T2[2] * dL_da + (T2[0] * T2[5] + T2[2] * T2[3]) * dL_db + RC(2.0) * T2[3] * T2[5] * dL_dc;
        dS[4] = RC(2.0) * T2[1] * T2[2] * dL_da + (T2[1] * T2[5] + T2[2] * T2[4]) * dL_db + RC(2.0) * T2[4] * T2[5] * dL_dc;
        if (d2inv == RC(0.0)) for (int k = 0; k < 6; k++) dS[k] = RC(0.0);
        /* dL/dT2 = 2 G2 T2 Sigma, G2 = [[da, db/2],[db/2, dc]]  ==  rows: 2*da*U + db*V ; db*U + 2*dc*V */
        real dT00 = RC(2.0) * dL_da * U0 + dL_db * V0, dT01 = RC(2.0) * dL_da * U1 + dL_db * V1, dT02 = RC(2.0) * dL_da * U2 + dL_db * V2;
        real dT10 = dL_db * U0 + RC(2.0) * dL_dc * V0, dT11 = dL_db * U1 + RC(2.0) * dL_dc * V1, dT12 = dL_db * U2 + RC(2.0) * dL_dc * V2;
        /* dL/dJ = dL/dT2 * Wm^T ; Wm[c][k] = vm[4k+c] */
        real dJ00 = dT00 * vm[0] + dT01 * vm[4] + dT02 * vm[8];
        real dJ02 = dT00 * vm[2] + dT01 * vm[6] + dT02 * vm[10];
        real dJ11 = dT10 * vm[1] + dT11 * vm[5] + dT12 * vm[9];
        real dJ12 = dT10 * vm[2] + dT11 * vm[6] + dT12 * vm[10];
        real tz = RC(1.0) / vz, tz2 = tz * tz, tz3 = tz2 * tz;
        real dtx = (xc ? RC(0.0) : RC(1.0)) * (-fx * tz2 * dJ02);
        real dty = (yc ? RC(0.0) : RC(1.0)) * (-fy * tz2 * dJ12);
        real dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (RC(2.0) * fx * txc) * tz3 * dJ02 + (RC(2.0) * fy * tyc) * tz3 * dJ12;
        /* t = p * viewmatrix (row-vector): dL/dp_k = sum_c vm[4k+c] dL/dt_c */
        dmx += vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
        dmy += vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        dmz += vm[8] * dtx + vm[9] * dty + vm[10] * dtz;

        /* ---- NDC mean gradient -> mean3D ---- */
        {
            real hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
            real hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
            real hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
            real m_w = RC(1.0) / (hw + RC(0.0000001));
            real mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
            dmx += (pm[0] * m_w - pm[3] * mul1) * gndx + (pm[1] * m_w - pm[3] * mul2) * gndy;
            dmy += (pm[4] * m_w - pm[7] * mul1) * gndx + (pm[5] * m_w - pm[7] * mul2) * gndy;
            dmz += (pm[8] * m_w - pm[11] * mul1) * gndx + (pm[9] * m_w - pm[11] * mul2) * gndy;
        }

        /* ---- colour ---- */
        if (colors_precomp) {
            if (dL_dcolors) for (int k = 0; k < 3; k++) dL_dcolors[3 * i + k] = a9[6 + k];
        } else if (shs) {
            real dRGB[3];
            for (int k = 0; k < 3; k++) dRGB[k] = clamped[3 * i + k] ? RC(0.0) : a9[6 + k];
            real ox = px - campos[0], oy = py - campos[1], oz = pz - campos[2];
            real len = R_SQRT(ox * ox + oy * oy + oz * oz);
            real x = ox / len, y = oy / len, z = oz / len;
            const real *sh = shs + (size_t)i * M * 3;
            real *dsh = dL_dshs + (size_t)i * M * 3;
            real dRdx[3] = {0, 0, 0}, dRdy[3] = {0, 0, 0}, dRdz[3] = {0, 0, 0};
#define SHV(k, c) sh[(k) * 3 + (c)]
            for (int cc = 0; cc < 3; cc++) {
                dsh[0 * 3 + cc] = SH_C0 * dRGB[cc];
                if (D > 0) {
                    dsh[1 * 3 + cc] = -SH_C1 * y * dRGB[cc];
                    dsh[2 * 3 + cc] = SH_C1 * z * dRGB[cc];
                    dsh[3 * 3 + cc] = -SH_C1 * x * dRGB[cc];
                    dRdx[cc] = -SH_C1 * SHV(3, cc);
                    dRdy[cc] = -SH_C1 * SHV(1, cc);
                    dRdz[cc] = SH_C1 * SHV(2, cc);
                    if (D > 1) {
                        real xx = x * x, yy = y * y, zz = z * z, xyv = x * y, yz = y * z, xz = x * z;
                        dsh[4 * 3 + cc] = SH_C2[0] * xyv * dRGB[cc];
                        dsh[5 * 3 + cc] = SH_C2[1] * yz * dRGB[cc];
                        dsh[6 * 3 + cc] = SH_C2[2] * (RC(2.0) * zz - xx - yy) * dRGB[cc];
                        dsh[7 * 3 + cc] = SH_C2[3] * xz * dRGB[cc];
                        dsh[8 * 3 + cc] = SH_C2[4] * (xx - yy) * dRGB[cc];
                        dRdx[cc] += SH_C2[0] * y * SHV(4, cc) + SH_C2[2] * RC(2.0) * -x * SHV(6, cc) +
                                    SH_C2[3] * z * SHV(7, cc) + SH_C2[4] * RC(2.0) * x * SHV(8, cc);
                        dRdy[cc] += SH_C2[0] * x * SHV(4, cc) + SH_C2[1] * z * SHV(5, cc) +
                                    SH_C2[2] * RC(2.0) * -y * SHV(6, cc) + SH_C2[4] * RC(2.0) * -y * SHV(8, cc);
                        dRdz[cc] += SH_C2[1] * y * SHV(5, cc) + SH_C2[2] * RC(2.0) * RC(2.0) * z * SHV(6, cc) +
                                    SH_C2[3] * x * SHV(7, cc);
                        if (D > 2) {
                            dsh[9 * 3 + cc] = SH_C3[0] * y * (RC(3.0) * xx - yy) * dRGB[cc];
                            dsh[10 * 3 + cc] = SH_C3[1] * xyv * z * dRGB[cc];
                            dsh[11 * 3 + cc] = SH_C3[2] * y * (RC(4.0) * zz - xx - yy) * dRGB[cc];
                            dsh[12 * 3 + cc] = SH_C3[3] * z * (RC(2.0) * zz - RC(3.0) * xx - RC(3.0) * yy) * dRGB[cc];
                            dsh[13 * 3 + cc] = SH_C3[4] * x * (RC(4.0) * zz - xx - yy) * dRGB[cc];
                            dsh[14 * 3 + cc] = SH_C3[5] * z * (xx - yy) * dRGB[cc];
                            dsh[15 * 3 + cc] = SH_C3[6] * x * (xx - RC(3.0) * yy) * dRGB[cc];
                            dRdx[cc] += SH_C3[0] * SHV(9, cc) * RC(3.0) * RC(2.0) * xyv + SH_C3[1] * SHV(10, cc) * yz +
                                        SH_C3[2] * SHV(11, cc) * -RC(2.0) * xyv +
                                        SH_C3[3] * SHV(12, cc) * -RC(3.0) * RC(2.0) * xz +
                                        SH_C3[4] * SHV(13, cc) * (-RC(3.0) * xx + RC(4.0) * zz - yy) +
                                        SH_C3[5] * SHV(14, cc) * RC(2.0) * xz +
                                        SH_C3[6] * SHV(15, cc) * RC(3.0) * (xx - yy);
                            dRdy[cc] += SH_C3[0] * SHV(9, cc) * RC(3.0) * (xx - yy) + SH_C3[1] * SHV(10, cc) * xz +
                                        SH_C3[2] * SHV(11, cc) * (-RC(3.0) * yy + RC(4.0) * zz - xx) +
                                        SH_C3[3] * SHV(12, cc) * -RC(3.0) * RC(2.0) * yz +
                                        SH_C3[4] * SHV(13, cc) * -RC(2.0) * xyv +
                                        SH_C3[5] * SHV(14, cc) * -RC(2.0) * yz +
                                        SH_C3[6] * SHV(15, cc) * -RC(3.0) * RC(2.0) * xyv;
                            dRdz[cc] += SH_C3[1] * SHV(10, cc) * xyv + SH_C3[2] * SHV(11, cc) * RC(4.0) * RC(2.0) * yz +
                                        SH_C3[3] * SHV(12, cc) * RC(3.0) * (RC(2.0) * zz - xx - yy) +
                                        SH_C3[4] * SHV(13, cc) * RC(4.0) * RC(2.0) * xz +
                                        SH_C3[5] * SHV(14, cc) * (xx - yy);
                        }
                    }
                }
            }
#undef SHV
            real ddx = dRdx[0] * dRGB[0] + dRdx[1] * dRGB[1] + dRdx[2] * dRGB[2];
            real ddy = dRdy[0] * dRGB[0] + dRdy[1] * dRGB[1] + dRdy[2] * dRGB[2];
            real ddz = dRdz[0] * dRGB[0] + dRdz[1] * dRGB[1] + dRdz[2] * dRGB[2];
            /* through dir/|dir| */
            real sum2 = ox * ox + oy * oy + oz * oz;
            real invsum32 = RC(1.0) / R_SQRT(sum2 * sum2 * sum2);
            dmx += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
            dmy += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
            dmz += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
        }

        /* ---- cov3D -> scale / rotation ---- */
        if (cov3D_precomp) {
            if (dL_dcov3D) for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = dS[k];
        } else {
            const real *q = rotations + 4 * i; const real *sc = scales + 3 * i;
            real s0 = scale_modifier * sc[0], s1 = scale_modifier * sc[1], s2 = scale_modifier * sc[2];
            real r = q[0], x = q[1], y = q[2], z = q[3];
            real Rm[9] = { RC(1.0) - RC(2.0) * (y * y + z * z), RC(2.0) * (x * y - r * z), RC(2.0) * (x * z + r * y),
                           RC(2.0) * (x * y + r * z), RC(1.0) - RC(2.0) * (x * x + z * z), RC(2.0) * (y * z - r * x),
                           RC(2.0) * (x * z - r * y), RC(2.0) * (y * z + r * x), RC(1.0) - RC(2.0) * (x * x + y * y) };
            real L[9], sv[3] = { s0, s1, s2 };
            for (int ii = 0; ii < 3; ii++) for (int jj = 0; jj < 3; jj++) L[3 * ii + jj] = Rm[3 * ii + jj] * sv[jj];
            /* full symmetric dL/dSigma */
            real Gs[9] = { dS[0], RC(0.5) * dS[1], RC(0.5) * dS[2], RC(0.5) * dS[1], dS[3], RC(0.5) * dS[4],
                           RC(0.5) * dS[2], RC(0.5) * dS[4], dS[5] };
            real dLm[9]; /* dL/dL = 2 Gs L */
            for (int ii = 0; ii < 3; ii++) for (int jj = 0; jj < 3; jj++)
                dLm[3 * ii + jj] = RC(2.0) * (Gs[3 * ii] * L[jj] + Gs[3 * ii + 1] * L[3 + jj] + Gs[3 * ii + 2] * L[6 + jj]);
            /* NOTE: like the published implementation, the scale gradient omits the scale_modifier factor */
            for (int jj = 0; jj < 3; jj++)
                dL_dscales[3 * i + jj] = dLm[jj] * Rm[jj] + dLm[3 + jj] * Rm[3 + jj] + dLm[6 + jj] * Rm[6 + jj];
            real g[9];
            for (int ii = 0; ii < 3; ii++) for (int jj = 0; jj < 3; jj++) g[3 * ii + jj] = dLm[3 * ii + jj] * sv[jj];
            dL_drots[4 * i + 0] = RC(2.0) * (-z * g[1] + y * g[2] + z * g[3] - x * g[5] - y * g[6] + x * g[7]);
            dL_drots[4 * i + 1] = RC(2.0) * (y * g[1] + z * g[2] + y * g[3] - RC(2.0) * x * g[4] - r * g[5] + z * g[6] + r * g[7] - RC(2.0) * x * g[8]);
            dL_drots[4 * i + 2] = RC(2.0) * (-RC(2.0) * y * g[0] + x * g[1] + r * g[2] + x * g[3] + z * g[5] - r * g[6] + z * g[7] - RC(2.0) * y * g[8]);
            dL_drots[4 * i + 3] = RC(2.0) * (-RC(2.0) * z * g[0] - r * g[1] + x * g[2] + r * g[3] - RC(2.0) * z * g[4] + y * g[5] + x * g[6] + y * g[7]);
        }
        dL_dmeans3D[3 * i] = dmx; dL_dmeans3D[3 * i + 1] = dmy; dL_dmeans3D[3 * i + 2] = dmz;
    }
    free(acc);
}

int lgo_real_bytes(void) { return (int)sizeof(real); }

/* ---- small hooks so that tests can pin the reference-owned pieces against tests/golden ---- */
void lgo_cov3d(int n, const real *scales, real mod, const real *rots, real *cov6)
{
    for (int i = 0; i < n; i++) cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov6 + 6 * i);
}
/* sh [n][M][3], unit dirs [n][3] -> raw SH value (no +0.5, no clamp) [n][3] */
void lgo_sh_eval(int n, int deg, int M, const real *sh, const real *dirs, real *out)
{
    for (int i = 0; i < n; i++) sh_basis_eval(deg, M, sh + (size_t)i * M * 3, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], out + 3 * i);
}
