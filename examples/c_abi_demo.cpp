// The drop-in boundary without PyTorch: a plain C++ host program over include/lightgaussian.h (C ABI) and the HIP runtime.
// Renders a small synthetic scene forward (+ count) and backward, and checks hit counts / image against the CPU oracle
// library when it is present (oracle/liblg_oracle_f32.so is test infrastructure: this demo only uses it as the checker).
//
//   hipcc --offload-arch=gfx950 -O2 examples/c_abi_demo.cpp -Iinclude -Llightgaussian_amd -llightgaussian_hip \
//         -Wl,-rpath,$PWD/lightgaussian_amd -o /tmp/c_abi_demo && /tmp/c_abi_demo
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lightgaussian.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define LGCHECK(x) do { int rc_ = (x); if (rc_ != LG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, lg_last_error()); return 3; } } while (0)

static void* g_binning = nullptr;
static void* alloc_cb(void*, size_t nbytes)   // the role of the torch extension's resize callbacks
{
    if (g_binning) (void)hipFree(g_binning);
    g_binning = nullptr;
    if (hipMalloc(&g_binning, nbytes ? nbytes : 1) != hipSuccess) return nullptr;
    return g_binning;
}
template <class T> static T* dev(const std::vector<T>& h)
{
    T* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); }

int main()
{
    const int N = 20000, W = 320, H = 200, M = 1;
    unsigned seed = 12345u;
    std::vector<float> xyz(3 * N), scl(3 * N), rot(4 * N), opa(N), shs(3 * N);
    for (int i = 0; i < N; i++) {
        xyz[3 * i] = 4.0f * frand(seed) - 2.0f; xyz[3 * i + 1] = 2.4f * frand(seed) - 1.2f; xyz[3 * i + 2] = 4.0f * frand(seed) - 2.0f;
        for (int k = 0; k < 3; k++) scl[3 * i + k] = 0.01f + 0.05f * frand(seed);
        float q[4], n2 = 0; for (int k = 0; k < 4; k++) { q[k] = frand(seed) - 0.5f; n2 += q[k] * q[k]; }
        for (int k = 0; k < 4; k++) rot[4 * i + k] = q[k] / std::sqrt(n2 + 1e-12f);
        opa[i] = 0.05f + 0.9f * frand(seed);
        for (int k = 0; k < 3; k++) shs[3 * i + k] = (frand(seed) - 0.5f) / 0.28209479177387814f;
    }
    // camera at (0, 0, -5) looking along +z, 60 degree horizontal field of view; matrices in the reference's row-vector
    // convention (scene/cameras.py:70-85: world_view_transform = W2C^T, full_proj = world_view x projection)
    const float tanx = std::tan(0.5f * 1.0471975512f), tany = tanx * (float)H / (float)W, zn = 0.01f, zf = 100.0f;
    std::vector<float> view = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 5, 1};
    std::vector<float> P = {1 / tanx, 0, 0, 0, 0, 1 / tany, 0, 0, 0, 0, zf / (zf - zn), 1, 0, 0, -(zf * zn) / (zf - zn), 0};
    std::vector<float> proj(16, 0.0f);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) proj[4 * r + c] += view[4 * r + k] * P[4 * k + c];
    std::vector<float> campos = {0, 0, -5}, bg = {0.1f, 0.2f, 0.3f};

    float *d_xyz = dev(xyz), *d_scl = dev(scl), *d_rot = dev(rot), *d_opa = dev(opa), *d_shs = dev(shs), *d_view = dev(view), *d_proj = dev(proj),
          *d_cam = dev(campos), *d_bg = dev(bg);
    void *geom = nullptr, *img = nullptr;
    float* color = nullptr; int32_t *radii = nullptr, *count = nullptr; float* score = nullptr;
    HIPCHECK(hipMalloc(&geom, lg_geom_bytes(N))); HIPCHECK(hipMalloc(&img, lg_img_bytes(W, H)));
    HIPCHECK(hipMalloc((void**)&color, 3 * W * H * 4)); HIPCHECK(hipMalloc((void**)&radii, N * 4));
    HIPCHECK(hipMalloc((void**)&count, N * 4)); HIPCHECK(hipMalloc((void**)&score, N * 4));
    hipStream_t stream; HIPCHECK(hipStreamCreate(&stream));

    lg_view v; memset(&v, 0, sizeof(v));
    v.image_height = H; v.image_width = W; v.tanfovx = tanx; v.tanfovy = tany; v.bg = d_bg; v.scale_modifier = 1.0f;
    v.viewmatrix = d_view; v.projmatrix = d_proj; v.sh_degree = 0; v.campos = d_cam; v.prefiltered = 0; v.flags = 0;
    lg_gaussians g; memset(&g, 0, sizeof(g));
    g.N = N; g.M = M; g.means3D = d_xyz; g.shs = d_shs; g.opacities = d_opa; g.scales = d_scl; g.rotations = d_rot;

    void* bin = nullptr; int64_t R = 0;
    LGCHECK(lg_forward_count(&v, &g, geom, img, alloc_cb, nullptr, LG_WEIGHT_OPACITY, color, radii, count, score, &bin, &R, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    std::vector<int32_t> h_count(N); std::vector<float> h_color(3 * W * H);
    HIPCHECK(hipMemcpy(h_count.data(), count, N * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h_color.data(), color, 3 * W * H * 4, hipMemcpyDeviceToHost));
    long long hits = 0; for (int c : h_count) hits += c;
    double mean = 0; for (float c : h_color) mean += c; mean /= h_color.size();
    printf("lg_forward_count: abi %d, %lld tile instances, %lld pixel hits, mean colour %.6f\n", lg_abi_version(), (long long)R, hits, mean);

    // ABI 7: the running hit count (lg_view.count_sum) and a per-hit weight policy (exact Q24.40 sums: two runs give the same bits).
    // Two more count forwards into a zeroed accumulator: it must hold twice the per-view counts; the ALPHA_T scores of both runs are equal
    int32_t* csum = nullptr; float *score_a = nullptr, *score_b = nullptr;
    HIPCHECK(hipMalloc((void**)&csum, N * 4)); HIPCHECK(hipMemset(csum, 0, N * 4));
    HIPCHECK(hipMalloc((void**)&score_a, N * 4)); HIPCHECK(hipMalloc((void**)&score_b, N * 4));
    lg_view va = v; va.count_sum = csum;
    LGCHECK(lg_forward_count(&va, &g, geom, img, alloc_cb, nullptr, LG_WEIGHT_ALPHA_T, color, radii, count, score_a, &bin, &R, stream));
    LGCHECK(lg_forward_count(&va, &g, geom, img, alloc_cb, nullptr, LG_WEIGHT_ALPHA_T, color, radii, count, score_b, &bin, &R, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    std::vector<int32_t> h_csum(N), h_cnt2(N); std::vector<float> h_sa(N), h_sb(N);
    HIPCHECK(hipMemcpy(h_csum.data(), csum, N * 4, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(h_cnt2.data(), count, N * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h_sa.data(), score_a, N * 4, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(h_sb.data(), score_b, N * 4, hipMemcpyDeviceToHost));
    double wsum = 0; int same = 1;
    for (int i = 0; i < N; i++) {
        same &= (h_csum[i] == 2 * h_count[i]) && (h_cnt2[i] == h_count[i]) && (memcmp(&h_sa[i], &h_sb[i], 4) == 0) && ((h_count[i] == 0) == (h_sa[i] == 0.0f));
        wsum += h_sa[i];
    }
    printf("count_sum + LG_WEIGHT_ALPHA_T: running count = 2 x per-view count and scores reproducible: %s; sum of alpha T weights %.6f\n", same ? "yes" : "NO", wsum);
    if (!same || !(wsum > 0)) { fprintf(stderr, "unexpected result\n"); return 1; }

    // backward of sum(color) through a plain render
    LGCHECK(lg_forward(&v, &g, geom, img, alloc_cb, nullptr, color, radii, &bin, &R, stream));
    std::vector<float> ones(3 * W * H, 1.0f); float* dL = dev(ones);
    float *g2 = nullptr, *g3 = nullptr, *gsh = nullptr, *gop = nullptr, *gsc = nullptr, *grt = nullptr; void* scratch = nullptr;
    HIPCHECK(hipMalloc((void**)&g2, N * 12)); HIPCHECK(hipMalloc((void**)&g3, N * 12)); HIPCHECK(hipMalloc((void**)&gsh, N * 12));
    HIPCHECK(hipMalloc((void**)&gop, N * 4)); HIPCHECK(hipMalloc((void**)&gsc, N * 12)); HIPCHECK(hipMalloc((void**)&grt, N * 16));
    HIPCHECK(hipMalloc(&scratch, lg_backward_scratch_bytes(N, R)));
    LGCHECK(lg_backward(&v, &g, radii, geom, bin, img, R, dL, g2, g3, gsh, nullptr, gop, gsc, grt, nullptr, nullptr, scratch, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    std::vector<float> h_gop(N); HIPCHECK(hipMemcpy(h_gop.data(), gop, N * 4, hipMemcpyDeviceToHost));
    double gsum = 0; int finite = 1; for (float x : h_gop) { gsum += std::fabs(x); finite &= std::isfinite(x) ? 1 : 0; }
    printf("lg_backward: sum |dL/dopacity| = %.6f, all finite: %s\n", gsum, finite ? "yes" : "NO");
    if (!finite || hits <= 0 || !(gsum > 0)) { fprintf(stderr, "unexpected result\n"); return 1; }
    // invalid argument combinations fail before any launch, with the reference's message
    lg_gaussians bad = g; bad.colors_precomp = d_shs;
    if (lg_forward(&v, &bad, geom, img, alloc_cb, nullptr, color, radii, &bin, &R, stream) != LG_ERR_INVALID_ARGUMENT) return 1;
    printf("invalid-argument path: \"%s\"\nC ABI demo OK\n", lg_last_error());
    return 0;
}
