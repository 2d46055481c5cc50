// gfx950 global atomic throughput on random addresses (round 6): what a (wave, Gaussian) significance atomic costs.
//   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(uint32_t n_slots, uint32_t* a32, unsigned long long* a64, uint32_t* b32, uint32_t* c32, float* f32, uint32_t active_mod)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (active_mod > 1 && (hash(t * 3u + 1u) % active_mod) != 0) return;      // a fraction of the lanes issue (as in K6c: lane < nhit && cnt > 0)
    const uint32_t id = hash(t) % n_slots;
    if (MODE == 0) atomicAdd(&a32[id], 1u);
    if (MODE == 1) atomicAdd(&a64[id], 0x10000000123ull);
    if (MODE == 2) { atomicAdd(&a32[id], 1u); atomicAdd(&a64[id], 0x10000000123ull); }
    if (MODE == 3) { atomicAdd(&a32[id], 1u); atomicAdd(&b32[id], 7u); }
    if (MODE == 4) { atomicAdd(&a32[id], 1u); atomicAdd(&b32[id], 7u); atomicAdd(&c32[id], 9u); }
    if (MODE == 5) { atomicAdd(&a32[id], 1u); atomicAdd(&f32[id], 0.37f); }
    if (MODE == 6) { atomicAdd(&a32[2 * id], 1u); atomicAdd(&a32[2 * id + 1], 7u); }              // two words of one 8-byte slot
    if (MODE == 7) { atomicAdd(&a32[4 * id], 1u); atomicAdd(&a32[4 * id + 1], 7u); atomicAdd(&a32[4 * id + 2], 7u); }   // three words of one 16-byte slot
}
int main()
{
    const uint32_t n_slots = 3000000, n_ops = 4800000 * 3;   // one of three lanes active
    uint32_t *a32, *b32, *c32; unsigned long long* a64; float* f32;
    hipMalloc(&a32, 16ull * n_slots); hipMalloc(&b32, 4ull * n_slots); hipMalloc(&c32, 4ull * n_slots); hipMalloc(&a64, 8ull * n_slots); hipMalloc(&f32, 4ull * n_slots);
    hipMemset(a32, 0, 16ull * n_slots); hipMemset(b32, 0, 4ull * n_slots); hipMemset(c32, 0, 4ull * n_slots); hipMemset(a64, 0, 8ull * n_slots); hipMemset(f32, 0, 4ull * n_slots);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"u32", "u64", "u32 + u64", "u32 + u32 (two arrays)", "3 x u32 (three arrays)", "u32 + f32", "2 x u32 in one 8-byte slot", "3 x u32 in one 16-byte slot"};
    for (int mode = 0; mode < 8; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; it++) {
                const dim3 g(n_ops / 256), b(256);
                switch (mode) {
                case 0: k<0><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                case 1: k<1><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                case 2: k<2><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                case 3: k<3><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                case 4: k<4><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                case 5: k<5><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                case 6: k<6><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                case 7: k<7><<<g, b>>>(n_slots, a32, a64, b32, c32, f32, 3); break;
                }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-32s %8.1f us per launch of %.1f M active lanes (%.2f ns per lane)\n", names[mode], ms * 100.0f, n_ops / 3e6, ms * 1e5 / (n_ops / 3.0));
        }
    }
    return 0;
}
