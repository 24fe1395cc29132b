// tools/ubench/lds_atomic.hip -- throughput of LDS ds_or_b32 (no return) vs plain ds_write_b32 on gfx950, at the occupancy of the
// neighbour-gather kernel (8 workgroups x 256 threads per CU).  hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256, 8) void k(const unsigned* __restrict__ idx, int per_lane, int active_mod, unsigned* out)
{
    __shared__ unsigned bm[2048];   // 8 KiB
    for (int i = threadIdx.x; i < 2048; i += 256) bm[i] = 0;
    __syncthreads();
    const unsigned* p = idx + (size_t)blockIdx.x * 256 * per_lane + threadIdx.x;
    unsigned v[8];
    for (int j0 = 0; j0 < per_lane; j0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(j0 + u) * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned i = v[u] & 0xffff;
            const bool act = (v[u] >> 16) % active_mod == 0;
            if (act) {
                if (MODE == 0) atomicOr(&bm[i >> 5], 1u << (i & 31));
                else if (MODE == 1) bm[i >> 5] = 1u << (i & 31);
                else if (MODE == 2) __hip_atomic_fetch_or(&bm[i >> 5], 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < 2048; i += 256) s += __popc(bm[i]);
    if (s == 0xdeadbeef) out[blockIdx.x] = s;
}

int main()
{
    const int WG = 256 * 8 * 4, per_lane = 64;
    const size_t n = (size_t)WG * 256 * per_lane;
    unsigned* h = (unsigned*)malloc(n * 4);
    unsigned long long x = 88172645463325252ULL;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (unsigned)(x >> 11); }
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, WG * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mod = 1; mod <= 4; mod *= 2)
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(WG), dim3(256), 0, 0, d, per_lane, mod, o);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(WG), dim3(256), 0, 0, d, per_lane, mod, o);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(WG), dim3(256), 0, 0, d, per_lane, mod, o);
                else hipLaunchKernelGGL(k<3>, dim3(WG), dim3(256), 0, 0, d, per_lane, mod, o);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const double ops = (double)n / mod;
            printf("active 1/%d mode %d (%s): %.3f ms  -> %.2f lane-ops/clk/CU (2.4 GHz, 256 CUs)\n", mod, mode,
                   mode == 0 ? "atomicOr" : mode == 1 ? "plain store" : mode == 2 ? "wg-scope fetch_or" : "loads only", best,
                   ops / (best * 1e-3) / 2.4e9 / 256);
        }
    return 0;
}
