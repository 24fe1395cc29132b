// micro-benchmark 2: what slows a v_mfma_f32_16x16x4_f32 stream down?  modes:
//  0 same A/B registers   1 distinct A regs (f32x4 per tile), 4 B regs   2 mode 1 + A re-read from LDS every tile (ds_read_b128)
//  3 mode 2 + u16 row-offset indirection   4 mode 1 with dependent chains of 4 on one accumulator (tile-major order)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int T = 8;   // tiles (accumulators)

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ in, float* __restrict__ out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[600 * 24];
    constexpr int RS = (MODE == 8 || MODE == 9) ? 24 : 20;   // row stride in floats
    __shared__ unsigned short ro[64 * T];
    for (int i = threadIdx.x; i < 600 * 24; i += blockDim.x) lds[i] = in[i & 1023];
    for (int i = threadIdx.x; i < 64 * T; i += blockDim.x) ro[i] = (unsigned short)((MODE == 6 || MODE == 9 || MODE >= 10) ? ((i >> 6) * 16 + (i & 15)) : ((i * 7) % 560));
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, kk = lane >> 4;
    f32x4 acc[T], a[T];
    float b[4];
    for (int i = 0; i < 4; ++i) b[i] = in[lane + i * 64];
    for (int t = 0; t < T; ++t) { acc[t] = (f32x4){0, 0, 0, 0}; a[t] = *reinterpret_cast<const f32x4*>(&in[(lane * 4 + t * 16) & 1020]); }
    const float* lb = lds + kk * 4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].x, b[0], acc[t], 0, 0, 0);
        } else if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b[3], acc[t], 0, 0, 0);
                if (MODE == 4) __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 12) {
            // one A read feeds FOUR column tiles (16 MFMAs); 16 B registers
            constexpr int PD = 1, TR = T / 4;
            f32x4 av[TR];
            float bb[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) bb[i] = b[i & 3] + (float)i;
            av[0] = *reinterpret_cast<const f32x4*>(lb + ((int)ro[li] + (it & 7)) * RS);
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                if (t + PD < TR) av[t + PD] = *reinterpret_cast<const f32x4*>(lb + ((int)ro[(t + PD) * 64 + li] + (it & 7)) * RS);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[4 * t + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].x, bb[4 * n + 0], acc[4 * t + n], 0, 0, 0);
                    acc[4 * t + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].y, bb[4 * n + 1], acc[4 * t + n], 0, 0, 0);
                    acc[4 * t + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].z, bb[4 * n + 2], acc[4 * t + n], 0, 0, 0);
                    acc[4 * t + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].w, bb[4 * n + 3], acc[4 * t + n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 10 || MODE == 11) {
            // one A read feeds TWO column tiles (8 MFMAs): half the LDS traffic per MFMA.  T/2 row tiles x 2 accumulators
            constexpr int PD = 2, TR = T / 2;
            f32x4 av[TR];
#pragma unroll
            for (int t = 0; t < PD; ++t) av[t] = *reinterpret_cast<const f32x4*>(lb + ((int)ro[t * 64 + li] + (it & 7)) * RS);
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                if (t + PD < TR) av[t + PD] = *reinterpret_cast<const f32x4*>(lb + ((int)ro[(t + PD) * 64 + li] + (it & 7)) * RS);
                __builtin_amdgcn_sched_barrier(0);
                acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].x, b[0], acc[2 * t], 0, 0, 0);
                acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].y, b[1], acc[2 * t], 0, 0, 0);
                acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].z, b[2], acc[2 * t], 0, 0, 0);
                acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].w, b[3], acc[2 * t], 0, 0, 0);
                if (MODE == 11) __builtin_amdgcn_sched_barrier(0);
                acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].x, b[3], acc[2 * t + 1], 0, 0, 0);
                acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].y, b[2], acc[2 * t + 1], 0, 0, 0);
                acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].z, b[1], acc[2 * t + 1], 0, 0, 0);
                acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].w, b[0], acc[2 * t + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE >= 5) {
            // pinned: A of tile t+PD requested, then the 4 dependent MFMAs of tile t back-to-back
            constexpr int PD = MODE == 7 ? 3 : 2;
            constexpr bool IND = MODE == 6 || MODE == 9;
            f32x4 av[T];
#pragma unroll
            for (int t = 0; t < PD; ++t) {
                int r = IND ? (int)ro[t * 64 + li] + (it & 7) : ((li + it * 3 + t * 16) % 560);
                av[t] = *reinterpret_cast<const f32x4*>(lb + r * RS);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t + PD < T) {
                    int r = IND ? (int)ro[(t + PD) * 64 + li] + (it & 7) : ((li + it * 3 + (t + PD) * 16) % 560);
                    av[t + PD] = *reinterpret_cast<const f32x4*>(lb + r * RS);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].x, b[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].y, b[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].z, b[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].w, b[3], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int r = MODE == 3 ? (int)ro[t * 64 + ((li + it) & 63)] : ((li + it * 3 + t * 16) % 560);
                f32x4 av = *reinterpret_cast<const f32x4*>(lb + r * 20);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[3], acc[t], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int t = 0; t < T; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int wgs, const float* d, float* out)
{
    const int iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs), dim3(512), 0, 0, d, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs), dim3(512), 0, 0, d, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)iters * T * 4 * 2048.0 * 256 * wgs * 8;
    printf("mode %d, %d WG(512 thr)/CU: %.1f TF\n", MODE, wgs, flops / (ms * 1e-3) / 1e12);
}

int main()
{
    std::vector<float> h(4096);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    float *d, *out;
    (void)hipMalloc(&d, 4096 * 4); (void)hipMalloc(&out, 256 * 4 * 512 * 4);
    (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) for (int wgs : {1, 2}) { run<0>(wgs, d, out); run<4>(wgs, d, out); run<6>(wgs, d, out); run<9>(wgs, d, out); run<10>(wgs, d, out); run<12>(wgs, d, out); }
    return 0;
}
