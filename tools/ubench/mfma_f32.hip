// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 on gfx950 (cycles per MFMA per SIMD) for
// {1,2,4} waves per SIMD, {2,4,8} independent accumulators, zero vs random operands.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(const float* __restrict__ in, float* __restrict__ out, int iters, long long* cyc)
{
    f32x4 acc[NACC];
    float a = in[threadIdx.x], b = in[threadIdx.x + 64];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){a, b, a, b};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int wps, bool rnd, const float* dz, const float* dr, float* out, long long* dcyc)
{
    const int iters = 2000;
    const int threads = 256 * wps;   // wps waves per SIMD, one workgroup per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, rnd ? dr : dz, out, 10, dcyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, rnd ? dr : dz, out, iters, dcyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    double nm = (double)iters * 4 * NACC;            // MFMAs per wave
    double flops = nm * 2048.0 * 256 * 4 * wps;
    printf("nacc=%d waves/SIMD=%d data=%s: %.1f cyc/MFMA/wave (x%d waves => %.1f cyc/MFMA/SIMD), %.1f TF, clock-from-counter %.2f GHz\n", NACC, wps,
           rnd ? "random" : "zero", cyc / nm, wps, cyc / nm / wps, flops / (ms * 1e-3) / 1e12, cyc / (ms * 1e-3) / 1e9);
}

int main()
{
    std::vector<float> h(1024);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *dz, *dr, *out; long long* dcyc;
    hipMalloc(&dz, 4096); hipMalloc(&dr, 4096); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&dcyc, 8);
    hipMemset(dz, 0, 4096); hipMemcpy(dr, h.data(), 4096, hipMemcpyHostToDevice);
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int wps : {1, 2, 4}) {
            run<2>(wps, rnd, dz, dr, out, dcyc);
            run<4>(wps, rnd, dz, dr, out, dcyc);
            run<8>(wps, rnd, dz, dr, out, dcyc);
        }
    return 0;
}
