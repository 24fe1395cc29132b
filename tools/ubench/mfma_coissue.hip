// micro-benchmark 4: does the VALU / LDS work of ONE wave run under the MFMAs of the OTHER wave of the same SIMD?
// A workgroup of 8 waves (two per SIMD, 1 workgroup per CU, 256 workgroups): waves 0..3 are "matrix" waves (a stream of independent
// v_mfma_f32_16x16x4_f32 on two alternating accumulators), waves 4..7 are "vector" waves (the F(4x4) input transform's instruction mix:
// LDS reads, fmaf chains, LDS writes).  Modes: 1 matrix waves only, 2 vector waves only, 3 both at once.  If the two kinds of work
// overlap, time(3) ~ max(time(1), time(2)); if a VALU / LDS instruction of the vector wave takes issue time from the matrix wave,
// time(3) ~ time(1) + time(2).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ in, float* __restrict__ out, int iters, int mode, int vwork)
{
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float res = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            const float x = in[lane], y = in[lane + 64];
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 72; ++u) {              // 144 MFMAs per iteration = one chunk of the F(4x4) kernel per wave
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                }
            }
            res = a0.x + a1.y;
        }
    } else {
        if (mode & 2) {
            float* w = lds + (wave - 4) * 4096 + lane;
            float acc = 0.f;
            for (int it = 0; it < iters; ++it) {
                for (int r = 0; r < vwork; ++r) {           // vwork x (36 LDS reads, 144 fmaf, 36 LDS writes) = vwork transform tasks per iteration
                    float t[36];
#pragma unroll
                    for (int i = 0; i < 36; ++i) t[i] = w[i * 64];
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int i = 0; i < 36; ++i) t[i] = fmaf(t[i], 1.0001f, t[(i + 7) % 36]);
#pragma unroll
                    for (int i = 0; i < 36; ++i) w[i * 64] = t[i];
                    acc += t[0];
                }
            }
            res = acc;
        }
    }
    if (res == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

int main()
{
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 37) % 101) * 0.01f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // loop time of 200 iterations = time(400 iterations) - time(200 iterations): launch, LDS fill and tail cancel
    for (int vwork = 1; vwork <= 4; ++vwork) {
        double d[4] = {0, 0, 0, 0};
        for (int mode = 1; mode <= 3; ++mode) {
            double best[2] = {1e9, 1e9};
            for (int rep = 0; rep < 5; ++rep)
                for (int h = 0; h < 2; ++h) {
                    hipEventRecord(a);
                    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, 200 * (h + 1), mode, vwork);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float t; hipEventElapsedTime(&t, a, b);
                    if (rep > 0 && t < best[h]) best[h] = t;
                }
            d[mode] = best[1] - best[0];
        }
        // per iteration and SIMD: 144 MFMAs x 32 cycles = 4 608 cycles of the matrix pipe
        printf("vector work x%d per iteration, 200 iterations: matrix only %.3f ms (%.0f ns / iteration), vector only %.3f ms, both %.3f ms -> both / max = %.2f, both / sum = %.2f\n",
               vwork, d[1], d[1] * 1e6 / 200, d[2], d[3], d[3] / (d[1] > d[2] ? d[1] : d[2]), d[3] / (d[1] + d[2]));
    }
    return 0;
}
