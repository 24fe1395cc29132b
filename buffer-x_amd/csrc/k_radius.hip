// k_radius.hip -- density-aware radius estimation (reference models/BUFFERX.py:627-696, :610-624).
//
// The reference materialises the nk x N squared-distance matrix once per scale and runs <= 13 bisection
// steps, each a full reduction plus a host sync.  Observation: the bisection on [0,5] only ever visits
// r = 5*m/8192, so one pass that histograms every d2 against the 8193 thresholds fp32(r_m^2) answers every
// possible count query; the bisection itself then runs on-device on the prefix sums (no host round trip),
// once per threshold, and the histogram is shared by all scales of a pair.
//   d2 = (|k|^2 + |p|^2) - 2 k.p   in fp32, same operation order as squared_cdist / oracle bxo_radius.
#include "bx_common.h"

namespace {
constexpr int NB = 8194;  // bins 0..8192 = smallest m with d2 < thr[m]; 8193 = beyond max_r

// threshold m of the bisection grid, fp32((5 m / 8192)^2) exactly as the host table d_rad_thr holds it: recomputed (two exact
// binary64 products, one rounding) instead of loaded -- the two dependent loads per distance were what the kernel waited for.
// (Round 4: the all-fp32 form (float)m * (5.0f / 8192.0f) squared is the same number for every m -- checked exhaustively -- and was
// measured 26-35 % SLOWER, 387 / 413 vs 306 us per launch; 16-bit LDS counters, two per word, for eight instead of four resident
// workgroups per CU: 306 vs 312 us.  Neither kept; tests/test_gpu_stages.py::test_radius_concentrated_bins stays.)
__device__ __forceinline__ float thr_of(int m)
{
    const double r = 5.0 * (double)m / 8192.0;
    return (float)(r * r);
}

// Round 6: 1024-thread workgroups (RH_T) share one LDS histogram: the 33 KB that capped a CU at four 256-thread workgroups (16 waves)
// now carry two 1024-thread ones (32 waves: the per-distance bin search is latency-bound, occupancy is what it wants) and the global flush
// -- 8194 64-bit atomics per workgroup -- happens 512 instead of 2048 times per launch; and the bin search starts at floor(t) + 1, where
// the answer almost always is (d2 < thr[m] <=> t < m up to the rounding of thr), so the common case costs two threshold evaluations
// instead of three.  The two correcting loops are unchanged: the bin is exact from any start.
constexpr int RH_T = 1024;
__global__ __launch_bounds__(RH_T) void radius_hist_kernel(const float* __restrict__ pts, int n_pts,
                                                          const float* __restrict__ kpts, int nk,
                                                          const float* __restrict__ thr, unsigned long long* hist)
{
    __shared__ unsigned h[NB];
    __shared__ float sk[64][4];
    for (int i = threadIdx.x; i < NB; i += RH_T) h[i] = 0;
    const int k0 = blockIdx.y * 64;
    const int kc = min(64, nk - k0);
    if (threadIdx.x < 64) {
        int q = threadIdx.x;
        float kx = 0.f, ky = 0.f, kz = 0.f;
        if (q < kc) { kx = kpts[(size_t)(k0 + q) * 3]; ky = kpts[(size_t)(k0 + q) * 3 + 1]; kz = kpts[(size_t)(k0 + q) * 3 + 2]; }
        sk[q][0] = kx; sk[q][1] = ky; sk[q][2] = kz;
        sk[q][3] = (kx * kx + ky * ky) + kz * kz;
    }
    __syncthreads();
    for (int j = blockIdx.x * RH_T + threadIdx.x; j < n_pts; j += gridDim.x * RH_T) {
        float px = pts[(size_t)j * 3], py = pts[(size_t)j * 3 + 1], pz = pts[(size_t)j * 3 + 2];
        float y2 = (px * px + py * py) + pz * pz;
        for (int q = 0; q < kc; ++q) {
            float xy = (sk[q][0] * px + sk[q][1] * py) + sk[q][2] * pz;
            float d2 = (sk[q][3] + y2) - 2.0f * xy;
            if (!(d2 <= 25.0f)) continue;  // dists_sqr[dists_sqr <= max_r*max_r]
            int m = (int)(sqrtf(fmaxf(d2, 0.0f)) * 1638.4f) + 1;
            m = m < 0 ? 0 : (m > 8192 ? 8192 : m);
            while (m <= 8192 && !(d2 < thr_of(m))) ++m;
            while (m > 0 && d2 < thr_of(m - 1)) --m;
            atomicAdd(&h[m], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += RH_T)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// prefix sums + the literal bisection loop, one workgroup
struct BisectThr { double t[BX_MAX_SCALES]; int n; };

// the bisections of all scales of a pair share the histogram: one launch, thread i < nthr runs the loop of threshold i
__global__ __launch_bounds__(1024) void radius_bisect_kernel(const unsigned long long* __restrict__ hist, long long n_orig,
                                                             int nk, BisectThr thr, double* des_r_out)
{
    __shared__ unsigned long long cum[NB];
    __shared__ unsigned long long part[1024];
    // cum[m] = #(d2 < thr[m]) = sum_{b<=m} hist[b]
    const int per = (NB + 1023) / 1024;
    unsigned long long s = 0;
    for (int i = 0; i < per; ++i) {
        int b = threadIdx.x * per + i;
        if (b < NB) s += hist[b];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { unsigned long long v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int i = 0; i < per; ++i) {
        int b = threadIdx.x * per + i;
        if (b < NB) { run += hist[b]; cum[b] = run; }
    }
    __syncthreads();
    if ((int)threadIdx.x < thr.n) {
        const double threshold = thr.t[threadIdx.x];
        const double tol = 0.01;
        double low = 0.0, high = 5.0, des_r = 0.0;
        float den = (float)((double)n_orig * (double)nk);
        while (high - low > 1e-3) {
            des_r = (low + high) / 2.0;
            int m = (int)(des_r * 1638.4 + 0.5);  // des_r = 5 m / 8192 exactly
            unsigned long long cnt = cum[m];
            float pct = ((float)cnt / den) * 100.0f;
            double p = (double)pct;
            if (p < threshold - tol) low = des_r;
            else if (p > threshold + tol) high = des_r;
            else break;
        }
        double r100 = rint(des_r * 100.0);
        des_r_out[threadIdx.x] = r100 / 100.0;
    }
}
}  // namespace

int bxk_radius_hist(bx_ctx* c, hipStream_t s, const float* pts, int n_pts, const float* kpts, int nk)
{
    BX_HIP(hipMemsetAsync(c->rad_hist, 0, sizeof(unsigned long long) * NB, s));
    if (n_pts <= 0 || nk <= 0) return BX_OK;
    // every workgroup ends with a flush of its 8194-bin LDS histogram into the global one (64-bit atomics; rounds 1-5: 2048 workgroups
    // of 256 threads = 16 M atomics = 102 MB of write traffic per launch for a 64 KB result).  Measured in round 3: cutting the flushes
    // eightfold by cutting the point slices (8 instead of 64) DOUBLES the kernel time (317 -> 629 us) -- 4 waves per CU cannot cover the
    // latency of the per-distance bin search; occupancy is what the kernel wants.  Round 6 cuts the flushes fourfold the other way:
    // 16 slices of 1024-thread workgroups (see RH_T): twice the waves per CU, a quarter of the flush traffic.
    int gx = (n_pts + RH_T - 1) / RH_T;
    const int gmax = c->rad_slices > 0 ? c->rad_slices : 16;
    if (gx > gmax) gx = gmax;
    dim3 grid(gx, (nk + 63) / 64);
    hipLaunchKernelGGL(radius_hist_kernel, grid, dim3(RH_T), 0, s, pts, n_pts, kpts, nk, c->d_rad_thr, c->rad_hist);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_radius_bisect_all(bx_ctx* c, hipStream_t s, int64_t n_orig, int nk, const double* thresholds_host, int nthr, double* des_r_out)
{
    if (nthr < 1 || nthr > BX_MAX_SCALES) { bx_set_error("bxk_radius_bisect_all: %d thresholds", nthr); return BX_ERR_ARG; }
    BisectThr t;
    t.n = nthr;
    for (int i = 0; i < BX_MAX_SCALES; ++i) t.t[i] = i < nthr ? thresholds_host[i] : 0.0;
    hipLaunchKernelGGL(radius_bisect_kernel, dim3(1), dim3(1024), 0, s, c->rad_hist, (long long)n_orig, nk, t, des_r_out);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_radius_bisect(bx_ctx* c, hipStream_t s, int64_t n_orig, int nk, double threshold, double* des_r_out)
{
    return bxk_radius_bisect_all(c, s, n_orig, nk, &threshold, 1, des_r_out);
}
