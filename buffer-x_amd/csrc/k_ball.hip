// k_ball.hip -- neighbour gather: MiniSpinNet.select_patches (reference models/patch_embedder.py:92-120) =
// pointnet2_ops.ball_query + grouping_operation + the pad/mask arithmetic, fused; plus the cloud permutation.
//
// Semantics that must survive any acceleration: per keypoint the FIRST P points, in the
// order of the (permuted) cloud, with fp32 d2 < r2 (strict, un-fused ((dx*dx+dy*dy)+dz*dz)); unfilled slots
// repeat the first hit; slots equal to the first hit (except slot 0) and slot P-1 become the keypoint.
//
// Round-1 kernel (exact brute force, streaming):
//   * one wave per keypoint, WPB keypoints per workgroup; the cloud streams through LDS in 2048-point SoA
//     tiles shared by the workgroup's waves (coalesced HBM/L2 -> LDS once per workgroup);
//   * each wave tests 64 points per step; the 64-bit ballot of the step is the hit BITMAP word for those 64
//     points -- kept in a VGPR (lane w&63 owns word w) and flushed to LDS once per 64 steps, so the scan
//     issues no per-step stores; the scan stops once P hits were seen (first-P semantics);
//   * expansion: popcount prefix over the bitmap words, then output slot j finds its word by binary search
//     and its bit by a 6-step select -- the P outputs are produced in order, idx and xyz written with
//     coalesced stores.  Algorithmic HBM bytes: 12N + 12K + 4KP + 12KP.
#include "bx_common.h"

namespace {
constexpr int TILE = 2048;

__global__ void permute_kernel(const float* __restrict__ pts, const int32_t* __restrict__ perm, int n, float* __restrict__ out,
                               const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        size_t s = (size_t)perm[i] * 3;
        out[(size_t)i * 3 + 0] = pts[s + 0];
        out[(size_t)i * 3 + 1] = pts[s + 1];
        out[(size_t)i * 3 + 2] = pts[s + 2];
    }
}

__device__ __forceinline__ int select_bit(unsigned long long x, int r)
{
    int pos = 0;
    int c = __popc((unsigned)(x & 0xffffffffu));
    if (r >= c) { r -= c; pos += 32; x >>= 32; }
    c = __popc((unsigned)(x & 0xffffu));
    if (r >= c) { r -= c; pos += 16; x >>= 16; }
    c = __popc((unsigned)(x & 0xffu));
    if (r >= c) { r -= c; pos += 8; x >>= 8; }
    c = __popc((unsigned)(x & 0xfu));
    if (r >= c) { r -= c; pos += 4; x >>= 4; }
    c = __popc((unsigned)(x & 0x3u));
    if (r >= c) { r -= c; pos += 2; x >>= 2; }
    c = (int)(x & 1u);
    if (r >= c) { pos += 1; }
    return pos;
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void ball_group_kernel(const float* __restrict__ pts, int n,
                                                              const float* __restrict__ kpts, int K,
                                                              const double* __restrict__ radius, int P,
                                                              int32_t* __restrict__ idx_out, float* __restrict__ patches,
                                                              const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W64 = (n + 63) >> 6;
    float* sx = reinterpret_cast<float*>(smem);
    float* sy = sx + TILE;
    float* sz = sy + TILE;
    unsigned long long* bm_all = reinterpret_cast<unsigned long long*>(sz + TILE);
    int* pf_all = reinterpret_cast<int*>(bm_all + (size_t)WPB * W64);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x * WPB + wave;
    unsigned long long* bm = bm_all + (size_t)wave * W64;
    int* pf = pf_all + (size_t)wave * W64;

    const float r = (float)(*radius);
    const float r2 = r * r;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    bool done = q >= K;
    if (!done) { qx = kpts[(size_t)q * 3]; qy = kpts[(size_t)q * 3 + 1]; qz = kpts[(size_t)q * 3 + 2]; }

    int total = 0;      // hits seen so far (wave-uniform)
    int wused = 0;      // bitmap words written
    unsigned long long myword = 0;

    for (int tile0 = 0; tile0 < n; tile0 += TILE) {
        for (int i = tid; i < TILE; i += WPB * 64) {
            int j = tile0 + i;
            if (j < n) {
                sx[i] = pts[(size_t)j * 3 + 0];
                sy[i] = pts[(size_t)j * 3 + 1];
                sz[i] = pts[(size_t)j * 3 + 2];
            }
        }
        __syncthreads();
        if (!done) {
            const int cmax = min(TILE, n - tile0);
            for (int c0 = 0; c0 < cmax; c0 += 64) {
                int i = c0 + lane;
                bool hit = false;
                if (i < cmax) {
                    float dx = qx - sx[i], dy = qy - sy[i], dz = qz - sz[i];
                    float d2 = (dx * dx + dy * dy) + dz * dz;
                    hit = d2 < r2;
                }
                unsigned long long mask = __ballot(hit);
                int w = (tile0 + c0) >> 6;
                if (lane == (w & 63)) myword = mask;
                total += __popcll(mask);
                wused = w + 1;
                bool last = (total >= P) || (tile0 + c0 + 64 >= n);
                if ((w & 63) == 63 || last) {
                    int wb = w & ~63;
                    if (wb + lane < W64) bm[wb + lane] = myword;
                    myword = 0;
                }
                if (total >= P) { done = true; break; }
            }
        }
        if (__syncthreads_and(done ? 1 : 0)) break;
    }
    if (q >= K) return;

    // ---- popcount prefix over the visited words
    int run = 0;
    for (int g0 = 0; g0 < wused; g0 += 64) {
        int w = g0 + lane;
        unsigned long long word = w < wused ? bm[w] : 0ULL;
        int pc = __popcll(word);
        int ex = bx_wave_excl_scan(pc, lane);
        if (w < wused) pf[w] = run + ex;
        run += bx_wave_sum_i(pc);
    }
    const int nhit = run < P ? run : P;

    // ---- ordered expansion
    int first = 0;
    for (int j0 = 0; j0 < P; j0 += 64) {
        int j = j0 + lane;
        int idx = 0;
        if (j < nhit) {
            int lo = 0, hi = wused;  // first word with pf > j
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (pf[mid] > j) hi = mid; else lo = mid + 1;
            }
            int w = lo - 1;
            idx = w * 64 + select_bit(bm[w], j - pf[w]);
        }
        if (j0 == 0) first = __shfl(idx, 0, 64);
        if (j >= nhit) idx = first;
        if (j < P) {
            float mask = (idx == first) ? 1.0f : 0.0f;
            if (j == 0) mask = 0.0f;
            if (j == P - 1) mask = 1.0f;
            float om = 1.0f - mask;
            float p0 = pts[(size_t)idx * 3 + 0], p1 = pts[(size_t)idx * 3 + 1], p2 = pts[(size_t)idx * 3 + 2];
            size_t o = ((size_t)q * P + j);
            if (idx_out) idx_out[o] = idx;
            patches[o * 3 + 0] = p0 * om + qx * mask;
            patches[o * 3 + 1] = p1 * om + qy * mask;
            patches[o * 3 + 2] = p2 * om + qz * mask;
        }
    }
}
}  // namespace

int bx_permute_launch(hipStream_t s, const float* pts, const int32_t* perm, int n, float* out, const int32_t* skip)
{
    if (n <= 0) return BX_OK;
    hipLaunchKernelGGL(permute_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, perm, n, out, skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_ball_group(bx_ctx* c, hipStream_t s, const float* pts_perm, int n, const float* kpts, int K, const double* radius,
                   int P, int32_t* idx_out, float* patches_out)
{
    if (K <= 0) return BX_OK;
    if (n <= 0 || P < 2) { bx_set_error("bxk_ball_group: n=%d P=%d", n, P); return BX_ERR_ARG; }
    const size_t W64 = ((size_t)n + 63) >> 6;
    auto lds = [&](int wpb) { return (size_t)3 * TILE * 4 + (size_t)wpb * W64 * 12; };
    const size_t cap = 150 * 1024;
    if (lds(8) <= cap) {
        hipLaunchKernelGGL(ball_group_kernel<8>, dim3((K + 7) / 8), dim3(512), lds(8), s, pts_perm, n, kpts, K, radius, P, idx_out, patches_out, c->skip);
    } else if (lds(4) <= cap) {
        hipLaunchKernelGGL(ball_group_kernel<4>, dim3((K + 3) / 4), dim3(256), lds(4), s, pts_perm, n, kpts, K, radius, P, idx_out, patches_out, c->skip);
    } else if (lds(1) <= cap) {
        hipLaunchKernelGGL(ball_group_kernel<1>, dim3(K), dim3(64), lds(1), s, pts_perm, n, kpts, K, radius, P, idx_out, patches_out, c->skip);
    } else {
        bx_set_error("bxk_ball_group: cloud of %d points too large for the LDS bitmap", n);
        return BX_ERR_ARG;
    }
    BX_LAUNCH_CHECK();
    return BX_OK;
}
