// k_pose.hip -- seeded correspondence RANSAC (binary64) and weighted-Kabsch post refinement (fp32).
//
// RANSAC replaces PoseEstimator._estimate_ransac -> Open3D registration_ransac_based_on_correspondence
// (reference models/pose_estimator.py:84-117; algorithm SURVEY.md §3.4/A.6): a CPU/OpenMP loop behind a
// device->host copy in the reference.  Here hypotheses are evaluated BX_RANSAC_BATCH at a time, one wave per
// hypothesis (3-point Umeyama via a binary64 Jacobi, the edge-length / distance checkers, then the inlier
// count and squared-error sum over all correspondences as a wave reduction); a single-thread scan kernel
// then replays the batch in iteration order, applying Open3D's "is better" rule and its confidence-based
// shrinking of the iteration bound, so the result is exactly the sequential algorithm's.  Later batches
// find est_k already reached and exit immediately.  Samples: counter RNG (seed, 3*itr + j) mod C.
//
// Refinement replaces BufferX.post_refinement + rigid_transform_3d (models/BUFFERX.py:522-603).
#include "bx_common.h"

namespace {

struct RansacCfg {
    double dist_th, similar_th, confidence;
    int max_iter;
    unsigned long long seed;
};

__global__ void ransac_init_kernel(PairState* st, const int32_t* skip_flag)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        bool skip = skip_flag && *skip_flag;
        if (!skip) {
            st->r_best_inl = 0;
            st->r_best_rmse = 0.0;
            st->r_est_k = 0x7fffffff;
            st->r_itr = 0;
            for (int i = 0; i < 16; ++i) st->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void ransac_eval_kernel(const float* __restrict__ ss, const float* __restrict__ tt,
                                                          const int32_t* __restrict__ corr, const int32_t* __restrict__ C_dev,
                                                          int max_C, RansacCfg cfg, int it0, const PairState* __restrict__ st,
                                                          const int32_t* __restrict__ skip_flag, int32_t* __restrict__ r_inl,
                                                          double* __restrict__ r_err, double* __restrict__ r_T)
{
    if (skip_flag && *skip_flag) return;
    int C = *C_dev;
    C = C < max_C ? C : max_C;
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int itr = it0 + slot;
    int est_k = st->r_est_k < cfg.max_iter ? st->r_est_k : cfg.max_iter;
    if (it0 >= est_k) return;                 // whole batch beyond the bound (uniform)
    if (slot >= BX_RANSAC_BATCH) return;
    if (C < 3 || itr >= cfg.max_iter) { if (lane == 0) r_inl[slot] = -1; return; }

    // ---- hypothesis (computed redundantly by all lanes: wave-uniform, no divergence)
    double a[3][3], b[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int sidx = corr[bx_mix64(cfg.seed, (uint64_t)itr * 3 + j) % (uint64_t)C];
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[j][c] = (double)ss[(size_t)sidx * 3 + c]; b[j][c] = (double)tt[(size_t)sidx * 3 + c]; }
    }
    double ma[3], mb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ma[c] = ((a[0][c] + a[1][c]) + a[2][c]) / 3.0;
        mb[c] = ((b[0][c] + b[1][c]) + b[2][c]) / 3.0;
    }
    double H[9], R[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            H[i * 3 + j] = ((a[0][i] - ma[i]) * (b[0][j] - mb[j]) + (a[1][i] - ma[i]) * (b[1][j] - mb[j])) + (a[2][i] - ma[i]) * (b[2][j] - mb[j]);
    bool valid = bxd_kabsch_from_H(H, R) != 0;
    if (valid) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            t[i] = mb[i] - ((R[i * 3] * ma[0] + R[i * 3 + 1] * ma[1]) + R[i * 3 + 2] * ma[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 3; ++j) {
                double ds = sqrt(((a[i][0] - a[j][0]) * (a[i][0] - a[j][0]) + (a[i][1] - a[j][1]) * (a[i][1] - a[j][1])) + (a[i][2] - a[j][2]) * (a[i][2] - a[j][2]));
                double dt = sqrt(((b[i][0] - b[j][0]) * (b[i][0] - b[j][0]) + (b[i][1] - b[j][1]) * (b[i][1] - b[j][1])) + (b[i][2] - b[j][2]) * (b[i][2] - b[j][2]));
                if (ds < dt * cfg.similar_th || dt < ds * cfg.similar_th) valid = false;
            }
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double p = ((R[i * 3] * a[j][0] + R[i * 3 + 1] * a[j][1]) + R[i * 3 + 2] * a[j][2]) + t[i];
                double df = b[j][i] - p;
                d2 = d2 + df * df;
            }
            if (sqrt(d2) > cfg.dist_th) valid = false;
        }
    }
    if (!valid) { if (lane == 0) r_inl[slot] = -1; return; }

    // ---- evaluation over all correspondences (wave-order error sum)
    int good = 0;
    double esum = 0.0;
    for (int k = lane; k < C; k += 64) {
        int ci = corr[k];
        double sx = (double)ss[(size_t)ci * 3], sy = (double)ss[(size_t)ci * 3 + 1], sz = (double)ss[(size_t)ci * 3 + 2];
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double p = ((R[i * 3] * sx + R[i * 3 + 1] * sy) + R[i * 3 + 2] * sz) + t[i];
            double df = p - (double)tt[(size_t)ci * 3 + i];
            d2 = d2 + df * df;
        }
        double dis = sqrt(d2);
        bool in = dis < cfg.dist_th;
        good += in ? 1 : 0;
        esum = esum + (in ? dis * dis : 0.0);
    }
    good = bx_wave_sum_i(good);
    esum = bx_wave_sum(esum);
    if (lane == 0) {
        r_inl[slot] = good;
        r_err[slot] = esum;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            r_T[(size_t)slot * 12 + i * 4 + 0] = R[i * 3 + 0];
            r_T[(size_t)slot * 12 + i * 4 + 1] = R[i * 3 + 1];
            r_T[(size_t)slot * 12 + i * 4 + 2] = R[i * 3 + 2];
            r_T[(size_t)slot * 12 + i * 4 + 3] = t[i];
        }
    }
}

__global__ void ransac_scan_kernel(const int32_t* __restrict__ C_dev, int max_C, RansacCfg cfg, int it0, PairState* st,
                                   const int32_t* __restrict__ skip_flag, const int32_t* __restrict__ r_inl,
                                   const double* __restrict__ r_err, const double* __restrict__ r_T, int last,
                                   double* __restrict__ T_out, int32_t* __restrict__ info_out)
{
    // the batch's inlier counts and squared-error sums are staged in LDS by the whole workgroup (coalesced), then ONE thread
    // replays Open3D's sequential best-so-far / iteration-bound update over them: a single thread reading 4096 slots straight from
    // global memory cost ~0.3 ms per batch, i.e. 4 ms per pair at confidence 1.0 (all 50 000 iterations)
    __shared__ int s_inl[BX_RANSAC_BATCH];
    __shared__ double s_err[BX_RANSAC_BATCH];
    if (skip_flag && *skip_flag) return;
    for (int i = threadIdx.x; i < BX_RANSAC_BATCH; i += blockDim.x) { s_inl[i] = r_inl[i]; s_err[i] = r_err[i]; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int C = *C_dev;
    C = C < max_C ? C : max_C;
    int est_k = st->r_est_k < cfg.max_iter ? st->r_est_k : cfg.max_iter;
    int best_inl = st->r_best_inl;
    double best_rmse = st->r_best_rmse;
    int itr = st->r_itr;
    if (C >= 3 && cfg.dist_th > 0.0 && it0 < est_k && itr == it0) {
        for (int slot = 0; slot < BX_RANSAC_BATCH; ++slot) {
            itr = it0 + slot;
            if (itr >= cfg.max_iter || itr >= est_k) break;
            int inl = s_inl[slot];
            if (inl >= 0) {
                double rmse = inl > 0 ? sqrt(s_err[slot] / (double)inl) : 0.0;
                if (inl > best_inl || (inl == best_inl && rmse < best_rmse)) {
                    best_inl = inl; best_rmse = rmse;
                    for (int i = 0; i < 12; ++i) st->T[i] = r_T[(size_t)slot * 12 + i];
                    if (cfg.confidence < 1.0) {
                        double ratio = (double)inl / (double)C;
                        double r3 = (ratio * ratio) * ratio;
                        double est_d;
                        if (r3 >= 1.0) est_d = 0.0;
                        else est_d = bxd_log(1.0 - cfg.confidence) / bxd_log(1.0 - r3);
                        if (est_d < (double)est_k) est_k = (int)ceil(est_d);
                    }
                }
            }
            itr = it0 + slot + 1;
        }
        st->r_best_inl = best_inl;
        st->r_best_rmse = best_rmse;
        st->r_est_k = est_k;
        st->r_itr = itr;
    }
    if (last) {
        st->num_inliers = best_inl;
        st->ransac_iters = st->r_itr;
        if (T_out)
            for (int i = 0; i < 16; ++i) T_out[i] = st->T[i];
        if (info_out) { info_out[0] = best_inl; info_out[1] = st->r_itr; }
    }
}

// ------------------------------------------------------------------ post refinement, one wave
__global__ __launch_bounds__(64) void refine_kernel(const float* __restrict__ ss, const float* __restrict__ tt,
                                                    const int32_t* __restrict__ M_dev, int max_M, float dist_th,
                                                    float* __restrict__ T_io, float* __restrict__ ws, int32_t* __restrict__ sel,
                                                    int32_t* __restrict__ iters_out, const int32_t* __restrict__ skip_flag)
{
    if (skip_flag && *skip_flag) return;
    const int lane = threadIdx.x;
    int M = *M_dev;
    M = M < max_M ? M : max_M;
    float* L2c = ws;
    float* wv = ws + max_M;
    float T[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = T_io[i];
    int prev = 0, it;
    for (it = 0; it < 20; ++it) {
        int n = 0;
        for (int j0 = 0; j0 < M; j0 += 64) {
            int j = j0 + lane;
            bool in = false;
            float l2 = 0.f;
            if (j < M) {
                float sx = ss[(size_t)j * 3], sy = ss[(size_t)j * 3 + 1], sz = ss[(size_t)j * 3 + 2];
                float d0 = (fmaf(T[2], sz, fmaf(T[1], sy, T[0] * sx)) + T[3]) - tt[(size_t)j * 3];
                float d1 = (fmaf(T[6], sz, fmaf(T[5], sy, T[4] * sx)) + T[7]) - tt[(size_t)j * 3 + 1];
                float d2 = (fmaf(T[10], sz, fmaf(T[9], sy, T[8] * sx)) + T[11]) - tt[(size_t)j * 3 + 2];
                l2 = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
                in = l2 < dist_th;
            }
            unsigned long long bal = __ballot(in);
            int pos = n + __popcll(bal & ((1ULL << lane) - 1ULL));
            if (in) { sel[pos] = j; L2c[pos] = l2; }
            n += __popcll(bal);
        }
        if (n == prev) break;
        prev = n;
        __syncthreads();
        float sw = 0.0f;
        for (int k = lane; k < n; k += 64) {
            float q = L2c[k] / dist_th;
            float w = 1.0f / (1.0f + q * q);
            wv[k] = w;
            sw = sw + w;
        }
        sw = bx_wave_sum(sw);
        __syncthreads();
        float cA[3], cB[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float pa = 0.0f, pb = 0.0f;
            for (int k = lane; k < n; k += 64) {
                float w = wv[k];
                pa = pa + ss[(size_t)sel[k] * 3 + a] * w;
                pb = pb + tt[(size_t)sel[k] * 3 + a] * w;
            }
            cA[a] = bx_wave_sum(pa) / (sw + 1e-6f);
            cB[a] = bx_wave_sum(pb) / (sw + 1e-6f);
        }
        float Hf[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float ph = 0.0f;
                for (int k = lane; k < n; k += 64) {
                    float am = ss[(size_t)sel[k] * 3 + a] - cA[a];
                    float bm = tt[(size_t)sel[k] * 3 + b] - cB[b];
                    ph = ph + (am * wv[k]) * bm;
                }
                Hf[a * 3 + b] = bx_wave_sum(ph);
            }
        double H[9], R[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) H[i] = (double)Hf[i];
        if (!bxd_kabsch_from_H(H, R)) break;
        float Rf[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float rc = fmaf(Rf[a * 3 + 2], cA[2], fmaf(Rf[a * 3 + 1], cA[1], Rf[a * 3 + 0] * cA[0]));
            T[a * 4 + 0] = Rf[a * 3 + 0]; T[a * 4 + 1] = Rf[a * 3 + 1]; T[a * 4 + 2] = Rf[a * 3 + 2];
            T[a * 4 + 3] = cB[a] - rc;
        }
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
        __syncthreads();
    }
    if (lane < 16) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v = (lane == i) ? T[i] : v;
        T_io[lane] = v;
    }
    if (lane == 0 && iters_out) *iters_out = it;
}
}  // namespace

int bxk_ransac(bx_ctx* c, hipStream_t s, const float* ss, const float* tt, const int32_t* corr, const int32_t* C_dev,
               int max_C, uint64_t seed, double* T_out, int32_t* info_out, const int32_t* skip_flag)
{
    RansacCfg cfg{c->p.dist_th, c->p.similar_th, c->p.confidence, c->p.iter_n, seed};
    hipLaunchKernelGGL(ransac_init_kernel, dim3(1), dim3(1), 0, s, c->state, skip_flag);
    int nb = (cfg.max_iter + BX_RANSAC_BATCH - 1) / BX_RANSAC_BATCH;
    if (nb < 1) nb = 1;
    for (int b = 0; b < nb; ++b) {
        int it0 = b * BX_RANSAC_BATCH;
        if (max_C >= 3 && cfg.max_iter > 0)
            hipLaunchKernelGGL(ransac_eval_kernel, dim3(BX_RANSAC_BATCH / 4), dim3(256), 0, s, ss, tt, corr, C_dev, max_C, cfg, it0,
                               c->state, skip_flag, c->ransac_inl, c->ransac_err, c->ransac_T);
        hipLaunchKernelGGL(ransac_scan_kernel, dim3(1), dim3(256), 0, s, C_dev, max_C, cfg, it0, c->state, skip_flag, c->ransac_inl,
                           c->ransac_err, c->ransac_T, b == nb - 1 ? 1 : 0, T_out, info_out);
    }
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_refine(bx_ctx* c, hipStream_t s, const float* ss, const float* tt, const int32_t* M_dev, int max_M, float* T_io,
               int32_t* iters_out)
{
    hipLaunchKernelGGL(refine_kernel, dim3(1), dim3(64), 0, s, ss, tt, M_dev, max_M, (float)c->p.dist_th, T_io, c->refine_ws, c->refine_sel,
                       iters_out, (const int32_t*)nullptr);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
