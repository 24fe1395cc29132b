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
    // CorrespondenceCheckerBasedOnEdgeLength FIRST: it needs no transformation, rejects most random triples, and a rejected
    // hypothesis then skips the binary64 Jacobi of the Umeyama fit (the accepted set is the same in either order: a hypothesis
    // must pass every check; at confidence 1.0 all 50 000 iterations run and the fit was 7.4 ms per pair)
    {
        bool edge_ok = true;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 3; ++j) {
                double ds = sqrt(((a[i][0] - a[j][0]) * (a[i][0] - a[j][0]) + (a[i][1] - a[j][1]) * (a[i][1] - a[j][1])) + (a[i][2] - a[j][2]) * (a[i][2] - a[j][2]));
                double dt = sqrt(((b[i][0] - b[j][0]) * (b[i][0] - b[j][0]) + (b[i][1] - b[j][1]) * (b[i][1] - b[j][1])) + (b[i][2] - b[j][2]) * (b[i][2] - b[j][2]));
                if (ds < dt * cfg.similar_th || dt < ds * cfg.similar_th) edge_ok = false;
            }
        if (!edge_ok) { if (lane == 0) r_inl[slot] = -1; return; }
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

// Open3D's sequential best-so-far / iteration-bound update over one batch, as two block-wide scans instead of one thread walking
// 4096 slots (that walk was 0.5 ms per batch: 6.6 of the 7.4 ms RANSAC cost at confidence 1.0, where all 13 batches run).
// The sequential rule, per slot in iteration order:   stop when itr >= max_iter or itr >= est_k;   a valid hypothesis becomes the
// best if it has more inliers, or as many and a smaller rmse (ties keep the EARLIER one);   a new best sets
// est_k = min(est_k, ceil(est_d(inliers)))  when confidence < 1  (`if (est_d < est_k) est_k = ceil(est_d)` is that minimum).
//   scan 1: running best = inclusive scan with  combine(L, R) = better(R, L) ? R : L  (associative, left-biased), seeded with the
//           best carried in from the previous batches;  slot i is a RECORD iff it beats the running best in front of it;
//   scan 2: est_k in front of slot i = min(carried est_k, est of every record before i)  (a record's est only depends on its inliers);
//   stop   = the first slot whose iteration number reaches max_iter or the est_k in front of it;  the result is the running state
//           in front of slot `stop`.  Every value is computed by the same expressions as in the sequential loop.
struct RBest { int inl; double rmse; int slot; };      // slot -1: the best carried in; inl -1: nothing valid
__device__ __forceinline__ bool rbetter(const RBest& a, const RBest& b)   // a (valid) replaces b in the sequential loop
{
    return a.inl > b.inl || (a.inl == b.inl && a.rmse < b.rmse);
}
__device__ __forceinline__ int ransac_est(int inl, int C, double confidence)
{
    const double ratio = (double)inl / (double)C;
    const double r3 = (ratio * ratio) * ratio;
    double est_d;
    if (r3 >= 1.0) est_d = 0.0;
    else est_d = bxd_log(1.0 - confidence) / bxd_log(1.0 - r3);
    return est_d < 2147483000.0 ? (int)ceil(est_d) : 0x7fffffff;      // est_k never exceeds max_iter: a huge est_d changes nothing
}

__global__ __launch_bounds__(1024) void ransac_scan_kernel(const int32_t* __restrict__ C_dev, int max_C, RansacCfg cfg, int it0, PairState* st,
                                   const int32_t* __restrict__ skip_flag, const int32_t* __restrict__ r_inl,
                                   const double* __restrict__ r_err, const double* __restrict__ r_T, int last,
                                   double* __restrict__ T_out, int32_t* __restrict__ info_out)
{
    constexpr int NT = 1024, PER = BX_RANSAC_BATCH / NT;       // 4 consecutive slots per thread
    static_assert(BX_RANSAC_BATCH % NT == 0, "slots per thread");
    __shared__ int s_inl[NT];
    __shared__ double s_rmse[NT];
    __shared__ int s_slot[NT];
    __shared__ int s_est[NT];
    __shared__ int s_stop;
    if (skip_flag && *skip_flag) return;
    const int t = threadIdx.x;
    int C = *C_dev;
    C = C < max_C ? C : max_C;
    const int est_in = st->r_est_k < cfg.max_iter ? st->r_est_k : cfg.max_iter;
    const RBest carry{st->r_best_inl, st->r_best_rmse, -1};
    const int itr_in = st->r_itr;
    const bool run = C >= 3 && cfg.dist_th > 0.0 && it0 < est_in && itr_in == it0;     // uniform
    const bool shrink = cfg.confidence < 1.0;
    if (!run) {
        if (last && t == 0) {
            st->num_inliers = carry.inl;
            st->ransac_iters = itr_in;
            if (T_out)
                for (int i = 0; i < 16; ++i) T_out[i] = st->T[i];
            if (info_out) { info_out[0] = carry.inl; info_out[1] = itr_in; }
        }
        return;
    }
    // ---- this thread's slots; leftmost best among the valid ones
    RBest v[PER];
    RBest loc{-1, 0.0, -2};
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int slot = t * PER + u;
        const int inl = r_inl[slot];
        v[u].inl = inl; v[u].slot = slot;
        v[u].rmse = inl > 0 ? sqrt(r_err[slot] / (double)inl) : 0.0;
        if (inl >= 0 && (loc.inl < 0 || rbetter(v[u], loc))) loc = v[u];
    }
    s_inl[t] = loc.inl; s_rmse[t] = loc.rmse; s_slot[t] = loc.slot;
    __syncthreads();
    // ---- scan 1 (inclusive, Hillis-Steele): an invalid element (inl -1) loses to everything and never replaces anything
    for (int d = 1; d < NT; d <<= 1) {
        RBest L{-1, 0.0, -2};
        const bool take = t >= d;
        if (take) L = RBest{s_inl[t - d], s_rmse[t - d], s_slot[t - d]};
        __syncthreads();
        if (take && L.inl >= 0) {
            const RBest R{s_inl[t], s_rmse[t], s_slot[t]};
            if (!(R.inl >= 0 && rbetter(R, L))) { s_inl[t] = L.inl; s_rmse[t] = L.rmse; s_slot[t] = L.slot; }     // left-biased
        }
        __syncthreads();
    }
    // running best in front of this thread's first slot
    RBest before = carry;
    if (t > 0) {
        const RBest Pm{s_inl[t - 1], s_rmse[t - 1], s_slot[t - 1]};
        if (Pm.inl >= 0 && rbetter(Pm, before)) before = Pm;
    }
    // ---- records among this thread's slots and their est_k
    bool rec[PER];
    int est[PER];
    int est_loc = 0x7fffffff;
    {
        RBest cur = before;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            rec[u] = v[u].inl >= 0 && rbetter(v[u], cur);
            est[u] = 0x7fffffff;
            if (rec[u]) {
                cur = v[u];
                if (shrink) est[u] = ransac_est(v[u].inl, C, cfg.confidence);
                est_loc = est[u] < est_loc ? est[u] : est_loc;
            }
        }
    }
    // ---- scan 2 (inclusive min)
    s_est[t] = est_loc;
    if (t == 0) s_stop = BX_RANSAC_BATCH;
    __syncthreads();
    for (int d = 1; d < NT; d <<= 1) {
        int L = 0x7fffffff;
        const bool take = t >= d;
        if (take) L = s_est[t - d];
        __syncthreads();
        if (take && L < s_est[t]) s_est[t] = L;
        __syncthreads();
    }
    int est_before = est_in;
    if (t > 0 && s_est[t - 1] < est_before) est_before = s_est[t - 1];
    // ---- stop: the first slot whose iteration number reaches max_iter or the est_k in front of it
    {
        int e = est_before, my_stop = BX_RANSAC_BATCH;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int slot = t * PER + u, itr = it0 + slot;
            if (my_stop == BX_RANSAC_BATCH && (itr >= cfg.max_iter || itr >= e)) my_stop = slot;
            if (rec[u] && est[u] < e) e = est[u];
        }
        if (my_stop < BX_RANSAC_BATCH) atomicMin(&s_stop, my_stop);
    }
    __syncthreads();
    const int stop = s_stop;
    // ---- state in front of slot `stop`: written by the thread that owns slot stop - 1 (thread 0 when stop == 0)
    const int owner = stop > 0 ? (stop - 1) / PER : 0;
    if (t == owner) {
        RBest cur = before;
        int e = est_before;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int slot = t * PER + u;
            if (slot < stop && rec[u]) {
                cur = v[u];
                if (est[u] < e) e = est[u];
            }
        }
        if (cur.slot >= 0)
            for (int i = 0; i < 12; ++i) st->T[i] = r_T[(size_t)cur.slot * 12 + i];
        st->r_best_inl = cur.inl;
        st->r_best_rmse = cur.rmse;
        st->r_est_k = e;
        st->r_itr = it0 + stop;
        if (last) {
            st->num_inliers = cur.inl;
            st->ransac_iters = it0 + stop;
            if (T_out) {
                for (int i = 0; i < 12; ++i) T_out[i] = cur.slot >= 0 ? r_T[(size_t)cur.slot * 12 + i] : st->T[i];
                T_out[12] = 0.0; T_out[13] = 0.0; T_out[14] = 0.0; T_out[15] = 1.0;
            }
            if (info_out) { info_out[0] = cur.inl; info_out[1] = it0 + stop; }
        }
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
        hipLaunchKernelGGL(ransac_scan_kernel, dim3(1), dim3(1024), 0, s, C_dev, max_C, cfg, it0, c->state, skip_flag, c->ransac_inl,
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
