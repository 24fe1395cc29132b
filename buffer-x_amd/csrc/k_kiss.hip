// k_kiss.hip -- the KISS-Matcher pose back-end on the GPU.
//
// Replaces PoseEstimator._estimate_kiss_matcher (reference models/pose_estimator.py:50-82):
//     matcher = KISSMatcher(KISSMatcherConfig(cfg.match.kiss_resolution)); result = matcher.solve(src[inlier_ind].T, tgt[inlier_ind].T)
//     num_inliers = matcher.get_num_final_inliers()
// The `kiss_matcher` package is a pip dependency that is not under /root/reference; solve() is restated from the published
// algorithm (KISS-Matcher, arXiv 2409.15615: maximum-k-core pruning of the compatibility graph (ROBIN) + the TEASER++ solver
// chain without scale: GNC-TLS rotation on a TIM chain, component-wise TLS translation).  The restatement, its named constants
// and its "parity unpinned" status are in oracle/bx_oracle.c (bxo_kiss_solve); this file computes the same thing bit for bit.
//
// Everything stays on the device (the correspondence count C only exists there): kernels are launched for the worst case
// (max_C) and read the live count.
//   kiss_prep_kernel    binary64 copies of the C correspondences
//   kiss_adj_kernel     compatibility graph as a C x C bitmap: one thread per 32-bit word (32 pair tests, two binary64 square roots each)
//   kiss_core_kernel    maximum k-core: binary search over k, Jacobi-style pruning passes (popcount(row & alive)) -- one workgroup;
//                       the k-core is unique, so any pruning order gives the oracle's set
//   kiss_gnc_kernel     ordered core list, TIM chain, GNC-TLS rotation (one wave: wave-order binary64 sums, Jacobi Kabsch)
//   kiss_tls_*          per axis: stable rank sort of the 2m interval end points, TLS cost of every mid-point (one thread per
//                       candidate, sequential sums: the oracle's order), arg-min, inlier masks -> pose + final inlier count
#include "bx_common.h"

namespace {

constexpr double KISS_ROT_INLIER_W = 0.4;

struct KissWs {            // carved from bx_ctx::kiss_ws (see kiss_carve)
    double *a, *b;         // [maxC][3]
    unsigned* adj;         // [maxC][W]
    unsigned* core;        // [W] bitmap of the maximum k-core
    int* v;                // [maxC] core vertices, ascending
    double *st, *dt;       // [maxC][3] TIM chain
    double *wgt, *res;     // [maxC]
    int* pt;               // [maxC] translation points (correspondence indices)
    double* X;             // [3][maxC] residual components
    double *h, *hs;        // [3][2 maxC]
    double *cost, *xh;     // [3][2 maxC]
    unsigned char* mask;   // [3][maxC]
    int* scal;             // [8]: 0 C, 1 n (core size), 2 m (rotation inliers), 3 GNC iterations, 4 ok
    double* Rt;            // [12] rotation (9) + translation (3)
    int W, maxC;           // words per adjacency row, capacity (stride of the per-axis arrays)
};

__host__ __device__ inline size_t kiss_align(size_t x) { return (x + 255) & ~(size_t)255; }

size_t kiss_carve(char* base, int maxC, KissWs* w)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = kiss_align(off + bytes); return p; };
    const size_t C = (size_t)maxC, W = (C + 31) / 32;
    w->W = (int)W; w->maxC = maxC;
    w->a = (double*)take(C * 24); w->b = (double*)take(C * 24);
    w->adj = (unsigned*)take(C * W * 4);
    w->core = (unsigned*)take(W * 4);
    w->v = (int*)take(C * 4);
    w->st = (double*)take(C * 24); w->dt = (double*)take(C * 24);
    w->wgt = (double*)take(C * 8); w->res = (double*)take(C * 8);
    w->pt = (int*)take(C * 4);
    w->X = (double*)take(3 * C * 8);
    w->h = (double*)take(6 * C * 8); w->hs = (double*)take(6 * C * 8);
    w->cost = (double*)take(6 * C * 8); w->xh = (double*)take(6 * C * 8);
    w->mask = (unsigned char*)take(3 * C);
    w->scal = (int*)take(8 * 4);
    w->Rt = (double*)take(12 * 8);
    return off;
}

__global__ void kiss_prep_kernel(const float* __restrict__ ss, const float* __restrict__ tt, const int32_t* __restrict__ corr,
                                 const int32_t* __restrict__ C_dev, int max_C, KissWs w, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    int C = *C_dev;
    C = C < max_C ? C : max_C;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { w.scal[0] = C; w.scal[1] = 0; w.scal[2] = 0; w.scal[3] = 0; w.scal[4] = 0; }
    if (i >= C) return;
    const int k = corr[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { w.a[(size_t)i * 3 + c] = (double)ss[(size_t)k * 3 + c]; w.b[(size_t)i * 3 + c] = (double)tt[(size_t)k * 3 + c]; }
}

__device__ __forceinline__ double kiss_dist(const double* p, const double* q)
{
    const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    return sqrt((dx * dx + dy * dy) + dz * dz);
}

// word (i, wd) of the adjacency bitmap: bit j set iff  j != i  and  | |a_i - a_j| - |b_i - b_j| | <= thr
__global__ __launch_bounds__(256) void kiss_adj_kernel(const int32_t* __restrict__ C_dev, int max_C, double thr, KissWs w,
                                                       const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    int C = *C_dev;
    C = C < max_C ? C : max_C;
    const int Wc = (C + 31) / 32;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)C * Wc) return;
    const int i = (int)(id / Wc), wd = (int)(id - (long long)i * Wc);
    const double ai[3] = {w.a[(size_t)i * 3], w.a[(size_t)i * 3 + 1], w.a[(size_t)i * 3 + 2]};
    const double bi[3] = {w.b[(size_t)i * 3], w.b[(size_t)i * 3 + 1], w.b[(size_t)i * 3 + 2]};
    unsigned bits = 0;
    for (int q = 0; q < 32; ++q) {
        const int j = wd * 32 + q;
        if (j < C && j != i) {
            const double d = fabs(kiss_dist(ai, w.a + (size_t)j * 3) - kiss_dist(bi, w.b + (size_t)j * 3));
            if (d <= thr) bits |= 1u << q;
        }
    }
    w.adj[(size_t)i * w.W + wd] = bits;
}

// maximum k-core, one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void kiss_core_kernel(KissWs w, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    extern __shared__ unsigned sm[];          // alive[W] | next[W] | best[W]
    __shared__ int s_red, s_left;
    const int C = w.scal[0];
    if (C < 2) { if (threadIdx.x == 0) w.scal[1] = 0; return; }
    const int Wc = (C + 31) / 32;
    unsigned* alive = sm;
    unsigned* next = sm + Wc;
    unsigned* best = sm + 2 * Wc;
    const int t = threadIdx.x;
    // maximum degree
    if (t == 0) s_red = 0;
    __syncthreads();
    int md = 0;
    for (int i = t; i < C; i += 1024) {
        int d = 0;
        for (int x = 0; x < Wc; ++x) d += __popc(w.adj[(size_t)i * w.W + x]);
        md = d > md ? d : md;
    }
    atomicMax(&s_red, md);
    for (int x = t; x < Wc; x += 1024) {
        const int lo_bit = x * 32;
        best[x] = C - lo_bit >= 32 ? 0xffffffffu : ((1u << (C - lo_bit)) - 1u);     // the 0-core: every vertex
    }
    __syncthreads();
    int lo = 0, hi = s_red;
    while (lo < hi) {
        const int k = (lo + hi + 1) / 2;
        for (int x = t; x < Wc; x += 1024) {
            const int lo_bit = x * 32;
            alive[x] = C - lo_bit >= 32 ? 0xffffffffu : ((1u << (C - lo_bit)) - 1u);
        }
        if (t == 0) s_left = C;
        __syncthreads();
        for (;;) {
            for (int x = t; x < Wc; x += 1024) next[x] = alive[x];
            if (t == 0) s_red = 0;
            __syncthreads();
            for (int i = t; i < C; i += 1024) {
                if (!((alive[i >> 5] >> (i & 31)) & 1u)) continue;
                int d = 0;
                const unsigned* row = w.adj + (size_t)i * w.W;
                for (int x = 0; x < Wc; ++x) d += __popc(row[x] & alive[x]);
                if (d < k) { atomicAnd(&next[i >> 5], ~(1u << (i & 31))); atomicAdd(&s_red, 1); }
            }
            __syncthreads();
            const int removed = s_red;
            for (int x = t; x < Wc; x += 1024) alive[x] = next[x];
            if (t == 0) s_left -= removed;
            __syncthreads();
            if (removed == 0 || s_left <= 0) break;
        }
        const int left = s_left;
        if (left > 0) {
            lo = k;
            for (int x = t; x < Wc; x += 1024) best[x] = alive[x];
        } else hi = k - 1;
        __syncthreads();
    }
    for (int x = t; x < Wc; x += 1024) w.core[x] = best[x];
}

// one wave: ordered core list, TIM chain, GNC-TLS rotation, translation residuals
__global__ __launch_bounds__(64) void kiss_gnc_kernel(KissWs w, double solver_nb, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    const int lane = threadIdx.x;
    const int C = w.scal[0];
    if (C < 2) return;
    const int Wc = (C + 31) / 32;
    // ---- ascending list of the core vertices (ballot compaction, 64 vertices per step)
    int n = 0;
    for (int i0 = 0; i0 < C; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < C && ((w.core[i >> 5] >> (i & 31)) & 1u);
        const unsigned long long bal = __ballot(in);
        if (in) w.v[n + __popcll(bal & ((1ULL << lane) - 1ULL))] = i;
        n += __popcll(bal);
    }
    (void)Wc;
    if (lane == 0) w.scal[1] = n;
    if (n < 2) return;
    __syncthreads();
    // ---- TIM chain with wrap-around
    for (int i = lane; i < n; i += 64) {
        const int p = w.v[i], q = w.v[(i + 1) % n];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            w.st[(size_t)i * 3 + c] = w.a[(size_t)q * 3 + c] - w.a[(size_t)p * 3 + c];
            w.dt[(size_t)i * 3 + c] = w.b[(size_t)q * 3 + c] - w.b[(size_t)p * 3 + c];
        }
        w.wgt[i] = 1.0;
    }
    __syncthreads();
    // ---- GNC-TLS
    const double nb = 2.0 * solver_nb, nb2 = nb * nb;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double mu = 1.0, prev_cost = 0.0;
    int it, ok = 1, have_prev = 0;
    for (it = 0; it < 100; ++it) {
        double H[9], Rn[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double acc = 0.0;
                for (int i = lane; i < n; i += 64) acc = acc + (w.wgt[i] * w.st[(size_t)i * 3 + r]) * w.dt[(size_t)i * 3 + c];
                H[r * 3 + c] = bx_wave_sum(acc);
            }
        if (!bxd_kabsch_from_H(H, Rn)) { ok = it > 0; break; }
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = Rn[q];
        double maxres = 0.0;
        for (int i = lane; i < n; i += 64) {
            double r2 = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double p = (R[c * 3] * w.st[(size_t)i * 3] + R[c * 3 + 1] * w.st[(size_t)i * 3 + 1]) + R[c * 3 + 2] * w.st[(size_t)i * 3 + 2];
                const double d = w.dt[(size_t)i * 3 + c] - p;
                r2 = r2 + d * d;
            }
            w.res[i] = r2;
            maxres = r2 > maxres ? r2 : maxres;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { const double o = __shfl_xor(maxres, s, 64); maxres = o > maxres ? o : maxres; }
        if (it == 0) {
            mu = 1.0 / (2.0 * maxres / nb2 - 1.0);
            if (!(mu > 0.0)) { ++it; break; }      // every residual already inside the bound: the unweighted fit is final
        }
        const double th1 = (mu + 1.0) / mu * nb2, th2 = mu / (mu + 1.0) * nb2;
        double cacc = 0.0;
        for (int i = lane; i < n; i += 64) cacc = cacc + w.wgt[i] * w.res[i];
        const double cost = bx_wave_sum(cacc);
        for (int i = lane; i < n; i += 64) {
            const double r2 = w.res[i];
            double ww;
            if (r2 >= th1) ww = 0.0;
            else if (r2 <= th2) ww = 1.0;
            else ww = sqrt(nb2 * mu * (mu + 1.0) / r2) - mu;
            w.wgt[i] = ww;
        }
        __syncthreads();
        const double diff = have_prev ? fabs(cost - prev_cost) : 1.0e300;
        mu = mu * 1.4;
        prev_cost = cost; have_prev = 1;
        if (diff < 1e-6) { ++it; break; }
    }
    if (lane == 0) { w.scal[3] = it; w.scal[4] = ok; }
    if (!ok) return;
    // ---- translation points = first end point of every inlier TIM, in chain order; residual components
    int m = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n && w.wgt[i] >= KISS_ROT_INLIER_W;
        const unsigned long long bal = __ballot(in);
        if (in) w.pt[m + __popcll(bal & ((1ULL << lane) - 1ULL))] = w.v[i];
        m += __popcll(bal);
    }
    __syncthreads();
    const size_t maxC = (size_t)w.maxC;       // stride of the per-axis arrays
    for (int i = lane; i < m; i += 64) {
        const double* s = w.a + (size_t)w.pt[i] * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            w.X[c * maxC + i] = w.b[(size_t)w.pt[i] * 3 + c] - ((R[c * 3] * s[0] + R[c * 3 + 1] * s[1]) + R[c * 3 + 2] * s[2]);
    }
    if (lane == 0) w.scal[2] = m;
    if (lane < 9) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) v = lane == q ? R[q] : v;
        w.Rt[lane] = v;
    }
}

// stable rank sort of the 2m interval end points of axis blockIdx.y
__global__ __launch_bounds__(256) void kiss_tls_rank_kernel(KissWs w, double beta, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    const int m = w.scal[2];
    if (!w.scal[4] || m <= 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * m) return;
    const size_t maxC = (size_t)w.maxC;
    const int ax = blockIdx.y;
    const double* X = w.X + ax * maxC;
    auto hval = [&](int q) { return q < m ? X[q] - beta : X[q - m] + beta; };
    const double hi_ = hval(i);
    int r = 0;
    for (int j = 0; j < 2 * m; ++j) {
        const double hj = hval(j);
        r += (hj < hi_) || (hj == hi_ && j < i);
    }
    w.hs[ax * 2 * maxC + r] = hi_;
}

// TLS cost of mid-point k of axis blockIdx.y
__global__ __launch_bounds__(256) void kiss_tls_eval_kernel(KissWs w, double beta, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    const int m = w.scal[2];
    if (!w.scal[4] || m <= 0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k + 1 >= 2 * m) return;
    const size_t maxC = (size_t)w.maxC;
    const int ax = blockIdx.y;
    const double* X = w.X + ax * maxC;
    const double* hs = w.hs + ax * 2 * maxC;
    const double c = (hs[k] + hs[k + 1]) / 2.0;
    const double wt = 1.0 / (beta * beta);
    double sx = 0.0, sw = 0.0;
    int cnt = 0;
    for (int j = 0; j < m; ++j)
        if (fabs(X[j] - c) <= beta) { sx = sx + X[j] * wt; sw = sw + wt; ++cnt; }
    double cost = 1.0e300, xh = 0.0;      // empty consensus set: never the minimum
    if (cnt > 0) {
        xh = sx / sw;
        double res = 0.0;
        for (int j = 0; j < m; ++j)
            if (fabs(X[j] - c) <= beta) { const double d = X[j] - xh; res = res + (d * d) * wt; }
        cost = res + beta * (double)(m - cnt);
    }
    w.cost[ax * 2 * maxC + k] = cnt > 0 ? cost : -1.0;      // -1 marks "no consensus" (a cost is never negative)
    w.xh[ax * 2 * maxC + k] = xh;
}

// arg-min per axis (first minimum), inlier masks, pose and counts -> PairState (one workgroup)
__global__ __launch_bounds__(256) void kiss_final_kernel(KissWs w, double beta, PairState* st, double* T_out, int32_t* info_out,
                                                         const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    __shared__ double s_c[256];
    __shared__ int s_k[256];
    __shared__ double s_ctr[3], s_t[3];
    __shared__ int s_cnt;
    const int t = threadIdx.x;
    const int C = w.scal[0], n = w.scal[1], m = w.scal[2], ok = w.scal[4];
    const size_t maxC = (size_t)w.maxC;
    if (t == 0) s_cnt = 0;
    const bool solved = C >= 2 && n >= 2 && ok;
    if (solved && m > 0) {
        for (int ax = 0; ax < 3; ++ax) {
            const double* cost = w.cost + ax * 2 * maxC;
            double bc = 1.0e301; int bk = 0x7fffffff;
            for (int k = t; k + 1 < 2 * m; k += 256) {
                const double c = cost[k];
                if (c >= 0.0 && (c < bc)) { bc = c; bk = k; }           // strided scan keeps the lowest k of a tie per thread
            }
            s_c[t] = bc; s_k[t] = bk;
            __syncthreads();
            if (t == 0) {
                double gc = 1.0e301; int gk = 0x7fffffff;
                for (int q = 0; q < 256; ++q)
                    if (s_k[q] != 0x7fffffff && (s_c[q] < gc || (s_c[q] == gc && s_k[q] < gk))) { gc = s_c[q]; gk = s_k[q]; }
                if (gk != 0x7fffffff) {
                    const double* hs = w.hs + ax * 2 * maxC;
                    s_ctr[ax] = (hs[gk] + hs[gk + 1]) / 2.0;
                    s_t[ax] = w.xh[ax * 2 * maxC + gk];
                } else { s_ctr[ax] = 1.0e300; s_t[ax] = 0.0; }
            }
            __syncthreads();
        }
        int mine = 0;
        for (int j = t; j < m; j += 256) {
            bool in = true;
            for (int ax = 0; ax < 3; ++ax) in = in && (s_ctr[ax] < 1.0e299) && fabs(w.X[ax * maxC + j] - s_ctr[ax]) <= beta;
            mine += in ? 1 : 0;
        }
        atomicAdd(&s_cnt, mine);
    } else if (t < 3) s_t[t] = 0.0;
    __syncthreads();
    if (t == 0) {
        double T[16];
        for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
        if (solved) {
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) T[r * 4 + c] = w.Rt[r * 3 + c];
                T[r * 4 + 3] = m > 0 ? s_t[r] : 0.0;
            }
        }
        const int fin = solved ? s_cnt : 0;
        if (st) {
            for (int i = 0; i < 16; ++i) st->T[i] = T[i];
            st->num_inliers = fin;
            st->ransac_iters = w.scal[3];
        }
        if (T_out) for (int i = 0; i < 16; ++i) T_out[i] = T[i];
        if (info_out) { info_out[0] = fin; info_out[1] = n; info_out[2] = solved ? m : 0; info_out[3] = w.scal[3]; }
    }
}
}  // namespace

size_t bxk_kiss_workspace_bytes(int max_C)
{
    KissWs w;
    return kiss_carve(nullptr, max_C, &w);
}

// KISS-Matcher solve() on the correspondences corr[0..*C_dev).  Writes the pose / inlier count / GNC iteration count into the
// context's PairState like bxk_ransac, and optionally T_out (device double[16]) / info_out (device int32[4] = {final inliers,
// core size, rotation inliers, GNC iterations}).
int bxk_kiss(bx_ctx* c, hipStream_t s, const float* ss, const float* tt, const int32_t* corr, const int32_t* C_dev, int max_C,
             double* T_out, int32_t* info_out, const int32_t* skip_flag)
{
    if (!c->kiss_ws) { bx_set_error("bxk_kiss: the context was created without the KISS-Matcher workspace (pose_estimator)"); return BX_ERR_STATE; }
    if (max_C > c->kiss_max_C) { bx_set_error("bxk_kiss: max_C=%d exceeds the workspace (%d)", max_C, c->kiss_max_C); return BX_ERR_ARG; }
    if (max_C < 1) return BX_OK;
    KissWs w;
    kiss_carve(c->kiss_ws, c->kiss_max_C, &w);
    const double robin_nb = 1.0 * c->p.kiss_resolution, solver_nb = 0.75 * c->p.kiss_resolution;   // KISSMatcherConfig gains
    const int W = (max_C + 31) / 32;
    hipLaunchKernelGGL(kiss_prep_kernel, dim3((max_C + 255) / 256), dim3(256), 0, s, ss, tt, corr, C_dev, max_C, w, skip_flag);
    const long long words = (long long)max_C * W;
    hipLaunchKernelGGL(kiss_adj_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, C_dev, max_C, 2.0 * robin_nb, w, skip_flag);
    hipLaunchKernelGGL(kiss_core_kernel, dim3(1), dim3(1024), (size_t)3 * W * 4, s, w, skip_flag);
    hipLaunchKernelGGL(kiss_gnc_kernel, dim3(1), dim3(64), 0, s, w, solver_nb, skip_flag);
    hipLaunchKernelGGL(kiss_tls_rank_kernel, dim3((2 * max_C + 255) / 256, 3), dim3(256), 0, s, w, solver_nb, skip_flag);
    hipLaunchKernelGGL(kiss_tls_eval_kernel, dim3((2 * max_C + 255) / 256, 3), dim3(256), 0, s, w, solver_nb, skip_flag);
    hipLaunchKernelGGL(kiss_final_kernel, dim3(1), dim3(256), 0, s, w, solver_nb, c->state, T_out, info_out, skip_flag);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
