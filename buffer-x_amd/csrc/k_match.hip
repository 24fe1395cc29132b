// k_match.hip -- descriptor head, mutual nearest-neighbour matching, soft-argmax, pose hypotheses and
// cross-scale consensus.  All counts (m, M, C) stay on the device: kernels are launched for the worst case
// and read the live count, so the whole pair runs without a host round trip.
//   desc head   : pool_layer + weighted avg-pool + L2 norms   (reference models/patch_embedder.py:32-39, 80-83)
//   mutual NN   : BufferX.mutual_matching, knn_cuda.KNN(k=1)   (models/BUFFERX.py:469-496)
//   soft-argmax : CostVolume.forward tail                      (models/BUFFERX.py:66-69)
//   hypotheses  : models/BUFFERX.py:382-389 (+ kornia axis_angle_to_rotation_matrix)
//   consensus   : models/BUFFERX.py:405-417
#include "bx_common.h"

namespace {

// ------------------------------------------------------------------ descriptor head: one wave per patch
__global__ __launch_bounds__(64) void desc_head_kernel(const float* __restrict__ x, int K, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ desc,
                                                       float* __restrict__ equi, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    const int q = blockIdx.x, lane = threadIdx.x;
    const float* xq = x + (size_t)q * 2 * BX_EA * 16;
    float part[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) part[c] = 0.0f;
    for (int p = lane; p < BX_EA; p += 64) {
        float v[32];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float4* r4 = reinterpret_cast<const float4*>(xq + ((size_t)ch * BX_EA + p) * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 f = r4[u];  // slots 4u..4u+3 = channels u, 4+u, 8+u, 12+u
                v[ch * 16 + u] = f.x; v[ch * 16 + 4 + u] = f.y; v[ch * 16 + 8 + u] = f.z; v[ch * 16 + 12 + u] = f.w;
            }
        }
        float acc2 = b2[0];
#pragma unroll 1
        for (int c = 0; c < 16; ++c) {
            float acc = b1[c];
#pragma unroll
            for (int ci = 0; ci < 32; ++ci) acc = fmaf(w1[c * 32 + ci], v[ci], acc);
            float h = acc > 0.0f ? acc : 0.0f;
            acc2 = fmaf(w2[c], h, acc2);
        }
        float wgt = acc2 > 0.0f ? acc2 : 0.0f;
        float s2 = 0.0f;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            part[c] = part[c] + v[c] * wgt;
            s2 = fmaf(v[c], v[c], s2);
        }
        float n2 = sqrtf(s2);
        n2 = n2 > 1e-12f ? n2 : 1e-12f;
        float4* eo = reinterpret_cast<float4*>(equi + ((size_t)q * BX_EA + p) * 32);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            eo[u] = make_float4(v[4 * u] / n2, v[4 * u + 1] / n2, v[4 * u + 2] / n2, v[4 * u + 3] / n2);
    }
    float f[32];
    float ss = 0.0f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        f[c] = bx_wave_sum(part[c]) / (float)BX_EA;
        ss = fmaf(f[c], f[c], ss);
    }
    float nn = sqrtf(ss);
    nn = nn > 1e-12f ? nn : 1e-12f;
    float mine = 0.0f;
#pragma unroll
    for (int c = 0; c < 32; ++c) mine = (lane == c) ? f[c] / nn : mine;
    if (lane < 32) desc[(size_t)q * 32 + lane] = mine;
}

// ------------------------------------------------------------------ brute-force 1-NN on 32-D descriptors
constexpr int NN_TILE = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
// One thread per query, the references of a tile in LDS, TWO references per step as packed fp32 (v_pk_add_f32 / v_pk_fma_f32: the
// squared distance of a (query, reference) pair is still the fmaf chain over d = 0..31 from 0 -- the two chains of a step are the two
// halves of one packed register, the query component is broadcast to both).  LDS layout [reference pair][d][2]: a ds_read_b128
// delivers two components of both references.  Half the VALU instructions of the scalar form (round 3: 109 -> see LABBOOK.md).
__global__ __launch_bounds__(256) void nn1_kernel(const float* __restrict__ qd, int nq, const float* __restrict__ rd, int nr,
                                                  int seg_len, unsigned long long* __restrict__ keys, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    __shared__ __attribute__((aligned(16))) float sr[NN_TILE / 2][32][2];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float q[32];
    if (i < nq) {
#pragma unroll
        for (int d = 0; d < 32; ++d) q[d] = qd[(size_t)i * 32 + d];
    } else {
#pragma unroll
        for (int d = 0; d < 32; ++d) q[d] = 0.f;
    }
    const int j0 = blockIdx.y * seg_len;
    const int j1 = min(nr, j0 + seg_len);
    unsigned long long best = ~0ULL;
    for (int t0 = j0; t0 < j1; t0 += NN_TILE) {
        __syncthreads();
        for (int f = threadIdx.x; f < NN_TILE * 32; f += 256) {
            const int jl = f >> 5, jj = t0 + jl;
            sr[jl >> 1][f & 31][jl & 1] = jj < j1 ? rd[(size_t)jj * 32 + (f & 31)] : 0.f;
        }
        __syncthreads();
        const int tn = min(NN_TILE, j1 - t0);
        for (int jp = 0; jp < (tn + 1) / 2; ++jp) {
            f32x2 acc = {0.0f, 0.0f};
            const float4* r4 = reinterpret_cast<const float4*>(&sr[jp][0][0]);
#pragma unroll
            for (int d2 = 0; d2 < 16; ++d2) {
                const float4 t = r4[d2];                       // components 2 d2, 2 d2 + 1 of references 2 jp, 2 jp + 1
                const f32x2 qa = {q[2 * d2], q[2 * d2]}, qb = {q[2 * d2 + 1], q[2 * d2 + 1]};
                const f32x2 ra = {t.x, t.y}, rb = {t.z, t.w};
                const f32x2 da = qa - ra;
                acc = __builtin_elementwise_fma(da, da, acc);
                const f32x2 db = qb - rb;
                acc = __builtin_elementwise_fma(db, db, acc);
            }
            const unsigned long long k0 = ((unsigned long long)__float_as_uint(acc.x) << 32) | (unsigned)(t0 + 2 * jp);
            best = k0 < best ? k0 : best;
            if (2 * jp + 1 < tn) {
                const unsigned long long k1 = ((unsigned long long)__float_as_uint(acc.y) << 32) | (unsigned)(t0 + 2 * jp + 1);
                best = k1 < best ? k1 : best;
            }
        }
    }
    if (i < nq && j0 < j1) atomicMin(&keys[i], best);
}

__global__ __launch_bounds__(1024) void mutual_kernel(const unsigned long long* __restrict__ skey, int ns,
                                                      const unsigned long long* __restrict__ tkey, int nt,
                                                      int32_t* __restrict__ s_mids, int32_t* __restrict__ t_mids,
                                                      int32_t* __restrict__ count_out, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < ns; i0 += 1024) {
        int i = i0 + tid;
        bool flag = false;
        int snn = 0;
        if (i < ns) {
            snn = (int)(skey[i] & 0xffffffffu);
            if (snn >= 0 && snn < nt) flag = (int)(tkey[snn] & 0xffffffffu) == i;
        }
        unsigned long long bal = __ballot(flag);
        int within = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (flag) { s_mids[off + within] = i; t_mids[off + within] = snn; }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            base_s += tot;
        }
        __syncthreads();
    }
    if (tid == 0) *count_out = base_s;
}

// ------------------------------------------------------------------ softmax + soft-argmax
__global__ void soft_argmax_kernel(const float* __restrict__ logits, const int32_t* __restrict__ m_dev, int max_m,
                                   float* __restrict__ ind, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int m = *m_dev;
    m = m < max_m ? m : max_m;
    if (i >= m) return;
    float c[BX_AZI];
#pragma unroll
    for (int a = 0; a < BX_AZI; ++a) {
        int sl = 4 * ((a & 15) & 3) + ((a & 15) >> 2);
        c[a] = logits[((size_t)i * 2 + (a >> 4)) * 16 + sl];
    }
    float mx = c[0];
#pragma unroll
    for (int a = 1; a < BX_AZI; ++a) mx = c[a] > mx ? c[a] : mx;
    float e[BX_AZI], S = 0.0f;
#pragma unroll
    for (int a = 0; a < BX_AZI; ++a) { e[a] = (float)bxd_exp((double)(c[a] - mx)); S = S + e[a]; }
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < BX_AZI; ++a) acc = fmaf(e[a] / S, (float)a, acc);
    ind[i] = acc;
}

// ------------------------------------------------------------------ hypotheses
__device__ __forceinline__ void mat3mul(const float* A, const float* B, float* C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = fmaf(A[i * 3 + 2], B[2 * 3 + j], fmaf(A[i * 3 + 1], B[1 * 3 + j], A[i * 3 + 0] * B[0 * 3 + j]));
}

__global__ void hypotheses_kernel(const float* __restrict__ ind, const int32_t* __restrict__ s_mids,
                                  const int32_t* __restrict__ t_mids, const int32_t* __restrict__ m_dev, int max_m,
                                  const float* __restrict__ s_R, const float* __restrict__ t_R, const float* __restrict__ s_k,
                                  const float* __restrict__ t_k, float* __restrict__ R_out, float* __restrict__ t_out,
                                  float* __restrict__ ss_out, float* __restrict__ tt_out,
                                  const int32_t* __restrict__ base_dev, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int m = *m_dev;
    m = m < max_m ? m : max_m;
    if (i >= m) return;
    const int si = s_mids[i], ti = t_mids[i];
    const size_t o = (size_t)(base_dev ? *base_dev : 0) + i;
    float angle = ((ind[i] * 2.0f) * BX_PI_F) / (float)BX_AZI + 1e-6f;
    float theta2 = angle * angle;
    float az[9];
    if (theta2 > 1e-6f) {
        float theta = sqrtf(theta2);
        float wz = angle / (theta + 1e-6f);
        double sd, cd;
        bxd_sincos((double)theta, &sd, &cd);
        float sn = (float)sd, cs = (float)cd;
        float ws = wz * sn;
        az[0] = cs; az[1] = 0.0f - ws; az[2] = 0.0f;
        az[3] = ws; az[4] = cs; az[5] = 0.0f;
        az[6] = 0.0f; az[7] = 0.0f; az[8] = cs + (wz * wz) * (1.0f - cs);
    } else {
        az[0] = 1.0f; az[1] = -angle; az[2] = 0.0f;
        az[3] = angle; az[4] = 1.0f; az[5] = 0.0f;
        az[6] = 0.0f; az[7] = 0.0f; az[8] = 1.0f;
    }
    float tR[9], sRt[9], T1[9], R[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) tR[a] = t_R[(size_t)ti * 9 + a];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) sRt[a * 3 + b] = s_R[(size_t)si * 9 + b * 3 + a];
    mat3mul(tR, az, T1);
    mat3mul(T1, sRt, R);
    float s[3] = {s_k[(size_t)si * 3], s_k[(size_t)si * 3 + 1], s_k[(size_t)si * 3 + 2]};
    float t[3] = {t_k[(size_t)ti * 3], t_k[(size_t)ti * 3 + 1], t_k[(size_t)ti * 3 + 2]};
#pragma unroll
    for (int a = 0; a < 9; ++a) R_out[o * 9 + a] = R[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float rs = fmaf(R[a * 3 + 2], s[2], fmaf(R[a * 3 + 1], s[1], R[a * 3 + 0] * s[0]));
        t_out[o * 3 + a] = t[a] - rs;
        ss_out[o * 3 + a] = s[a];
        tt_out[o * 3 + a] = t[a];
    }
}

// ------------------------------------------------------------------ consensus
constexpr int CS_TILE = 256;
__device__ __forceinline__ bool cons_inlier(const float* R, const float* t, float sx, float sy, float sz, float gx, float gy,
                                            float gz, float thr)
{
    float d0 = (fmaf(sz, R[2], fmaf(sy, R[1], sx * R[0])) + t[0]) - gx;
    float d1 = (fmaf(sz, R[5], fmaf(sy, R[4], sx * R[3])) + t[1]) - gy;
    float d2 = (fmaf(sz, R[8], fmaf(sy, R[7], sx * R[6])) + t[2]) - gz;
    float dist = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
    return dist < thr;
}
__device__ __forceinline__ float cons_thr(float sx, float sy, float sz, float inlier_th)
{
    float n = sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
    return ((n * BX_PI_F) / (float)BX_AZI) * inlier_th;
}

__global__ __launch_bounds__(256) void consensus_count_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                              const float* __restrict__ ss, const float* __restrict__ tt,
                                                              const int32_t* __restrict__ M_dev, int max_M, float inlier_th,
                                                              int seg_len, int32_t* __restrict__ cnt, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    __shared__ float sj[CS_TILE][7];
    int M = *M_dev;
    M = M < max_M ? M : max_M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= M) return;
    const int j0 = blockIdx.y * seg_len;
    const int j1 = min(M, j0 + seg_len);
    if (j0 >= j1) return;
    float Ri[9], ti[3];
    if (i < M) {
#pragma unroll
        for (int a = 0; a < 9; ++a) Ri[a] = R[(size_t)i * 9 + a];
#pragma unroll
        for (int a = 0; a < 3; ++a) ti[a] = t[(size_t)i * 3 + a];
    } else {
#pragma unroll
        for (int a = 0; a < 9; ++a) Ri[a] = 0.f;
        ti[0] = ti[1] = ti[2] = 0.f;
    }
    int c = 0;
    for (int t0 = j0; t0 < j1; t0 += CS_TILE) {
        __syncthreads();
        {
            int j = t0 + threadIdx.x;
            if (j < j1) {
                float sx = ss[(size_t)j * 3], sy = ss[(size_t)j * 3 + 1], sz = ss[(size_t)j * 3 + 2];
                sj[threadIdx.x][0] = sx; sj[threadIdx.x][1] = sy; sj[threadIdx.x][2] = sz;
                sj[threadIdx.x][3] = tt[(size_t)j * 3]; sj[threadIdx.x][4] = tt[(size_t)j * 3 + 1]; sj[threadIdx.x][5] = tt[(size_t)j * 3 + 2];
                sj[threadIdx.x][6] = cons_thr(sx, sy, sz, inlier_th);
            }
        }
        __syncthreads();
        const int tn = min(CS_TILE, j1 - t0);
        for (int jj = 0; jj < tn; ++jj)
            c += cons_inlier(Ri, ti, sj[jj][0], sj[jj][1], sj[jj][2], sj[jj][3], sj[jj][4], sj[jj][5], sj[jj][6]) ? 1 : 0;
    }
    if (i < M) atomicAdd(&cnt[i], c);
}

__global__ __launch_bounds__(1024) void consensus_select_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                                const float* __restrict__ ss, const float* __restrict__ tt,
                                                                const int32_t* __restrict__ M_dev, int max_M, float inlier_th,
                                                                const int32_t* __restrict__ cnt, int32_t* __restrict__ inlier_out,
                                                                int32_t* __restrict__ count_out, int32_t* __restrict__ best_out,
                                                                const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    __shared__ unsigned long long wk[16];
    __shared__ int wsum[16];
    __shared__ int base_s, best_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int M = *M_dev;
    M = M < max_M ? M : max_M;
    if (M <= 0) {
        if (tid == 0) { *count_out = 0; if (best_out) *best_out = -1; }
        return;
    }
    // argmax, first maximum: key = (count << 32) | ~i
    unsigned long long key = 0;
    for (int i = tid; i < M; i += 1024) {
        unsigned long long k = ((unsigned long long)(unsigned)cnt[i] << 32) | (unsigned)(~(unsigned)i);
        key = k > key ? k : key;
    }
    key = bx_wave_max(key);
    if (lane == 0) wk[wave] = key;
    if (tid == 0) base_s = 0;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = wk[0];
        for (int w = 1; w < 16; ++w) b = wk[w] > b ? wk[w] : b;
        best_s = (int)(~(unsigned)(b & 0xffffffffu));
    }
    __syncthreads();
    const int bi = best_s;
    float Ri[9], ti[3];
#pragma unroll
    for (int a = 0; a < 9; ++a) Ri[a] = R[(size_t)bi * 9 + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) ti[a] = t[(size_t)bi * 3 + a];
    for (int j0 = 0; j0 < M; j0 += 1024) {
        int j = j0 + tid;
        bool flag = false;
        if (j < M) {
            float sx = ss[(size_t)j * 3], sy = ss[(size_t)j * 3 + 1], sz = ss[(size_t)j * 3 + 2];
            flag = cons_inlier(Ri, ti, sx, sy, sz, tt[(size_t)j * 3], tt[(size_t)j * 3 + 1], tt[(size_t)j * 3 + 2],
                               cons_thr(sx, sy, sz, inlier_th));
        }
        unsigned long long bal = __ballot(flag);
        int within = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (flag) inlier_out[off + within] = j;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            base_s += tot;
        }
        __syncthreads();
    }
    if (tid == 0) { *count_out = base_s; if (best_out) *best_out = bi; }
}
}  // namespace

int bxk_desc_head(bx_ctx* c, hipStream_t s, const float* x, int K, float* desc, float* equi)
{
    if (K <= 0) return BX_OK;
    hipLaunchKernelGGL(desc_head_kernel, dim3(K), dim3(64), 0, s, x, K, c->d_pool_w1, c->d_pool_b1, c->d_pool_w2, c->d_pool_b2, desc, equi, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_mutual(bx_ctx* c, hipStream_t s, const float* sd, int ns, const float* td, int nt, int32_t* s_mids, int32_t* t_mids,
               int32_t* count_out)
{
    if (ns <= 0 || nt <= 0) {
        BX_HIP(hipMemsetAsync(count_out, 0, sizeof(int32_t), s));
        return BX_OK;
    }
    BX_HIP(hipMemsetAsync(c->nn_key[0], 0xff, sizeof(unsigned long long) * (size_t)ns, s));
    BX_HIP(hipMemsetAsync(c->nn_key[1], 0xff, sizeof(unsigned long long) * (size_t)nt, s));
    // reference segments of 128 rows (two LDS tiles): 5000 queries x 16 segments were 1.2 waves per SIMD -- a latency-bound kernel at
    // 111 us per launch; x 40 segments: 58 us (the packed atomicMin keys make the result independent of the segmentation)
    const int nmax = nt > ns ? nt : ns;
    const int SEG = nmax > 16 * 128 ? (nmax + 127) / 128 : 16;
    {
        int seg_len = (nt + SEG - 1) / SEG;
        seg_len = ((seg_len + NN_TILE - 1) / NN_TILE) * NN_TILE;
        dim3 grid((ns + 255) / 256, (nt + seg_len - 1) / seg_len);
        hipLaunchKernelGGL(nn1_kernel, grid, dim3(256), 0, s, sd, ns, td, nt, seg_len, c->nn_key[0], c->skip);
    }
    {
        int seg_len = (ns + SEG - 1) / SEG;
        seg_len = ((seg_len + NN_TILE - 1) / NN_TILE) * NN_TILE;
        dim3 grid((nt + 255) / 256, (ns + seg_len - 1) / seg_len);
        hipLaunchKernelGGL(nn1_kernel, grid, dim3(256), 0, s, td, nt, sd, ns, seg_len, c->nn_key[1], c->skip);
    }
    hipLaunchKernelGGL(mutual_kernel, dim3(1), dim3(1024), 0, s, c->nn_key[0], ns, c->nn_key[1], nt, s_mids, t_mids, count_out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_soft_argmax(hipStream_t s, const float* logits, const int32_t* m_dev, int max_m, float* ind, const int32_t* skip)
{
    if (max_m <= 0) return BX_OK;
    hipLaunchKernelGGL(soft_argmax_kernel, dim3((max_m + 127) / 128), dim3(128), 0, s, logits, m_dev, max_m, ind, skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_hypotheses(hipStream_t s, const float* ind, const int32_t* s_mids, const int32_t* t_mids, const int32_t* m_dev,
                   int max_m, const float* s_R, const float* t_R, const float* s_k, const float* t_k, float* R_out,
                   float* t_out, float* ss_out, float* tt_out, const int32_t* base_dev, const int32_t* skip)
{
    if (max_m <= 0) return BX_OK;
    hipLaunchKernelGGL(hypotheses_kernel, dim3((max_m + 127) / 128), dim3(128), 0, s, ind, s_mids, t_mids, m_dev, max_m, s_R, t_R,
                       s_k, t_k, R_out, t_out, ss_out, tt_out, base_dev, skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_consensus(bx_ctx* c, hipStream_t s, const float* R, const float* t, const float* ss, const float* tt,
                  const int32_t* M_dev, int max_M, int32_t* inlier_out, int32_t* count_out, int32_t* best_out)
{
    if (max_M <= 0) {
        BX_HIP(hipMemsetAsync(count_out, 0, sizeof(int32_t), s));
        return BX_OK;
    }
    BX_HIP(hipMemsetAsync(c->cons_cnt, 0, sizeof(int32_t) * (size_t)max_M, s));
    // segments of one LDS tile: the grid is sized by max_M (3 K) but only the blocks below the device-side M do work -- 16 segments
    // left ~90 active workgroups on 256 CUs
    int seg_len = CS_TILE;
    dim3 grid((max_M + 255) / 256, (max_M + seg_len - 1) / seg_len);
    hipLaunchKernelGGL(consensus_count_kernel, grid, dim3(256), 0, s, R, t, ss, tt, M_dev, max_M, (float)c->p.inlier_th, seg_len, c->cons_cnt, c->skip);
    hipLaunchKernelGGL(consensus_select_kernel, dim3(1), dim3(1024), 0, s, R, t, ss, tt, M_dev, max_M, (float)c->p.inlier_th, c->cons_cnt,
                       inlier_out, count_out, best_out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
