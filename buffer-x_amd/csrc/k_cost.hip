// k_cost.hip -- CostNet layer 0 on the IMPLICIT cost volume, collapsed (round 3).
//
// Reference: CostVolume.forward (models/BUFFERX.py:59-65) builds cost[c][n][k][l] = S[c][k][(l - n) mod 20] - T[c][k][l]
// (20 shifted copies of the source map minus the target map, 256 KB per match) and CostNet's first layer
// (models/patchnet.py:196: Conv3d(32, 32, 3x3x3) + BatchNorm + ReLU) convolves it: 26.9 MMAC per match, a third of CostNet.
// The convolution is linear and the volume has only 2 x (5 x 20) independent values per channel, so
//     out[o][n][k][l] = relu( b[o] + P[o][k][(l - n) mod 20] - Q[o][k][l] )            n, l in [0, 18), k in [0, 3)
//     P[o][k][e] = sum_{c, b, delta = -2..2} Wp[c][b][delta][o] * S[c][k + b][(e + delta) mod 20],  Wp = sum_{d - a = delta} W[o][c][a][b][d]
//     Q[o][k][l] = sum_{c, b, d = 0..2}      Wq[c][b][d][o]     * T[c][k + b][l + d],               Wq = sum_a W[o][c][a][b][d]
// -- 1.42 MMAC per match (a 15-tap circular convolution of S on a 3 x 20 map and a 9-tap convolution of T on a 3 x 18 map).
// P and Q are individually larger than their difference, so the contract (oracle/bx_oracle.c: bxo_cost_l0) is binary64:
// weights summed in binary64, accumulators from 0 through an fma chain in the order c > b > delta | d, ((b + P) - Q) rounded to
// fp32 once, ReLU.  Plain v_fma_f64 on the vector ALUs (full rate on gfx950): one workgroup per match, a wave owns OT output
// channels of P or of Q, a lane one map position; the weights of a step are wave-uniform and arrive through the scalar cache.
// The expanded layer-0 output [m][2][972][16] (what layer 1 consumes) is written with 16-byte coalesced stores.
#include "bx_common.h"
#include <cstdlib>
#include <vector>

namespace {
constexpr int A = BX_AZI, H = BX_ELE - 2;                 // 20 x 5 input rows (elevation rows 1..5)
constexpr int AO = A - 2, HO = H - 2;                      // 18 x 3
constexpr int POUT = AO * HO * AO;                         // 972 output positions
constexpr int SW = A + 4;                                  // S rows carry a +-2 column wrap-around halo
constexpr int SCS = H * SW + 1, TCS = H * A + 1;           // channel strides (odd: conflict-free transposing writes)

// OT output channels per wave: the weights of a step are OT doubles in SGPRs (one s_load), the step costs 1 LDS read + 1 convert
// + OT fma.  The chain waits on the weight stream through the scalar cache (197 KB of weights per match against a 16 KB cache).
// Measured (m = 1458 matches): OT = 8 with 8 waves per workgroup 181 us, OT = 4 with 16 waves (8 waves per SIMD) 200 us -- the
// extra waves do not pay for the lower fma share of a step; OT = 8 is the default (BX_COST_OT=4 selects the other form).
template <int OT, int ND, int ROWW, int CS>
__device__ __forceinline__ void pq_chain(const float* __restrict__ sm, int base, const double* __restrict__ w, double (&acc)[OT])
{
#pragma unroll 1
    for (int c = 0; c < 32; ++c) {
        const float* r = sm + c * CS + base;
        const double* wc = w + (size_t)c * 3 * ND * OT;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const double x = (double)r[b * ROWW + d];
                const double* ws = wc + (b * ND + d) * OT;
#pragma unroll
                for (int j = 0; j < OT; ++j) acc[j] = fma(ws[j], x, acc[j]);
            }
    }
}

template <int OT>
__global__ __launch_bounds__(64 * (64 / OT), 8) void cost_l0_kernel(const float* __restrict__ s_equi, const float* __restrict__ t_equi,
                                                        const int32_t* __restrict__ s_mids, const int32_t* __restrict__ t_mids,
                                                        const int32_t* __restrict__ m_dev, int max_m, const double* __restrict__ Wp,
                                                        const double* __restrict__ Wq, const float* __restrict__ bias,
                                                        float* __restrict__ out, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    // LDS: the staged maps (28.4 KB) and, once every chain has finished with them, b + P / Q (32 KB) in the SAME bytes: 32 KB per
    // workgroup instead of 61 KB, i.e. five resident workgroups per CU -- the chains wait on the scalar weight stream, more waves hide it
    __shared__ __attribute__((aligned(16))) char smem[32 * 64 * 8 * 2];
    static_assert(sizeof(float) * 32 * (SCS + TCS) <= sizeof(smem), "maps fit the P / Q bytes");
    float* sS = reinterpret_cast<float*>(smem);
    float* sT = sS + 32 * SCS;
    double* sP = reinterpret_cast<double*>(smem);          // [o][k*20 + e]  = b[o] + P
    double* sQ = sP + 32 * 64;                              // [o][k*18 + l]
    int m = *m_dev;
    m = m < max_m ? m : max_m;
    const int u = blockIdx.x;
    if (u >= m) return;
    constexpr int NTILE = 32 / OT, CT = 64 * 2 * NTILE;     // waves [0, NTILE): P tiles, [NTILE, 2 NTILE): Q tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- stage S (with the wrap-around halo) and T, channel-major
    const float* sp = s_equi + ((size_t)s_mids[u] * BX_EA + A) * 32;      // elevation rows 1..5
    const float* tp = t_equi + ((size_t)t_mids[u] * BX_EA + A) * 32;
    for (int f = tid; f < H * A * 32; f += CT) {
        const int row = f >> 5, c = f & 31;
        const int k = row / A, l = row - k * A;
        const float sv = sp[f];
        float* d = sS + c * SCS + k * SW;
        d[l + 2] = sv;
        if (l >= A - 2) d[l + 2 - A] = sv;      // columns -2, -1
        if (l < 2) d[l + 2 + A] = sv;           // columns 20, 21
        sT[c * TCS + row] = tp[f];
    }
    __syncthreads();

    // ---- P (first half of the waves) and Q (second half): OT output channels per wave, one map position per lane
    {
        const int ot = wave % NTILE;
        double acc[OT];
#pragma unroll
        for (int j = 0; j < OT; ++j) acc[j] = 0.0;
        if (wave < NTILE) {
            const int pos = lane < HO * A ? lane : 0;
            const int k = pos / A, e = pos - k * A;
            pq_chain<OT, 5, SW, SCS>(sS, k * SW + e, Wp + (size_t)ot * 32 * 15 * OT, acc);
        } else {
            const int pos = lane < HO * AO ? lane : 0;
            const int k = pos / AO, l = pos - k * AO;
            pq_chain<OT, 3, A, TCS>(sT, k * A + l, Wq + (size_t)ot * 32 * 9 * OT, acc);
        }
        __syncthreads();                    // every chain is done with the maps: their bytes become P / Q
        if (wave < NTILE) {
            if (lane < HO * A) {
#pragma unroll
                for (int j = 0; j < OT; ++j) sP[(ot * OT + j) * 64 + lane] = (double)bias[ot * OT + j] + acc[j];
            }
        } else if (lane < HO * AO) {
#pragma unroll
            for (int j = 0; j < OT; ++j) sQ[(ot * OT + j) * 64 + lane] = acc[j];
        }
    }
    __syncthreads();

    // ---- expansion: out[u][chunk][pos][slot], slot 4j + i holds channel j + 4i of the chunk (bx_chunk_slot); 16 bytes per thread
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)u * 2 * POUT * 16);
    for (int f = tid; f < 2 * POUT * 4; f += CT) {
        const int ch = f / (POUT * 4), rem = f - ch * (POUT * 4);
        const int pos = rem >> 2, j = rem & 3;
        const int n = pos / (HO * AO), r = pos - n * (HO * AO);
        const int k = r / AO, l = r - k * AO;
        int e = l - n;
        e = e < 0 ? e + A : e;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = ch * 16 + j + 4 * i;
            const double d = sP[o * 64 + k * A + e] - sQ[o * 64 + k * AO + l];
            const float x = (float)d;
            v[i] = x > 0.0f ? x : 0.0f;
        }
        o4[f] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
}  // namespace

// Wp / Wq of the collapsed form from the packed layer-0 weights [2][27][16][32] (tap = (a*3 + b)*3 + d): binary64 sums in the
// order of the oracle (a ascending), device layout [o / OT][c][b][delta | d][o % OT].
static int cost_ot()
{
    static int v = 0;
    if (!v) { const char* e = getenv("BX_COST_OT"); v = (e && atoi(e) == 4) ? 4 : 8; }
    return v;
}

int bxk_cost_l0_weights(const float* w0, double** d_wp, double** d_wq)
{
    const int OT = cost_ot();
    std::vector<double> Wp((size_t)32 * 3 * 5 * 32, 0.0), Wq((size_t)32 * 3 * 3 * 32, 0.0);
    for (int c = 0; c < 32; ++c)
        for (int b = 0; b < 3; ++b)
            for (int a = 0; a < 3; ++a)
                for (int d = 0; d < 3; ++d)
                    for (int o = 0; o < 32; ++o) {
                        const double w = (double)w0[(((size_t)(c / 16) * 27 + (a * 3 + b) * 3 + d) * 16 + (c % 16)) * 32 + o];
                        Wp[(((size_t)c * 3 + b) * 5 + (d - a + 2)) * 32 + o] += w;
                        Wq[(((size_t)c * 3 + b) * 3 + d) * 32 + o] += w;
                    }
    std::vector<double> dp(Wp.size()), dq(Wq.size());
    for (int o = 0; o < 32; ++o)
        for (int c = 0; c < 32; ++c)
            for (int b = 0; b < 3; ++b) {
                for (int d = 0; d < 5; ++d) dp[((((size_t)(o / OT) * 32 + c) * 3 + b) * 5 + d) * OT + (o % OT)] = Wp[(((size_t)c * 3 + b) * 5 + d) * 32 + o];
                for (int d = 0; d < 3; ++d) dq[((((size_t)(o / OT) * 32 + c) * 3 + b) * 3 + d) * OT + (o % OT)] = Wq[(((size_t)c * 3 + b) * 3 + d) * 32 + o];
            }
    BX_HIP(hipMalloc(reinterpret_cast<void**>(d_wp), dp.size() * sizeof(double)));
    BX_HIP(hipMemcpy(*d_wp, dp.data(), dp.size() * sizeof(double), hipMemcpyHostToDevice));
    BX_HIP(hipMalloc(reinterpret_cast<void**>(d_wq), dq.size() * sizeof(double)));
    BX_HIP(hipMemcpy(*d_wq, dq.data(), dq.size() * sizeof(double), hipMemcpyHostToDevice));
    return BX_OK;
}

int bxk_cost_l0(bx_ctx* c, hipStream_t s, const float* s_equi, const float* t_equi, const int32_t* s_mids, const int32_t* t_mids,
                const int32_t* m_dev, int max_m, float* out)
{
    if (max_m <= 0) return BX_OK;
    if (cost_ot() == 8)
        hipLaunchKernelGGL(cost_l0_kernel<8>, dim3(max_m), dim3(512), 0, s, s_equi, t_equi, s_mids, t_mids, m_dev, max_m, c->d_cost_wp, c->d_cost_wq,
                           c->pose[0].b, out, c->skip);
    else
        hipLaunchKernelGGL(cost_l0_kernel<4>, dim3(max_m), dim3(1024), 0, s, s_equi, t_equi, s_mids, t_mids, m_dev, max_m, c->d_cost_wp, c->d_cost_wq,
                           c->pose[0].b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
