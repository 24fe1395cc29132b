// k_conv.hip -- the convolution stacks on the f32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// Replaces the cuDNN conv stacks of Cylindrical_Net (reference models/patchnet.py:49-84, padding
// utils/common.py:265-310) and CostNet (models/patchnet.py:184-210) plus the gather-expanded cost volume
// of CostVolume.forward (models/BUFFERX.py:59-65).
//
// Formulation: implicit GEMM, rows = (unit, output position), cols = output channels,
// K = (16-channel chunk, tap, channel).  Feature maps live in HBM as [unit][chunk][pos][16] with the
// 16 channels of a chunk in "slot" order (bx_chunk_slot) so that ONE ds_read_b128 per lane feeds the A
// operand of four consecutive 16x16x4 MFMAs in natural channel order.  The f32 MFMA is an exact k-ordered
// fmaf chain, so the result equals the oracle's  acc = bias; for chunk/tap/c: acc = fmaf(x, w, acc)
// bit for bit -- BatchNorm folded, ReLU fused into the epilogue.
//
// Workgroup = 8 waves, G units: the G*P_IN x 16 input slab of one chunk is staged in LDS (80-byte rows ->
// conflict-light b128 reads; double-buffered against the next chunk's global loads), padding / geometry is
// a per-workgroup row-offset table built from the layer's tap table (zero padding -> a shared zero row),
// wave (wm, wn) owns output tiles {wm + t*WM} x 16 channels with fp32 accumulators in VGPRs, weights
// stream from L2 as B fragments (one dword per lane per MFMA, reused across the wave's tiles).
#include "bx_common.h"
#include <cstdlib>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CT = 512;    // threads
constexpr int ROWF = 20;   // floats per LDS row (16 + 4 pad)

template <int NCHUNK, int NTAPS, int P_IN, int P_OUT, int COUT, int G, bool RELU, bool DB>
struct ConvCfg {
    static constexpr int M = G * P_OUT;
    static constexpr int MT = (M + 15) / 16;
    static constexpr int NT = (COUT + 15) / 16;
    static constexpr int WN = NT;
    static constexpr int WM = 8 / WN;
    static constexpr int TPW = (MT + WM - 1) / WM;
    static constexpr int ROWS = G * P_IN;
    static constexpr int BUF_FLOATS = (ROWS + 1) * ROWF;
    static constexpr int NBUF = DB ? 2 : 1;
    static constexpr int NLD = (ROWS * 4 + CT - 1) / CT;
    static constexpr size_t LDS_BYTES = (size_t)NBUF * BUF_FLOATS * 4 + (size_t)NTAPS * MT * 16 * 2;
    // register budget: two co-resident workgroups (4 waves/SIMD, <= 128 VGPRs) whenever the accumulator tile allows it
    static constexpr int MINW = (LDS_BYTES <= 80 * 1024 && (TPW <= 10 || P_IN == 140)) ? 4 : 2;
    // fat accumulator tiles: no one-tap-ahead row-offset registers / pinned A pairs (they would cost the second workgroup)
    static constexpr bool LEAN = TPW > 10;
    static_assert(NT == 2 || NT == 4 || NT == 8, "COUT must give 2/4/8 column tiles");
    static_assert(ROWS + 1 < 65536, "row table is u16");
};

// Persistent workgroups: workgroup b walks the unit groups b, b + gridDim.x, ... and treats (group, chunk) as ONE flat
// sequence of input slabs: while slab s is on the matrix cores, slab s+1 (the next chunk, or chunk 0 of the NEXT group)
// is already in flight from HBM, so neither the per-group prologue (row table, first load) nor the epilogue stores
// leave the MFMA pipe idle.  The row table depends only on the layer geometry and is built once per workgroup.
template <int NCHUNK, int NTAPS, int P_IN, int P_OUT, int COUT, int G, bool RELU, bool DB>
__global__ __launch_bounds__(CT, (ConvCfg<NCHUNK, NTAPS, P_IN, P_OUT, COUT, G, RELU, DB>::MINW)) void conv_kernel(const float* __restrict__ in, const int32_t* __restrict__ units_dev,
                                                  int max_units, const float* __restrict__ W, const float* __restrict__ bias,
                                                  const int32_t* __restrict__ tap, float* __restrict__ out,
                                                  const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    using C = ConvCfg<NCHUNK, NTAPS, P_IN, P_OUT, COUT, G, RELU, DB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* buf = reinterpret_cast<float*>(smem);
    unsigned short* roff = reinterpret_cast<unsigned short*>(buf + (size_t)C::NBUF * C::BUF_FLOATS);

    int units = max_units;
    if (units_dev) { int u = *units_dev; units = u < max_units ? u : max_units; }
    const int ngroups = (units + G - 1) / G;
    if ((int)blockIdx.x >= ngroups) return;
    const int my_groups = (ngroups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nslabs = my_groups * NCHUNK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % C::WN, wm = wave / C::WN;
    const int li = lane & 15, kk = lane >> 4;

    // ---- row-offset table (geometry + padding; rows of units beyond the tail group are loaded as zeros), zero rows
    for (int idx = tid; idx < NTAPS * C::MT * 16; idx += CT) {
        int tp = idx / (C::MT * 16), m = idx - tp * (C::MT * 16);
        int r = C::ROWS;
        if (m < C::M) {
            int g = m / P_OUT, pos = m - g * P_OUT;
            int ip = tap[tp * P_OUT + pos];
            if (ip >= 0) r = g * P_IN + ip;
        }
        roff[idx] = (unsigned short)r;
    }
    if (tid < ROWF * C::NBUF) {
        int b = tid / ROWF;
        buf[(size_t)b * C::BUF_FLOATS + (size_t)C::ROWS * ROWF + (tid - b * ROWF)] = 0.0f;
    }

    // ---- staging helpers: slab s = (group blockIdx.x + (s / NCHUNK) * gridDim.x, chunk s % NCHUNK)
    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4 st[C::NLD];
    auto gload = [&](int s_) {
        const int gi = s_ / NCHUNK, cc = s_ - gi * NCHUNK;
        const int u0 = ((int)blockIdx.x + gi * (int)gridDim.x) * G;
#pragma unroll
        for (int q = 0; q < C::NLD; ++q) {
            int f = tid + q * CT;
            int row = f >> 2, part = f & 3;
            int g = row / P_IN, p = row - g * P_IN;
            if (row < C::ROWS && u0 + g < units)
                st[q] = in4[(((size_t)(u0 + g) * NCHUNK + cc) * P_IN + p) * 4 + part];
            else
                st[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lwrite = [&](int b) {
        float* d = buf + (size_t)b * C::BUF_FLOATS;
#pragma unroll
        for (int q = 0; q < C::NLD; ++q) {
            int f = tid + q * CT;
            int row = f >> 2, part = f & 3;
            if (row < C::ROWS) *reinterpret_cast<float4*>(d + (size_t)row * ROWF + part * 4) = st[q];
        }
    };

    gload(0);
    lwrite(0);
    __syncthreads();

    const int n0 = wn * 16;
    const bool colok = (n0 + li) < COUT;
    const float bv = colok ? bias[n0 + li] : 0.0f;
    f32x4 acc[C::TPW];

    // B fragments (weights, L2-resident) and the A row offsets are fetched ONE TAP AHEAD of the MFMAs that consume
    // them, so neither the ~500-cycle L2 round trip nor the dependent LDS table read sits in front of the matrix pipe.
    // A wave whose last tile does not exist (mt >= MT) runs it on the zero row: uniform schedule, no exec-mask branches.
    const float* wbase = W + (size_t)kk * COUT + n0 + li;
    auto loadB = [&](int ct, float (&b)[4]) {
        const float* wp = wbase + (size_t)ct * 16 * COUT;
        b[0] = colok ? wp[0] : 0.f;
        b[1] = colok ? wp[4 * COUT] : 0.f;
        b[2] = colok ? wp[8 * COUT] : 0.f;
        b[3] = colok ? wp[12 * COUT] : 0.f;
    };
    auto loadR = [&](int tp, int (&r)[C::TPW]) {
        const unsigned short* ro = roff + (size_t)tp * C::MT * 16 + li;
#pragma unroll
        for (int t = 0; t < C::TPW; ++t) {
            const int mt = wm + t * C::WM;
            r[t] = mt < C::MT ? (int)ro[mt * 16] : C::ROWS;
        }
    };
    float bc[4], bn[4];
    int rc[C::TPW], rn[C::TPW];
    loadB(0, bc);
    if constexpr (!C::LEAN) loadR(0, rc);
#pragma unroll
    for (int i = 0; i < 4; ++i) bn[i] = 0.f;
#pragma unroll
    for (int t = 0; t < C::TPW; ++t) rn[t] = C::ROWS;
    // drain the prologue loads here: otherwise the waitcnt pass sees them pending on the loop's entry edge and
    // waits for the NEWEST loads (the counters are in-order) in front of every tap's first MFMAs
    __builtin_amdgcn_s_waitcnt(0);

    const int slot = 4 * (li & 3) + (li >> 2);
    int cc = 0, gi = 0;
    for (int s_ = 0; s_ < nslabs; ++s_) {
        const int cur = DB ? (s_ & 1) : 0;
        if (cc == 0) {
#pragma unroll
            for (int t = 0; t < C::TPW; ++t) acc[t] = (f32x4){bv, bv, bv, bv};
        }
        if (DB && s_ + 1 < nslabs) gload(s_ + 1);
        const float* lb = buf + (size_t)cur * C::BUF_FLOATS + kk * 4;
#pragma unroll 1
        for (int tp = 0; tp < NTAPS; ++tp) {
            {
                int ctn = cc * NTAPS + tp + 1;
                if (ctn == NCHUNK * NTAPS) ctn = 0;      // first tap of the next group's chunk 0 (harmless after the last slab)
                loadB(ctn, bn);
                if constexpr (!C::LEAN) loadR(tp + 1 < NTAPS ? tp + 1 : 0, rn);
            }
#ifdef BX_EXP_NOLDS
#define BX_A(r) (f32x4){(float)(r), bc[1], bc[2], bc[3]}
#else
#define BX_A(r) (*reinterpret_cast<const f32x4*>(lb + (size_t)(r) * ROWF))
#endif
            if constexpr (C::LEAN) {
                const unsigned short* ro = roff + (size_t)tp * C::MT * 16 + li;
#pragma unroll
                for (int t = 0; t < C::TPW; ++t) {
                    const int mt = wm + t * C::WM;
                    if (mt < C::MT) {
                        f32x4 a = BX_A((int)ro[mt * 16]);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bc[0], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bc[1], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bc[2], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bc[3], acc[t], 0, 0, 0);
                    }
                }
            } else {
            // two tiles in flight, the next pair's A operands requested before this pair's MFMAs issue
                f32x4 a0 = BX_A(rc[0]);
                f32x4 a1 = C::TPW > 1 ? BX_A(rc[C::TPW > 1 ? 1 : 0]) : a0;
#pragma unroll
                for (int t = 0; t + 1 < C::TPW; t += 2) {
                    f32x4 n0v = a0, n1v = a1;
                    if (t + 2 < C::TPW) n0v = BX_A(rc[t + 2 < C::TPW ? t + 2 : 0]);
                    if (t + 3 < C::TPW) n1v = BX_A(rc[t + 3 < C::TPW ? t + 3 : 0]);
                    // (pinning this order with sched_barrier measured 10 % SLOWER than the compiler's own schedule)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bc[0], acc[t], 0, 0, 0);
                    acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bc[0], acc[t + 1], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bc[1], acc[t], 0, 0, 0);
                    acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bc[1], acc[t + 1], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bc[2], acc[t], 0, 0, 0);
                    acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bc[2], acc[t + 1], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bc[3], acc[t], 0, 0, 0);
                    acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bc[3], acc[t + 1], 0, 0, 0);
                    a0 = n0v; a1 = n1v;
                }
                if (C::TPW & 1) {
                    constexpr int t = C::TPW - 1;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bc[0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bc[1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bc[2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bc[3], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) bc[i] = bn[i];
            if constexpr (!C::LEAN) {
#pragma unroll
                for (int t = 0; t < C::TPW; ++t) rc[t] = rn[t];
            }
        }
        if (cc == NCHUNK - 1) {
            // ---- epilogue of this group: ReLU, store in chunk-slot order (fire-and-forget; the next slab's MFMAs follow)
            const int u0 = ((int)blockIdx.x + gi * (int)gridDim.x) * G;
            int mrow0 = kk * 4;
            asm volatile("" : "+v"(mrow0));   // keeps the 4*TPW store addresses out of the loop-invariant hoisting (VGPR budget)
#pragma unroll
            for (int t = 0; t < C::TPW; ++t) {
                const int mt = wm + t * C::WM;
                if (mt >= C::MT) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = mt * 16 + mrow0 + r;
                    if (m < C::M) {
                        int g = m / P_OUT, pos = m - g * P_OUT;
                        if (u0 + g < units) {
                            float v = acc[t][r];
                            if (RELU) v = v > 0.0f ? v : 0.0f;
                            out[(((size_t)(u0 + g) * C::NT + wn) * P_OUT + pos) * 16 + slot] = v;
                        }
                    }
                }
            }
        }
#ifdef BX_EXP_NOSYNC
        if (DB) {
            if (s_ + 1 < nslabs && s_ < 0) lwrite(cur ^ 1);
        } else
#endif
        if (DB) {
            if (s_ + 1 < nslabs) lwrite(cur ^ 1);
            __syncthreads();
        } else if (s_ + 1 < nslabs) {
            __syncthreads();
            gload(s_ + 1);
            lwrite(0);
            __syncthreads();
        }
        if (++cc == NCHUNK) { cc = 0; ++gi; }
    }
}

// ---------------------------------------------------------------- CostNet layer 0 on the implicit cost volume
// cost[c][n][k][l] = S[c][k+1][(l-n) mod 20] - T[c][k+1][l]   (models/BUFFERX.py:59-65), valid 3x3x3 conv 32->32.
constexpr int CV_D = BX_AZI, CV_H = BX_ELE - 2, CV_W = BX_AZI;       // 20 x 5 x 20
constexpr int CV_DO = CV_D - 2, CV_HO = CV_H - 2, CV_WO = CV_W - 2;  // 18 x 3 x 18
constexpr int CV_POUT = CV_DO * CV_HO * CV_WO;                        // 972
constexpr int CV_ROWF = 36;                                           // 32 + 4 pad floats per LDS row

__global__ __launch_bounds__(CT) void cost_l1_kernel(const float* __restrict__ s_equi, const float* __restrict__ t_equi,
                                                     const int32_t* __restrict__ s_mids, const int32_t* __restrict__ t_mids,
                                                     const int32_t* __restrict__ m_dev, int max_m, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ out, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    constexpr int MT = (CV_POUT + 15) / 16;  // 61
    constexpr int WN = 2, WM = 4, TPW = (MT + WM - 1) / WM;
    constexpr int NTAPS = 27, COUT = 32;
    __shared__ __attribute__((aligned(16))) float sS[CV_H * CV_W * CV_ROWF];
    __shared__ __attribute__((aligned(16))) float sT[CV_H * CV_W * CV_ROWF];

    int m = *m_dev;
    m = m < max_m ? m : max_m;
    const int u = blockIdx.x;
    if (u >= m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int li = lane & 15, kk = lane >> 4;

    const float* sp = s_equi + ((size_t)s_mids[u] * BX_EA + BX_AZI) * 32;  // elevation rows 1..5
    const float* tp_ = t_equi + ((size_t)t_mids[u] * BX_EA + BX_AZI) * 32;
    for (int f = tid; f < CV_H * CV_W * 32; f += CT) {
        int row = f >> 5, c = f & 31;
        int sl = (c & 16) + 4 * (c & 3) + ((c & 15) >> 2);
        sS[row * CV_ROWF + sl] = sp[f];
        sT[row * CV_ROWF + sl] = tp_[f];
    }
    __syncthreads();

    const int n0 = wn * 16;
    const float bv = bias[n0 + li];
    f32x4 acc[TPW];
    int nkl[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        acc[t] = (f32x4){bv, bv, bv, bv};
        int mrow = (wm + t * WM) * 16 + li;
        if (mrow >= CV_POUT) mrow = CV_POUT - 1;
        int n = mrow / (CV_HO * CV_WO), rem = mrow - n * (CV_HO * CV_WO);
        int k = rem / CV_WO, l = rem - k * CV_WO;
        nkl[t] = (n << 16) | (k << 8) | l;
    }
    for (int cc = 0; cc < 2; ++cc) {
#pragma unroll 1
        for (int tp = 0; tp < NTAPS; ++tp) {
            const int a = tp / 9, b = (tp / 3) % 3, c = tp % 3;
            const float* wp = W + (((size_t)cc * NTAPS + tp) * 16 + kk) * COUT + n0 + li;
            float b0 = wp[0], b1 = wp[4 * COUT], b2 = wp[8 * COUT], b3 = wp[12 * COUT];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int mt = wm + t * WM;
                if (mt < MT) {
                    int n = nkl[t] >> 16, k = (nkl[t] >> 8) & 255, l = nkl[t] & 255;
                    int tc = l + c;
                    int sc = tc - (n + a);
                    sc = sc < 0 ? sc + CV_W : sc;
                    int rbase = (k + b) * CV_W;
                    f32x4 sv = *reinterpret_cast<const f32x4*>(sS + (size_t)(rbase + sc) * CV_ROWF + cc * 16 + kk * 4);
                    f32x4 tv = *reinterpret_cast<const f32x4*>(sT + (size_t)(rbase + tc) * CV_ROWF + cc * 16 + kk * 4);
                    f32x4 av = sv - tv;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b3, acc[t], 0, 0, 0);
                }
            }
        }
    }
    const int slot = 4 * (li & 3) + (li >> 2);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int mt = wm + t * WM;
        if (mt >= MT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int mrow = mt * 16 + kk * 4 + r;
            if (mrow < CV_POUT) {
                float v = acc[t][r];
                v = v > 0.0f ? v : 0.0f;
                out[(((size_t)u * 2 + wn) * CV_POUT + mrow) * 16 + slot] = v;
            }
        }
    }
}

template <int NCHUNK, int NTAPS, int P_IN, int P_OUT, int COUT, int G, bool RELU, bool DB>
int launch_conv(hipStream_t s, const ConvLayerDev& L, const float* in, const int32_t* units_dev, int max_units, float* out,
                const int32_t* skip)
{
    using C = ConvCfg<NCHUNK, NTAPS, P_IN, P_OUT, COUT, G, RELU, DB>;
    if (L.nchunk != NCHUNK || L.ntaps != NTAPS || L.p_in != P_IN || L.p_out != P_OUT || L.cout != COUT || (L.relu != 0) != RELU) {
        bx_set_error("conv layer geometry mismatch (%d %d %d %d %d)", L.nchunk, L.ntaps, L.p_in, L.p_out, L.cout);
        return BX_ERR_STATE;
    }
    auto k = conv_kernel<NCHUNK, NTAPS, P_IN, P_OUT, COUT, G, RELU, DB>;
    static bool attr_set = false;
    if (!attr_set) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    int grid = (max_units + G - 1) / G;
    if (grid <= 0) return BX_OK;
    // persistent workgroups: as many as are co-resident (LDS- and register-limited), each walking its groups
    static int wg_per_cu = 0, n_cu = 0;
    if (!wg_per_cu) {
        int dev = 0, occ = 0;
        hipDeviceProp_t prop;
        BX_HIP(hipGetDevice(&dev));
        BX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
        BX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, CT, C::LDS_BYTES));
        wg_per_cu = 1 << 20;   // default: one unit group per workgroup (measured faster than the persistent walk)
        (void)occ;
        const char* e = getenv("BX_CONV_WGPCU");
        if (e && atoi(e) > 0) wg_per_cu = atoi(e);
    }
    const long long cap = (long long)wg_per_cu * n_cu;
    if ((long long)grid > cap) grid = (int)cap;
    hipLaunchKernelGGL(k, dim3(grid), dim3(CT), C::LDS_BYTES, s, in, units_dev, max_units, L.W, L.b, L.tap, out, skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
}  // namespace

int bxk_conv(bx_ctx* c, hipStream_t s, int net, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (net == 0) {
        const ConvLayerDev& L = c->desc[layer];
        // Units per workgroup chosen by measurement (tools/gpu_conv.sh, K = 5000): 64- and 32-channel layers run 2 units
        // per workgroup (9 accumulator tiles per wave, <= 128 VGPRs, 2-3 workgroups per CU); the 128-channel layers
        // keep 2 units (18 tiles per wave, "lean" loop) -- 1 unit per workgroup halves the reuse of the B fragments.
        switch (layer) {
            //                     NCHUNK taps P_IN P_OUT COUT G  RELU  DB
            case 0: return launch_conv<3, 9, 140, 140, 64, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 1: return launch_conv<4, 9, 140, 140, 64, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 2: return launch_conv<4, 9, 140, 140, 128, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 3: return launch_conv<8, 9, 140, 140, 128, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 4: return launch_conv<8, 9, 140, 140, 64, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 5: return launch_conv<4, 9, 140, 140, 64, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 6: return launch_conv<4, 9, 140, 140, 32, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 7: return launch_conv<2, 9, 140, 140, 32, 2, false, true>(s, L, in, units_dev, max_units, out, c->skip);
        }
    } else if (net == 1) {
        const ConvLayerDev& L = c->pose[layer];
        switch (layer) {
            case 1: return launch_conv<2, 27, 972, 256, 64, 1, true, false>(s, L, in, units_dev, max_units, out, c->skip);
            case 2: return launch_conv<4, 9, 256, 196, 64, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 3: return launch_conv<4, 9, 196, 144, 128, 2, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 4: return launch_conv<8, 9, 144, 100, 128, 4, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 5: return launch_conv<8, 9, 100, 64, 64, 8, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 6: return launch_conv<4, 9, 64, 36, 64, 8, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 7: return launch_conv<4, 9, 36, 16, 32, 16, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 8: return launch_conv<2, 9, 16, 4, 32, 32, true, true>(s, L, in, units_dev, max_units, out, c->skip);
            case 9: return launch_conv<2, 4, 4, 1, 20, 128, false, true>(s, L, in, units_dev, max_units, out, c->skip);
        }
    }
    bx_set_error("bxk_conv: bad net/layer %d/%d", net, layer);
    return BX_ERR_ARG;
}

int bxk_cost_l1(bx_ctx* c, hipStream_t s, const float* s_equi, const float* t_equi, const int32_t* s_mids,
                const int32_t* t_mids, const int32_t* m_dev, int max_m, float* out)
{
    if (max_m <= 0) return BX_OK;
    const ConvLayerDev& L = c->pose[0];
    hipLaunchKernelGGL(cost_l1_kernel, dim3(max_m), dim3(CT), 0, s, s_equi, t_equi, s_mids, t_mids, m_dev, max_m, L.W, L.b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
