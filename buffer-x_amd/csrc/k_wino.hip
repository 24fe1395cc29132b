// k_wino.hip -- Cylindrical_Net layers as Winograd F(2x2, 3x3) convolutions on the f32 matrix cores (round 3).
//
// Same layers as conv_kernel of k_conv.hip (reference models/patchnet.py:49-84; padding utils/common.py:265-310: circular in
// azimuth, zero in elevation) with 2.25x fewer multiplications (Lavin & Gray 2016): the 7 x 20 map is cut into 4 x 10 output tiles
// of 2 x 2 (the 8th output row does not exist and is dropped); per tile and channel the 4 x 4 input window d becomes
// V = B^T d B (additions only), the channel contraction M[xi][nu] = sum_c V[xi][nu][c] U[xi][nu][c][o] is SIXTEEN independent
// GEMMs (rows = tiles, K = input channels, columns = output channels) on v_mfma_f32_16x16x4_f32, and Y = A^T M A (+ bias, ReLU)
// folds the sixteen results back into the 2 x 2 outputs.  U = G g G^T is computed on the host in binary64 and rounded once.
// The arithmetic contract is restated by oracle/bx_oracle.c::bxo_conv_wino; GPU == oracle bit for bit.
//
// Two kernels: wino_pair_kernel (Cylindrical_Net layers with >= 64 output channels under bx_params.desc_conv_form = winograd22: two units
// per workgroup, 2 x 40 tiles = five full MFMA row tiles, single-buffered slab + V planes, serialised phases) and wino_pose_kernel
// (CostNet layers 1..5 as VALID F(2x2, 3x3) convolutions, the default pose_conv_form).  Wave (ct, half) owns the column tile ct and the
// eight planes of rows xi = 2 half, 2 half + 1.  Output transform (wino_output): every wave folds its two rows into
// r_xi[j] = (M0 + M1) + M2 | (M1 - M2) - M3 lane-locally and hands its two quantities -- half 0: r0 + r1 and r1, half 1: r2 and r3 --
// to an LDS exchange (bytes of the consumed plane set) that all threads then read in output order: Y[0][j] = ((r0 + r1) + r2),
// Y[1][j] = ((r1 - r2) - r3), + bias, ReLU, 16-byte stores.  The round-3 one-unit pipelined kernel (transform of one wave beside the
// MFMAs of its SIMD sibling: 8 100 cycles per unit and chunk against 7 170 serialised) was removed in round 4; LABBOOK.md section 2 keeps
// its measurements.
#include "bx_common.h"
#include <cstdlib>
#include <vector>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWF = 20;                         // floats per LDS row (16 + 4 pad)
constexpr int WP = BX_AZI + 2;                   // slab columns (wrap-around halo)
constexpr int HP = BX_ELE + 3;                   // slab rows: h = -1 .. ele_n + 1 (tile row 3 reaches two rows below the map)
constexpr int SLAB_FLOATS = HP * WP * ROWF;      // 4400
constexpr int TR = (BX_ELE + 1) / 2, TC = BX_AZI / 2, NT_ = TR * TC;   // 4 x 10 = 40 tiles
constexpr int VROWS = 48;                        // tile rows per plane, padded to three MFMA row tiles
constexpr int VPLANE = VROWS * ROWF;             // floats per (xi, nu) plane
constexpr size_t WINO_LDS = (size_t)(2 * SLAB_FLOATS + 2 * 16 * VPLANE) * 4;   // 2 x 17.6 KB + 2 x 61.4 KB = 158 KB
constexpr int CW = 64, CT = 512;                 // output channels / threads of a workgroup
static_assert(WINO_LDS <= 160 * 1024, "double-buffered slab + V planes fit the LDS");
static_assert(4 * NT_ * CW <= 16 * VPLANE, "E of one output column fits the bytes of one set of V planes");

// Output transform shared by the kernels below.  Lane (li, kk) of wave (ct, half) holds M[xi][0..3] (xi = 2 half, 2 half + 1) of tile
// rows R = rt*16 + 4 kk + r for ONE output channel; r_xi[j] = (M0 + M1) + M2 | (M1 - M2) - M3 is lane-local.  Per output column j
// every wave writes its two quantities -- half 0: r0 + r1 and r1, half 1: r2 and r3 -- of all its tile rows to LDS as
// E[quantity][R][64 channels of the workgroup in OUTPUT slot order], 16-channel groups XOR-swizzled by kk = (R >> 2) & 3 (the four
// kk of a wave hit four different bank groups: conflict-free b32 writes, and a row stays 64 contiguous floats for b128 reads);
// then ALL threads walk (R, 4-channel quad) in output order: four ds_read_b128, Y[0][j] = ((r0 + r1) + r2) + bias,
// Y[1][j] = ((r1 - r2) - r3) + bias, ReLU, two 16-byte stores (a lane-local finish would store 4 bytes per lane: six times the
// store instructions).  E of one j = 4 x ROWS x 64 floats lives in the bytes of a consumed set of V planes.
template <int RT, int ROWS, bool RELU, class StoreFn>
__device__ __forceinline__ void wino_output(const f32x4 (&acc)[8][RT], float* ex, int half, int ctl, int li, int kk, int tid,
                                            const float4 b4, StoreFn&& store)
{
    int kko = kk, slot = ctl * 16 + 4 * (li & 3) + (li >> 2), tq = tid;
    asm volatile("" : "+v"(kko), "+v"(slot), "+v"(tq));   // keeps the LDS / store addresses out of loop-invariant hoisting
    const int wcol = slot ^ (kko << 4);
    constexpr int NITEM = ROWS * 16, NIT = (NITEM + CT - 1) / CT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = rt * 16 + kko * 4 + r;
                const float ra = j == 0 ? (acc[0][rt][r] + acc[1][rt][r]) + acc[2][rt][r] : (acc[1][rt][r] - acc[2][rt][r]) - acc[3][rt][r];
                const float rb = j == 0 ? (acc[4][rt][r] + acc[5][rt][r]) + acc[6][rt][r] : (acc[5][rt][r] - acc[6][rt][r]) - acc[7][rt][r];
                const float q0 = half == 0 ? ra + rb : ra;          // half 0: r0 + r1 | half 1: r2
                if (rt * 16 + 15 < ROWS || R < ROWS) {
                    ex[((half * 2 + 0) * ROWS + R) * 64 + wcol] = q0;
                    ex[((half * 2 + 1) * ROWS + R) * 64 + wcol] = rb;   // half 0: r1 | half 1: r3
                }
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = tq + k * CT;
            if (it < NITEM) {
                const int R = it >> 4, quad = it & 15;
                const float* e = ex + R * 64 + ((quad * 4) ^ (((R >> 2) & 3) << 4));
                const float4 s01 = *reinterpret_cast<const float4*>(e);
                const float4 r1 = *reinterpret_cast<const float4*>(e + ROWS * 64);
                const float4 r2 = *reinterpret_cast<const float4*>(e + 2 * ROWS * 64);
                const float4 r3 = *reinterpret_cast<const float4*>(e + 3 * ROWS * 64);
                float4 y0 = make_float4((s01.x + r2.x) + b4.x, (s01.y + r2.y) + b4.y, (s01.z + r2.z) + b4.z, (s01.w + r2.w) + b4.w);
                float4 y1 = make_float4(((r1.x - r2.x) - r3.x) + b4.x, ((r1.y - r2.y) - r3.y) + b4.y, ((r1.z - r2.z) - r3.z) + b4.z,
                                        ((r1.w - r2.w) - r3.w) + b4.w);
                if (RELU) {
                    y0 = make_float4(y0.x > 0.f ? y0.x : 0.f, y0.y > 0.f ? y0.y : 0.f, y0.z > 0.f ? y0.z : 0.f, y0.w > 0.f ? y0.w : 0.f);
                    y1 = make_float4(y1.x > 0.f ? y1.x : 0.f, y1.y > 0.f ? y1.y : 0.f, y1.z > 0.f ? y1.z : 0.f, y1.w > 0.f ? y1.w : 0.f);
                }
                store(R, j, quad, y0, y1);
            }
        }
        __syncthreads();             // E(j) consumed: the next j (or the next transform) may overwrite it
    }
}

// ---------------------------------------------------------------------------------------------------- two units per workgroup
// wino_pair_kernel: the Cylindrical_Net layer with TWO units per workgroup: 2 x 40 tiles fill FIVE MFMA row tiles exactly (the
// one-unit kernel above pads 40 tiles to 48 rows: one MFMA in six multiplies padding).  Single-buffered slab (2 units) + V planes
// (16 x 80 rows): 137 KB, two barriers per chunk (the structure of wino_pose_kernel below); wave (ct, half) owns 8 planes x 5 row
// tiles = 160 accumulator VGPRs.  The default for all six Winograd layers (bxk_wino); 5-11 % faster than the one-unit kernel once the
// slab traffic's addressing is computed once per thread instead of once per chunk.
constexpr int GP = 2, ROWSP = GP * NT_, RTP = ROWSP / 16, VPLP = ROWSP * ROWF;
constexpr size_t WINOP_LDS = (size_t)(GP * SLAB_FLOATS + 16 * VPLP) * 4;
static_assert(ROWSP % 16 == 0 && WINOP_LDS <= 160 * 1024 && 4 * ROWSP * CW <= 16 * VPLP, "two units: five full row tiles, LDS, exchange");

template <int NCHUNK, int COUT, bool RELU>
__global__ __launch_bounds__(CT, 2) void wino_pair_kernel(const float* __restrict__ in, int units, const float* __restrict__ U,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    constexpr int NT = COUT / 16;
    constexpr int NPU = BX_EA * 4, NPIECE = GP * NPU, NLD = (NPIECE + CT - 1) / CT;
    constexpr int NITEM = 4 * ROWSP * 4, NIT = (NITEM + CT - 1) / CT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    float* Vp = slab + GP * SLAB_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, ctl = wave >> 1;
    const int ctg = (int)blockIdx.y * (CW / 16) + ctl;
    const int li = lane & 15, kk = lane >> 4;
    const int ngroups = (units + GP - 1) / GP;
    if ((int)blockIdx.x >= ngroups) return;

    for (int i = tid; i < (int)(WINOP_LDS / 16); i += CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4 st[NLD];
    // per-thread constants of the slab traffic: piece f = tid + q CT of the group's two units -> source offset inside the group's
    // [2][NCHUNK][140][16] floats and destination row in the slab (+ the wrap-around copy of azimuth columns 0 / 19); computed once --
    // recomputing them per chunk was ~100 instructions of integer division per wave and chunk (1 050 cycles in the s_memtime profile)
    int lsrc[NLD], ldst[NLD], lhalo[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int f = tid + q * CT;
        lsrc[q] = -1; ldst[q] = 0; lhalo[q] = 0;
        if (f < NPIECE) {
            const int g = f >= NPU ? 1 : 0, fr = f - g * NPU;
            const int p = fr >> 2, part = fr & 3;
            const int h = p / BX_AZI, w = p - h * BX_AZI;
            lsrc[q] = g * NCHUNK * NPU + fr;
            ldst[q] = g * SLAB_FLOATS + ((h + 1) * WP + (w + 1)) * ROWF + part * 4;
            lhalo[q] = w == 0 ? BX_AZI * ROWF : (w == BX_AZI - 1 ? -BX_AZI * ROWF : 0);
        }
    }
    auto gload = [&](int ug, int cc) {
        const float4* base = in4 + ((size_t)ug * GP * NCHUNK + cc) * NPU;
        const bool second = ug * GP + 1 < units;               // the last group of an odd unit count has one unit
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const bool ok = lsrc[q] >= 0 && (second || lsrc[q] < NCHUNK * NPU);
            st[q] = ok ? bx_ld_stream(base + lsrc[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lwrite = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            if (lsrc[q] >= 0) {
                float* d = slab + ldst[q];
                *reinterpret_cast<float4*>(d) = st[q];
                if (lhalo[q] != 0) *reinterpret_cast<float4*>(d + lhalo[q]) = st[q];
            }
        }
    };
    int ia[NIT], ib[NIT], iv[NIT];
    float isg[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = tid + k * CT;
        ia[k] = -1; ib[k] = 0; iv[k] = 0; isg[k] = 0.f;
        if (it < NITEM) {
            const int x = it / (ROWSP * 4), rem = it - x * (ROWSP * 4);
            const int R = rem >> 2, part = rem & 3;
            const int g = R / NT_, t = R - g * NT_;
            const int tr = t / TC, tc = t - tr * TC;
            const int ra = x == 0 ? 0 : (x == 2 ? 2 : 1);
            const int rb = x == 0 ? 2 : (x == 1 ? 2 : (x == 2 ? 1 : 3));
            isg[k] = x == 1 ? 1.0f : -1.0f;
            ia[k] = g * SLAB_FLOATS + ((2 * tr + ra) * WP + 2 * tc) * ROWF + part * 4;
            ib[k] = g * SLAB_FLOATS + ((2 * tr + rb) * WP + 2 * tc) * ROWF + part * 4;
            iv[k] = (x * 4) * VPLP + R * ROWF + part * 4;
        }
    }
    auto transform = [&]() {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (ia[k] < 0) continue;
            const float sg = isg[k];
            float4 tj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(slab + ia[k] + j * ROWF);
                const float4 b = *reinterpret_cast<const float4*>(slab + ib[k] + j * ROWF);
                tj[j] = make_float4(fmaf(sg, b.x, a.x), fmaf(sg, b.y, a.y), fmaf(sg, b.z, a.z), fmaf(sg, b.w, a.w));
            }
            float* vd = Vp + iv[k];
            *reinterpret_cast<float4*>(vd) = make_float4(tj[0].x - tj[2].x, tj[0].y - tj[2].y, tj[0].z - tj[2].z, tj[0].w - tj[2].w);
            *reinterpret_cast<float4*>(vd + VPLP) = make_float4(tj[1].x + tj[2].x, tj[1].y + tj[2].y, tj[1].z + tj[2].z, tj[1].w + tj[2].w);
            *reinterpret_cast<float4*>(vd + 2 * VPLP) = make_float4(tj[2].x - tj[1].x, tj[2].y - tj[1].y, tj[2].z - tj[1].z, tj[2].w - tj[1].w);
            *reinterpret_cast<float4*>(vd + 3 * VPLP) = make_float4(tj[1].x - tj[3].x, tj[1].y - tj[3].y, tj[1].z - tj[3].z, tj[1].w - tj[3].w);
        }
    };

    const float* bq = bias + ((int)blockIdx.y * (CW / 16) + ((tid & 15) >> 2)) * 16 + (tid & 3);
    const float4 b4 = make_float4(bq[0], bq[4], bq[8], bq[12]);
    const float4* wbase = reinterpret_cast<const float4*>(U) + ((size_t)(half * 8) * NT + ctg) * 64 + lane;
    const char* abase = reinterpret_cast<const char*>(Vp) + ((half * 8 * ROWSP + li) * ROWF + kk * 4) * 4;

    f32x4 acc[8][RTP];
    float4 bring[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) bring[p] = wbase[((size_t)p * NT) * 64];

    int ug = blockIdx.x;
    gload(ug, 0);
    __syncthreads();
    lwrite();

    for (;;) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int rt = 0; rt < RTP; ++rt) acc[p][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ugn = ug + (int)gridDim.x;
#pragma unroll 1
        for (int cc = 0; cc < NCHUNK; ++cc) {
            __syncthreads();
            // the next chunk's slab is requested in front of the transform and written right behind its barrier: the loads have the
            // transform to land, the registers that carry them are free during the MFMAs, and the write's s_waitcnt vmcnt(0) does
            // not wait for B fragments requested a moment ago (as it did behind the MFMA loop)
            const bool more = cc + 1 < NCHUNK || ugn < ngroups;
            if (cc + 1 < NCHUNK) gload(ug, cc + 1);
            else if (ugn < ngroups) gload(ugn, 0);
            transform();
            __syncthreads();
            if (more) lwrite();
            const int cn = cc + 1 == NCHUNK ? 0 : cc + 1;
            f32x4 ar[3];
            ar[0] = *reinterpret_cast<const f32x4*>(abase);
            ar[1] = *reinterpret_cast<const f32x4*>(abase + (16 * ROWF) * 4);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 bqq = bring[p & 3];
#pragma unroll
                for (int rt = 0; rt < RTP; ++rt) {
                    const int s0 = p * RTP + rt, s2 = s0 + 2;
                    if (s2 < 8 * RTP) ar[s2 % 3] = *reinterpret_cast<const f32x4*>(abase + (((s2 / RTP) * ROWSP + (s2 % RTP) * 16) * ROWF) * 4);
                    const f32x4 a = ar[s0 % 3];
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bqq.x, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bqq.y, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bqq.z, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bqq.w, acc[p][rt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                bring[p & 3] = p < 4 ? wbase[((size_t)(cc * 16 + p + 4) * NT) * 64] : wbase[((size_t)(cn * 16 + p - 4) * NT) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();             // every wave is done with the V planes: their bytes carry E now
        wino_output<RTP, ROWSP, RELU>(acc, Vp, half, ctl, li, kk, tid, b4,
                                      [&](int R, int j, int quad, const float4& y0, const float4& y1) {
                                          const int g = R >= NT_ ? 1 : 0, t = R - g * NT_;
                                          const int u = ug * GP + g;
                                          if (u < units) {
                                              const int tr = t / TC, tc = t - tr * TC;
                                              float* ou = out + ((((size_t)u * NT + (int)blockIdx.y * (CW / 16) + (quad >> 2)) * BX_EA +
                                                                  (2 * tr) * BX_AZI + 2 * tc + j) * 16 + (quad & 3) * 4);
                                              bx_st_stream(reinterpret_cast<float4*>(ou), y0);
                                              if (2 * tr + 1 < BX_ELE) bx_st_stream(reinterpret_cast<float4*>(ou + BX_AZI * 16), y1);
                                          }
                                      });
        ug = ugn;
        if (ug >= ngroups) break;
    }
}

template <int NCHUNK, int COUT, bool RELU>
int launch_wino_pair(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, int units, float* out)
{
    if (L.nchunk != NCHUNK || L.cout != COUT || (L.relu != 0) != RELU || !L.Wwino) {
        bx_set_error("winograd layer %d: geometry mismatch (%d chunks, %d channels)", layer, L.nchunk, L.cout);
        return BX_ERR_STATE;
    }
    auto k = wino_pair_kernel<NCHUNK, COUT, RELU>;
    int& cap = c->wino_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINOP_LDS));
        cap = c->n_cu / (COUT / CW);
        if (cap < 1) cap = 1;
        if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
    }
    int grid = (units + GP - 1) / GP;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k, dim3(grid, COUT / CW), dim3(CT), WINOP_LDS, s, in, units, L.Wwino, L.b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

// ---------------------------------------------------------------------------------------------------- CostNet layers 1..5
// CostNet's layers 1..5 (models/patchnet.py:197-201: k(3,3,3) on [18][3][18], then k(3,1,3) on [16][1][16] .. [10][1][10], un-padded)
// are 2-D valid 3x3 convolutions over (n, l); layer 1 folds its three k rows into the channel dimension (6 effective chunks
// e = chunk * 3 + k).  Same transform, same wave roles and output exchange as wino_kernel; differences: valid geometry (a D x D slab
// without halo, ((D - 2) / 2)^2 tiles per unit), G units per workgroup so that the tile rows fill 3-4 MFMA row tiles whatever the map
// size (G = 1, 1, 1, 2, 4 for D = 18 .. 10), the unit count is a device-side value (mutual matches of the scale), and -- as the
// transform and the MFMAs of different waves do not overlap on this chip anyway (every non-MFMA instruction costs matrix-pipe
// time: the double-buffered pipeline of wino_kernel ran at the speed of its single-buffered predecessor) -- slab and V planes are
// single-buffered: two barriers per chunk, next slab fetched into registers under the MFMAs.  Restated by bxo_conv_wino_valid.
template <int NE, int FOLD, int COUT, int D, int G, bool RELU>
__global__ __launch_bounds__(CT, 2) void wino_pose_kernel(const float* __restrict__ in, const int32_t* __restrict__ units_dev, int max_units,
                                                          const float* __restrict__ U, const float* __restrict__ bias, float* __restrict__ out,
                                                          const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    constexpr int NT = COUT / 16, NCHUNK = NE / FOLD;
    constexpr int TT = (D - 2) / 2, NTU = TT * TT, DO = D - 2, PIN = D * FOLD * D, POUT = DO * DO;
    constexpr int ROWS = G * NTU, RT = (ROWS + 15) / 16, VR = RT * 16, VPL = VR * ROWF;
    constexpr int SLABF = G * D * D * ROWF;
    constexpr int NPIECE = G * D * D * 4, NLD = (NPIECE + CT - 1) / CT;
    constexpr int NITEM = 4 * ROWS * 4, NIT = (NITEM + CT - 1) / CT;
    static_assert(ROWS <= 64 && 4 * ROWS * CW <= 16 * VPL, "tile rows fit four MFMA row tiles; E of one output column fits the V planes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    float* Vp = slab + SLABF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, ctl = wave >> 1;
    const int ctg = (int)blockIdx.y * (CW / 16) + ctl;
    const int li = lane & 15, kk = lane >> 4;

    int units = max_units;
    if (units_dev) { const int ud = *units_dev; units = ud < max_units ? ud : max_units; }
    const int ngroups = (units + G - 1) / G;
    if ((int)blockIdx.x >= ngroups) return;

    for (int i = tid; i < (SLABF + 16 * VPL) / 4; i += CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4 st[NLD];
    auto gload = [&](int ug, int e) {
        const int cc = e / FOLD, b = e - cc * FOLD;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int f = tid + q * CT;
            const int row = f >> 2, part = f & 3;
            const int g = row / (D * D), p = row - g * (D * D);
            const int n = p / D, l = p - n * D;
            const int u = ug * G + g;
            st[q] = (f < NPIECE && u < units) ? in4[(((size_t)u * NCHUNK + cc) * PIN + (n * FOLD + b) * D + l) * 4 + part]
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lwrite = [&]() {
        int tq = tid;
        asm volatile("" : "+v"(tq));
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int f = tq + q * CT;
            if (f < NPIECE) *reinterpret_cast<float4*>(slab + (f >> 2) * ROWF + (f & 3) * 4) = st[q];
        }
    };
    int ia[NIT], ib[NIT], iv[NIT];
    float isg[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = tid + k * CT;
        ia[k] = -1; ib[k] = 0; iv[k] = 0; isg[k] = 0.f;
        if (it < NITEM) {
            const int x = it / (ROWS * 4), rem = it - x * (ROWS * 4);
            const int R = rem >> 2, part = rem & 3;
            const int g = R / NTU, t = R - g * NTU;
            const int tr = t / TT, tc = t - tr * TT;
            const int ra = x == 0 ? 0 : (x == 2 ? 2 : 1);
            const int rb = x == 0 ? 2 : (x == 1 ? 2 : (x == 2 ? 1 : 3));
            isg[k] = x == 1 ? 1.0f : -1.0f;
            ia[k] = (g * D * D + (2 * tr + ra) * D + 2 * tc) * ROWF + part * 4;
            ib[k] = (g * D * D + (2 * tr + rb) * D + 2 * tc) * ROWF + part * 4;
            iv[k] = (x * 4) * VPL + R * ROWF + part * 4;
        }
    }
    auto transform = [&]() {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (ia[k] < 0) continue;
            const float sg = isg[k];
            float4 tj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(slab + ia[k] + j * ROWF);
                const float4 b = *reinterpret_cast<const float4*>(slab + ib[k] + j * ROWF);
                tj[j] = make_float4(fmaf(sg, b.x, a.x), fmaf(sg, b.y, a.y), fmaf(sg, b.z, a.z), fmaf(sg, b.w, a.w));
            }
            float* vd = Vp + iv[k];
            *reinterpret_cast<float4*>(vd) = make_float4(tj[0].x - tj[2].x, tj[0].y - tj[2].y, tj[0].z - tj[2].z, tj[0].w - tj[2].w);
            *reinterpret_cast<float4*>(vd + VPL) = make_float4(tj[1].x + tj[2].x, tj[1].y + tj[2].y, tj[1].z + tj[2].z, tj[1].w + tj[2].w);
            *reinterpret_cast<float4*>(vd + 2 * VPL) = make_float4(tj[2].x - tj[1].x, tj[2].y - tj[1].y, tj[2].z - tj[1].z, tj[2].w - tj[1].w);
            *reinterpret_cast<float4*>(vd + 3 * VPL) = make_float4(tj[1].x - tj[3].x, tj[1].y - tj[3].y, tj[1].z - tj[3].z, tj[1].w - tj[3].w);
        }
    };

    const float* bq = bias + ((int)blockIdx.y * (CW / 16) + ((tid & 15) >> 2)) * 16 + (tid & 3);
    const float4 b4 = make_float4(bq[0], bq[4], bq[8], bq[12]);
    const float4* wbase = reinterpret_cast<const float4*>(U) + ((size_t)(half * 8) * NT + ctg) * 64 + lane;
    const char* abase = reinterpret_cast<const char*>(Vp) + ((half * 8 * VR + li) * ROWF + kk * 4) * 4;

    f32x4 acc[8][RT];
    // B fragments in a ring of four: plane p + 4 (of this chunk, or plane p - 4 of the next) is requested right after the MFMAs of
    // plane p -- 4 RT x 4 MFMAs ahead, longer than an L2 round trip
    float4 bring[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) bring[p] = wbase[((size_t)p * NT) * 64];      // effective chunk 0, planes 0..3

    int ug = blockIdx.x;
    gload(ug, 0);
    __syncthreads();                 // zero fill complete
    lwrite();

    for (;;) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[p][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ugn = ug + (int)gridDim.x;
#pragma unroll 1
        for (int e = 0; e < NE; ++e) {
            __syncthreads();         // slab of chunk e in place; every wave is done with the V planes of the chunk before
            transform();
            __syncthreads();         // V complete; the slab is free
            const bool more = e + 1 < NE || ugn < ngroups;
            if (e + 1 < NE) gload(ug, e + 1);
            else if (ugn < ngroups) gload(ugn, 0);
            const int en = e + 1 == NE ? 0 : e + 1;
            // 8 planes x RT row tiles, A operand two steps ahead in a ring of three (static indices: no register copies)
            f32x4 ar[3];
            ar[0] = *reinterpret_cast<const f32x4*>(abase);
            ar[1] = *reinterpret_cast<const f32x4*>(abase + ((RT > 1 ? 16 : VR) * ROWF) * 4);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 bq = bring[p & 3];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int s0 = p * RT + rt, s2 = s0 + 2;
                    if (s2 < 8 * RT) ar[s2 % 3] = *reinterpret_cast<const f32x4*>(abase + (((s2 / RT) * VR + (s2 % RT) * 16) * ROWF) * 4);
                    const f32x4 a = ar[s0 % 3];
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc[p][rt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                bring[p & 3] = p < 4 ? wbase[((size_t)(e * 16 + p + 4) * NT) * 64] : wbase[((size_t)(en * 16 + p - 4) * NT) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) lwrite();
        }
        __syncthreads();             // every wave is done with the V planes: their bytes carry E now
        wino_output<RT, ROWS, RELU>(acc, Vp, half, ctl, li, kk, tid, b4,
                                    [&](int R, int j, int quad, const float4& y0, const float4& y1) {
                                        const int g = R / NTU, t = R - g * NTU;
                                        const int u = ug * G + g;
                                        if (u < units) {
                                            const int tr = t / TT, tc = t - tr * TT;
                                            float* ou = out + ((((size_t)u * NT + (int)blockIdx.y * (CW / 16) + (quad >> 2)) * POUT +
                                                                (2 * tr) * DO + 2 * tc + j) * 16 + (quad & 3) * 4);
                                            *reinterpret_cast<float4*>(ou) = y0;   // plain stores: the next CostNet layer finds the map in L2
                                            *reinterpret_cast<float4*>(ou + DO * 16) = y1;
                                        }
                                    });
        ug = ugn;
        if (ug >= ngroups) break;
    }
}

template <int NE, int FOLD, int COUT, int D, int G, bool RELU>
int launch_wino_pose(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, const int32_t* units_dev, int max_units,
                     float* out)
{
    constexpr int TT = (D - 2) / 2, ROWS = G * TT * TT, RT = (ROWS + 15) / 16;
    constexpr size_t LDS = (size_t)(G * D * D * ROWF + 16 * RT * 16 * ROWF) * 4;
    static_assert(LDS <= 160 * 1024, "slab + V planes fit the LDS");
    if (L.nchunk * FOLD != NE || L.cout != COUT || (L.relu != 0) != RELU || !L.Wwino || L.ntaps != 9 * FOLD) {
        bx_set_error("winograd CostNet layer %d: geometry mismatch (%d chunks, %d taps, %d channels)", layer, L.nchunk, L.ntaps, L.cout);
        return BX_ERR_STATE;
    }
    auto k = wino_pose_kernel<NE, FOLD, COUT, D, G, RELU>;
    int& cap = c->wino_pose_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        int occ = 0;
        BX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, CT, LDS));
        cap = (occ >= 1 ? occ : 1) * c->n_cu / (COUT / CW);
        if (cap < 1) cap = 1;
        if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
    }
    int grid = (max_units + G - 1) / G;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k, dim3(grid, COUT / CW), dim3(CT), LDS, s, in, units_dev, max_units, L.Wwino, L.b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
}  // namespace

// U = G g G^T of every (chunk, channel, output channel) in binary64, rounded once (the same expressions as the oracle's
// wino_filter), packed as B fragments [chunk * 16 + plane][column tile][lane = kk*16 + li][4], element i = U[plane][chunk][kk + 4 i][col]
int bxk_wino_weights(const float* w /* [nchunk][9 * fold][16][cout] */, int nchunk, int fold, int cout, float** d_out)
{
    // fold = 3: the k(3,3,3) CostNet layer, tap = (a*3 + b)*3 + d, effective chunk e = chunk * 3 + b; fold = 1: tap = a*3 + d
    const int nt = cout / 16, ntaps = 9 * fold;
    std::vector<float> frag((size_t)nchunk * fold * 16 * nt * 64 * 4, 0.0f);
    for (int cs = 0; cs < nchunk; ++cs)
      for (int fb = 0; fb < fold; ++fb) {
        const int cc = cs * fold + fb;                     // effective chunk
        for (int ch = 0; ch < 16; ++ch)
            for (int o = 0; o < cout; ++o) {
                double g[3][3], Gg[4][3];
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw)
                        g[kh][kw] = (double)w[(((size_t)cs * ntaps + (fold == 3 ? (kh * 3 + fb) * 3 + kw : kh * 3 + kw)) * 16 + ch) * cout + o];
                for (int kw = 0; kw < 3; ++kw) {
                    Gg[0][kw] = g[0][kw];
                    Gg[1][kw] = 0.5 * ((g[0][kw] + g[1][kw]) + g[2][kw]);
                    Gg[2][kw] = 0.5 * ((g[0][kw] - g[1][kw]) + g[2][kw]);
                    Gg[3][kw] = g[2][kw];
                }
                for (int xi = 0; xi < 4; ++xi) {
                    double uu[4];
                    uu[0] = Gg[xi][0];
                    uu[1] = 0.5 * ((Gg[xi][0] + Gg[xi][1]) + Gg[xi][2]);
                    uu[2] = 0.5 * ((Gg[xi][0] - Gg[xi][1]) + Gg[xi][2]);
                    uu[3] = Gg[xi][2];
                    for (int nu = 0; nu < 4; ++nu) {
                        const int pl = xi * 4 + nu, kk = ch & 3, i = ch >> 2, t = o / 16, li = o % 16;
                        frag[((((size_t)(cc * 16 + pl) * nt + t) * 4 + kk) * 16 + li) * 4 + i] = (float)uu[nu];
                    }
                }
            }
      }
    BX_HIP(hipMalloc(reinterpret_cast<void**>(d_out), frag.size() * sizeof(float)));
    BX_HIP(hipMemcpy(*d_out, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice));
    return BX_OK;
}

// layer of Cylindrical_Net in the Winograd form; returns -1 when this layer / unit count is not served (caller falls back)
int bxk_wino(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (units_dev || max_units < 1) return -1;
    const ConvLayerDev& L = c->desc[layer];
    switch (layer) {
        case 0: return launch_wino_pair<3, 64, true>(c, layer, s, L, in, max_units, out);
        case 1: return launch_wino_pair<4, 64, true>(c, layer, s, L, in, max_units, out);
        case 2: return launch_wino_pair<4, 128, true>(c, layer, s, L, in, max_units, out);
        case 3: return launch_wino_pair<8, 128, true>(c, layer, s, L, in, max_units, out);
        case 4: return launch_wino_pair<8, 64, true>(c, layer, s, L, in, max_units, out);
        case 5: return launch_wino_pair<4, 64, true>(c, layer, s, L, in, max_units, out);
        // layers 6 and 7 (32 output channels: half a workgroup) stay on the direct kernels in this form
    }
    return -1;
}

// CostNet layers 1..5 in the Winograd form; returns -1 for the other layers (caller falls back to the direct kernels)
int bxk_wino_pose(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (max_units < 1) return -1;
    const ConvLayerDev& L = c->pose[layer];
    switch (layer) {
        //                             NE FOLD COUT  D  G
        case 1: return launch_wino_pose<6, 3, 64, 18, 1, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 2: return launch_wino_pose<4, 1, 64, 16, 1, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 3: return launch_wino_pose<4, 1, 128, 14, 1, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 4: return launch_wino_pose<8, 1, 128, 12, 2, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 5: return launch_wino_pose<8, 1, 64, 10, 4, true>(c, layer, s, L, in, units_dev, max_units, out);
    }
    return -1;
}
