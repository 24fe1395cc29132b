// k_wino43.hip -- Cylindrical_Net layers as Winograd F(4x4, 3x3) convolutions on the f32 matrix cores (round 3, BX_DESC_CONV=winograd43).
//
// Same layers as k_wino.hip (reference models/patchnet.py:49-84; padding utils/common.py:265-310) with 4 x 4 output tiles: the 7 x 20 map
// is cut into 2 x 5 tiles (the 8th output row does not exist); per tile and channel the 6 x 6 input window d becomes V = B^T d B, the
// channel contraction is THIRTY-SIX independent GEMMs M[xi][nu] = V[xi][nu] U[xi][nu] (rows = tiles, K = input channels, columns =
// output channels) on v_mfma_f32_16x16x4_f32, and Y = A^T M A (+ bias, ReLU) folds them into the 16 outputs: 36 multiplications per 16
// outputs against 16 per 4 (F(2x2, 3x3)) and 9 per 1 (direct) -- 10 tile rows x 36 planes per unit = 0.29x the direct form's MFMA work
// (F(2x2): 0.51x).  The transforms hold non-dyadic constants (B^T: 4, -5, 2; G: 1/4, 1/6, 1/24; A^T up to 8), so the error against a
// binary64 convolution is ~4x F(2x2)'s and ~2.4x the direct fp32 form's (rms; tests/study_wino43_error.py) -- still fp32-grade: every
// reference-minted fixture incl. the three real-size ones keeps identical counts / mutual sets / consensus sets
// (tests/study_wino43_pipeline.py on the CPU emulation, then the GPU suite under the switch).  The arithmetic contract is restated by
// oracle/bx_oracle.c::bxo_conv_wino43; GPU == oracle bit for bit.
//
// Workgroup = 8 waves, 64 output channels of THREE units (30 tile rows = two MFMA row tiles, 2 padding rows): wave (ct, half) owns the
// column tile ct and the EIGHTEEN planes of rows xi = 3 half .. 3 half + 2: 36 accumulator tiles = 144 VGPRs.  Phases per 16-channel
// chunk, serialised (k_wino.hip: on this chip a VALU / LDS wave beside an MFMA wave costs more than it hides): barrier, input transform
// (item = (xi, tile row, 4-channel quad): the xi row of B^T d down the six columns, then B^T along the row: six V planes; the item type
// xi is wave-uniform), barrier, request the next slab, MFMAs (B fragments in a ring of three planes, A operands two steps ahead), write
// the slab.  Output transform: nu pass and the half's partial xi sums lane-local; per output row i of the tiles the two halves drop
// their four partials P_h[i][0..3] into an LDS exchange (swizzled as in k_wino.hip) and all threads finish Y = (P_0 + P_1) + bias in
// output order: 256 contiguous bytes per (tile, column tile).
#include "bx_common.h"
#include <cstdlib>
#include <vector>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWF = 20;                         // floats per LDS row (16 + 4 pad)
constexpr int WP = BX_AZI + 2;                   // slab columns (wrap-around halo)
constexpr int HP = BX_ELE + 3;                   // slab rows h = -1 .. 8 (tile row 1 reaches two rows below the map)
constexpr int SLAB_FLOATS = HP * WP * ROWF;      // 4400
constexpr int TR4 = (BX_ELE + 3) / 4, TC4 = BX_AZI / 4, NT4 = TR4 * TC4;   // 2 x 5 = 10 tiles
constexpr int G4 = 3, ROWS4 = G4 * NT4, RT4 = (ROWS4 + 15) / 16, VR4 = RT4 * 16, VPL4 = VR4 * ROWF;   // 30 tile rows -> 32
constexpr int NPL = 36, NPH = 18;                // planes, planes per wave half
constexpr size_t W43_LDS = (size_t)(G4 * SLAB_FLOATS + NPL * VPL4) * 4;    // 52.8 KB + 92.2 KB
constexpr int CW = 64, CT = 512;
static_assert(NPH % 3 == 0 && BX_AZI % 4 == 0 && W43_LDS <= 160 * 1024 && 8 * VR4 * CW <= NPL * VPL4, "geometry, LDS, exchange of one output row");

// the six results of B^T on a 6-vector (contract: bxo_conv_wino43)
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&o)[6])
{
#define BX_C4(expr) make_float4(expr(x), expr(y), expr(z), expr(w))
#define T0(c) fmaf(4.0f, d[0].c, fmaf(-5.0f, d[2].c, d[4].c))
#define TA(c) fmaf(-4.0f, d[2].c, d[4].c)
#define TB(c) fmaf(-4.0f, d[1].c, d[3].c)
#define TC_(c) (d[4].c - d[2].c)
#define TE(c) (2.0f * (d[3].c - d[1].c))
#define T5(c) fmaf(4.0f, d[1].c, fmaf(-5.0f, d[3].c, d[5].c))
    o[0] = BX_C4(T0);
    const float4 a = BX_C4(TA), b = BX_C4(TB), c = BX_C4(TC_), e = BX_C4(TE);
    o[1] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    o[2] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    o[3] = make_float4(c.x + e.x, c.y + e.y, c.z + e.z, c.w + e.w);
    o[4] = make_float4(c.x - e.x, c.y - e.y, c.z - e.z, c.w - e.w);
    o[5] = BX_C4(T5);
#undef T0
#undef TA
#undef TB
#undef TC_
#undef TE
#undef T5
#undef BX_C4
}

template <int NCHUNK, int COUT, bool RELU>
__global__ __launch_bounds__(CT, 2) void wino43_kernel(const float* __restrict__ in, int units, const float* __restrict__ U,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    constexpr int NT = COUT / 16;
    constexpr int NPU = BX_EA * 4, NPIECE = G4 * NPU, NLD = (NPIECE + CT - 1) / CT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    float* Vp = slab + G4 * SLAB_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, ctl = wave >> 1;
    const int ctg = (int)blockIdx.y * (CW / 16) + ctl;
    const int li = lane & 15, kk = lane >> 4;
    const int ngroups = (units + G4 - 1) / G4;
    if ((int)blockIdx.x >= ngroups) return;

    for (int i = tid; i < (int)(W43_LDS / 16); i += CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- slab traffic: per-thread constants (source offset inside the group's [3][NCHUNK][140][16] floats, destination row, halo copy)
    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4 st[NLD];
    int lsrc[NLD], ldst[NLD], lhalo[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int f = tid + q * CT;
        lsrc[q] = -1; ldst[q] = 0; lhalo[q] = 0;
        if (f < NPIECE) {
            const int g = f / NPU, fr = f - g * NPU;
            const int p = fr >> 2, part = fr & 3;
            const int h = p / BX_AZI, w = p - h * BX_AZI;
            lsrc[q] = g * NCHUNK * NPU + fr;
            ldst[q] = g * SLAB_FLOATS + ((h + 1) * WP + (w + 1)) * ROWF + part * 4;
            lhalo[q] = w == 0 ? BX_AZI * ROWF : (w == BX_AZI - 1 ? -BX_AZI * ROWF : 0);
        }
    }
    auto gload = [&](int ug, int cc) {
        const float4* base = in4 + ((size_t)ug * G4 * NCHUNK + cc) * NPU;
        const int lim = (units - ug * G4) * NCHUNK * NPU;       // pieces of units that do not exist read as zeros
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            // streamed once: non-temporal, so that the activations do not push the B fragments (re-read by every workgroup) out of L2
            const f32x4 v = (lsrc[q] >= 0 && lsrc[q] < lim) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + lsrc[q])) : (f32x4){0.f, 0.f, 0.f, 0.f};
            st[q] = make_float4(v.x, v.y, v.z, v.w);
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

    // ---- transform items: wave-item wi = wave + 8 k (k = 0, 1; 12 wave-items): xi = wi >> 1 is wave-uniform, (tile row, quad) =
    //      (wi & 1) * 64 + lane (120 of 128 used).  Per SIMD: wave s does two wave-items, wave s + 4 one.
    int tsrc[2], tdst[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int wi = wave + 8 * k;
        const int sub = (wi & 1) * 64 + lane;
        tsrc[k] = -1; tdst[k] = 0;
        if (wi < 12 && sub < ROWS4 * 4) {
            const int R = sub >> 2, part = sub & 3;
            const int g = R / NT4, t = R - g * NT4;
            const int tr = t / TC4, tc = t - tr * TC4;
            tsrc[k] = g * SLAB_FLOATS + ((4 * tr) * WP + 4 * tc) * ROWF + part * 4;
            tdst[k] = ((wi >> 1) * 6) * VPL4 + R * ROWF + part * 4;
        }
    }
    auto transform = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int xi = (wave + 8 * k) >> 1;                 // wave-uniform
            if (wave + 8 * k >= 12) continue;
            if (tsrc[k] >= 0) {
                const float* s0 = slab + tsrc[k];
                float4 t[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {                   // the xi row of B^T d, column j
                    const float* sc = s0 + j * ROWF;
                    if (xi == 0 || xi == 5) {
                        const int i0 = xi == 0 ? 0 : 1;
                        const float4 d0 = *reinterpret_cast<const float4*>(sc + (i0 * WP) * ROWF);
                        const float4 d2 = *reinterpret_cast<const float4*>(sc + ((i0 + 2) * WP) * ROWF);
                        const float4 d4 = *reinterpret_cast<const float4*>(sc + ((i0 + 4) * WP) * ROWF);
                        t[j] = make_float4(fmaf(4.0f, d0.x, fmaf(-5.0f, d2.x, d4.x)), fmaf(4.0f, d0.y, fmaf(-5.0f, d2.y, d4.y)),
                                           fmaf(4.0f, d0.z, fmaf(-5.0f, d2.z, d4.z)), fmaf(4.0f, d0.w, fmaf(-5.0f, d2.w, d4.w)));
                    } else {
                        const float4 d1 = *reinterpret_cast<const float4*>(sc + (1 * WP) * ROWF);
                        const float4 d2 = *reinterpret_cast<const float4*>(sc + (2 * WP) * ROWF);
                        const float4 d3 = *reinterpret_cast<const float4*>(sc + (3 * WP) * ROWF);
                        const float4 d4 = *reinterpret_cast<const float4*>(sc + (4 * WP) * ROWF);
                        if (xi <= 2) {
                            const float4 a = make_float4(fmaf(-4.0f, d2.x, d4.x), fmaf(-4.0f, d2.y, d4.y), fmaf(-4.0f, d2.z, d4.z), fmaf(-4.0f, d2.w, d4.w));
                            const float4 b = make_float4(fmaf(-4.0f, d1.x, d3.x), fmaf(-4.0f, d1.y, d3.y), fmaf(-4.0f, d1.z, d3.z), fmaf(-4.0f, d1.w, d3.w));
                            const float sg = xi == 1 ? 1.0f : -1.0f;            // a + b | a - b (fmaf(+-1, b, a) is exact)
                            t[j] = make_float4(fmaf(sg, b.x, a.x), fmaf(sg, b.y, a.y), fmaf(sg, b.z, a.z), fmaf(sg, b.w, a.w));
                        } else {
                            const float4 c = make_float4(d4.x - d2.x, d4.y - d2.y, d4.z - d2.z, d4.w - d2.w);
                            const float4 e = make_float4(2.0f * (d3.x - d1.x), 2.0f * (d3.y - d1.y), 2.0f * (d3.z - d1.z), 2.0f * (d3.w - d1.w));
                            const float sg = xi == 3 ? 1.0f : -1.0f;
                            t[j] = make_float4(fmaf(sg, e.x, c.x), fmaf(sg, e.y, c.y), fmaf(sg, e.z, c.z), fmaf(sg, e.w, c.w));
                        }
                    }
                }
                float4 o[6];
                bt6(t, o);                                      // along the row: the six planes (xi, 0..5)
                float* vd = Vp + tdst[k];
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<float4*>(vd + nu * VPL4) = o[nu];
            }
        }
    };

    // bias of the four output slots this thread stores in the output transform (as k_wino.hip)
    const float* bq = bias + ((int)blockIdx.y * (CW / 16) + ((tid & 15) >> 2)) * 16 + (tid & 3);
    const float4 b4 = make_float4(bq[0], bq[4], bq[8], bq[12]);
    // B fragments [chunk * 36 + plane][column tile][lane][4]; this wave's planes are half * 18 + 0..17
    const float4* wbase = reinterpret_cast<const float4*>(U) + ((size_t)(half * NPH) * NT + ctg) * 64 + lane;
    const char* abase = reinterpret_cast<const char*>(Vp) + ((half * NPH * VR4 + li) * ROWF + kk * 4) * 4;

    f32x4 acc[NPH][RT4];
    // B fragments in a ring of THREE planes (18 planes per wave: the ring size must divide it so that plane q of the next chunk lands in
    // the slot plane q is read from): plane p + 3 is requested right after the MFMAs of plane p
    float4 bring[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bring[p] = wbase[((size_t)p * NT) * 64];

    int ug = blockIdx.x;
    gload(ug, 0);
    __syncthreads();                 // zero fill complete
    lwrite();

    for (;;) {
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int rt = 0; rt < RT4; ++rt) acc[p][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ugn = ug + (int)gridDim.x;
#pragma unroll 1
        for (int cc = 0; cc < NCHUNK; ++cc) {
            __syncthreads();         // slab of chunk cc in place; every wave is done with the V planes of the chunk before
            // the next slab is requested in front of the transform and written right behind its barrier: by then these loads AND the B
            // fragments requested at the end of the previous MFMA phase have landed (behind the MFMA loop the write's s_waitcnt vmcnt(0)
            // waited an L2 round trip for fragments requested a moment earlier: 1 500-2 000 cycles per chunk in the s_memtime profile)
            const bool more = cc + 1 < NCHUNK || ugn < ngroups;
            if (cc + 1 < NCHUNK) gload(ug, cc + 1);
            else if (ugn < ngroups) gload(ugn, 0);
            transform();
            __syncthreads();         // V complete; the slab is free
            if (more) lwrite();
            const int cn = cc + 1 == NCHUNK ? 0 : cc + 1;
            f32x4 ar[3];
            ar[0] = *reinterpret_cast<const f32x4*>(abase);
            ar[1] = *reinterpret_cast<const f32x4*>(abase + (16 * ROWF) * 4);
#pragma unroll
            for (int p = 0; p < NPH; ++p) {
                const float4 bqq = bring[p % 3];
                // the slot is refilled BEFORE the plane's MFMAs (they read the copy): four planes of look-ahead from a ring of three
                bring[p % 3] = p + 3 < NPH ? wbase[((size_t)(cc * NPL + p + 3) * NT) * 64] : wbase[((size_t)(cn * NPL + p + 3 - NPH) * NT) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < RT4; ++rt) {
                    const int s0 = p * RT4 + rt, s2 = s0 + 2;
                    if (s2 < NPH * RT4) ar[s2 % 3] = *reinterpret_cast<const f32x4*>(abase + (((s2 / RT4) * VR4 + (s2 % RT4) * 16) * ROWF) * 4);
                    const f32x4 a = ar[s0 % 3];
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bqq.x, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bqq.y, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bqq.z, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bqq.w, acc[p][rt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();             // every wave is done with the V planes: their bytes carry the exchange now

        // ---- output transform.  nu pass lane-local: r_xi[j] of the half's three xi rows, then what the xi pass needs of them:
        //      half 0: (r_0, pp = r_1 + r_2, qq = r_1 - r_2); half 1: (ss = r_3 + r_4, tt = r_3 - r_4, r_5)
        float ua[RT4][4][4], ub[RT4][4][4], uc[RT4][4][4];      // [rt][r][j]
#pragma unroll
        for (int rt = 0; rt < RT4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float rr[3][4];
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const float m0 = acc[x * 6 + 0][rt][r], m1 = acc[x * 6 + 1][rt][r], m2 = acc[x * 6 + 2][rt][r], m3 = acc[x * 6 + 3][rt][r],
                                m4 = acc[x * 6 + 4][rt][r], m5 = acc[x * 6 + 5][rt][r];
                    const float p = m1 + m2, q = m1 - m2, s = m3 + m4, t = m3 - m4;
                    rr[x][0] = (m0 + p) + s;
                    rr[x][1] = fmaf(2.0f, t, q);
                    rr[x][2] = fmaf(4.0f, s, p);
                    rr[x][3] = fmaf(8.0f, t, q) + m5;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (half == 0) { ua[rt][r][j] = rr[0][j]; ub[rt][r][j] = rr[1][j] + rr[2][j]; uc[rt][r][j] = rr[1][j] - rr[2][j]; }
                    else           { ua[rt][r][j] = rr[0][j] + rr[1][j]; ub[rt][r][j] = rr[0][j] - rr[1][j]; uc[rt][r][j] = rr[2][j]; }
                }
            }
        float* ex = Vp;
        int kko = kk, slot = ctl * 16 + 4 * (li & 3) + (li >> 2), tq = tid;
        asm volatile("" : "+v"(kko), "+v"(slot), "+v"(tq));
        const int wcol = slot ^ (kko << 4);
        // this thread's output item: (tile row R, 4-channel quad)
        const int oR = tq >> 4, oquad = tq & 15;
        int ooff = -1, otr = 0;
        if (oR < ROWS4) {
            const int g = oR / NT4, t = oR - g * NT4;
            const int u = ug * G4 + g;
            otr = t / TC4;
            const int tc = t - otr * TC4;
            if (u < units)
                ooff = (((u * NT + (int)blockIdx.y * (CW / 16) + (oquad >> 2)) * BX_EA + (4 * otr) * BX_AZI + 4 * tc) * 16 + (oquad & 3) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int rt = 0; rt < RT4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int R = rt * 16 + kko * 4 + r;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float pv;
                        if (half == 0) pv = i == 0 ? ua[rt][r][j] + ub[rt][r][j] : (i == 2 ? ub[rt][r][j] : uc[rt][r][j]);
                        else pv = i == 0 ? ua[rt][r][j] : (i == 1 ? 2.0f * ub[rt][r][j] : (i == 2 ? 4.0f * ua[rt][r][j] : fmaf(8.0f, ub[rt][r][j], uc[rt][r][j])));
                        ex[((half * 4 + j) * VR4 + R) * 64 + wcol] = pv;
                    }
                }
            __syncthreads();
            if (ooff >= 0 && 4 * otr + i < BX_ELE) {
                const float* e = ex + oR * 64 + ((oquad * 4) ^ (((oR >> 2) & 3) << 4));
                float* ou = out + ooff + i * BX_AZI * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 p0 = *reinterpret_cast<const float4*>(e + (j * VR4) * 64);
                    const float4 p1 = *reinterpret_cast<const float4*>(e + ((4 + j) * VR4) * 64);
                    float4 y = make_float4((p0.x + p1.x) + b4.x, (p0.y + p1.y) + b4.y, (p0.z + p1.z) + b4.z, (p0.w + p1.w) + b4.w);
                    if (RELU) y = make_float4(y.x > 0.f ? y.x : 0.f, y.y > 0.f ? y.y : 0.f, y.z > 0.f ? y.z : 0.f, y.w > 0.f ? y.w : 0.f);
                    __builtin_nontemporal_store((f32x4){y.x, y.y, y.z, y.w}, reinterpret_cast<f32x4*>(ou + j * 16));
                }
            }
            __syncthreads();
        }
        ug = ugn;
        if (ug >= ngroups) break;
    }
}

template <int NCHUNK, int COUT, bool RELU>
int launch_wino43(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, int units, float* out)
{
    if (L.nchunk != NCHUNK || L.cout != COUT || (L.relu != 0) != RELU || !L.Wwino43) {
        bx_set_error("winograd F(4x4) layer %d: geometry mismatch (%d chunks, %d channels)", layer, L.nchunk, L.cout);
        return BX_ERR_STATE;
    }
    auto k = wino43_kernel<NCHUNK, COUT, RELU>;
    int& cap = c->wino_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W43_LDS));
        cap = c->n_cu / (COUT / CW);
        if (cap < 1) cap = 1;
        if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
    }
    int grid = (units + G4 - 1) / G4;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k, dim3(grid, COUT / CW), dim3(CT), W43_LDS, s, in, units, L.Wwino43, L.b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

void g6(const double g[3], double o[6])       // the expressions of oracle/bx_oracle.c::wino43_g
{
    o[0] = g[0] / 4.0;
    o[1] = -((g[0] + g[1]) + g[2]) / 6.0;
    o[2] = -((g[0] - g[1]) + g[2]) / 6.0;
    o[3] = ((g[0] / 4.0 + g[1] / 2.0) + g[2]) / 6.0;
    o[4] = ((g[0] / 4.0 - g[1] / 2.0) + g[2]) / 6.0;
    o[5] = g[2];
}
}  // namespace

// U = G g G^T (F(4x4, 3x3)) of every (chunk, channel, output channel) in binary64, rounded once, packed as B fragments
// [chunk * 36 + plane][column tile][lane = kk*16 + li][4], element i = U[plane][chunk][kk + 4 i][col]
int bxk_wino43_weights(const float* w /* [nchunk][9][16][cout] */, int nchunk, int cout, float** d_out)
{
    const int nt = cout / 16;
    std::vector<float> frag((size_t)nchunk * NPL * nt * 64 * 4, 0.0f);
    for (int cc = 0; cc < nchunk; ++cc)
        for (int ch = 0; ch < 16; ++ch)
            for (int o = 0; o < cout; ++o) {
                double g[3][3], Gg[6][3];
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw) g[kh][kw] = (double)w[(((size_t)cc * 9 + kh * 3 + kw) * 16 + ch) * cout + o];
                for (int kw = 0; kw < 3; ++kw) {
                    const double col[3] = {g[0][kw], g[1][kw], g[2][kw]};
                    double r6[6];
                    g6(col, r6);
                    for (int xi = 0; xi < 6; ++xi) Gg[xi][kw] = r6[xi];
                }
                for (int xi = 0; xi < 6; ++xi) {
                    double u6[6];
                    g6(Gg[xi], u6);
                    for (int nu = 0; nu < 6; ++nu) {
                        const int pl = xi * 6 + nu, kk = ch & 3, i = ch >> 2, t = o / 16, li = o % 16;
                        frag[((((size_t)(cc * NPL + pl) * nt + t) * 4 + kk) * 16 + li) * 4 + i] = (float)u6[nu];
                    }
                }
            }
    BX_HIP(hipMalloc(reinterpret_cast<void**>(d_out), frag.size() * sizeof(float)));
    BX_HIP(hipMemcpy(*d_out, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice));
    return BX_OK;
}

// layer of Cylindrical_Net in the F(4x4, 3x3) form; -1 when this layer / unit count is not served (caller falls back)
int bxk_wino43(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (units_dev || max_units < 1) return -1;
    const ConvLayerDev& L = c->desc[layer];
    switch (layer) {
        case 0: return launch_wino43<3, 64, true>(c, layer, s, L, in, max_units, out);
        case 1: return launch_wino43<4, 64, true>(c, layer, s, L, in, max_units, out);
        case 2: return launch_wino43<4, 128, true>(c, layer, s, L, in, max_units, out);
        case 3: return launch_wino43<8, 128, true>(c, layer, s, L, in, max_units, out);
        case 4: return launch_wino43<8, 64, true>(c, layer, s, L, in, max_units, out);
        case 5: return launch_wino43<4, 64, true>(c, layer, s, L, in, max_units, out);
    }
    return -1;
}
