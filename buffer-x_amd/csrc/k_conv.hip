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
// Workgroup = 8 waves, G units: the input slab of one chunk is staged in LDS (80-byte rows, double-buffered
// against the next chunk's global loads) WITH its halo, so padding / wrap-around are plain rows and a tap is a
// constant row offset; wave (wm, wn) owns output tiles {wm + t*WM} x 16 channels with fp32 accumulators in VGPRs,
// weights stream from L2 as B fragments (one dword per lane per MFMA, reused across the wave's tiles).
#include "bx_common.h"
#include <cstdlib>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWF = 20;   // floats per LDS row (16 + 4 pad)

// Geometry is table-driven (ConvLayerDev, built on the host per layer) so that the MFMA loop carries NO address
// arithmetic beyond one add per tile:
//   lrow [P_IN]   LDS row (inside one unit's P_LDS-row slab) that receives input position p; lrow2 [P_IN] a second
//                 copy or -1.  Cylindrical maps are staged with a halo -- (7+2) x (20+2) rows: the azimuth wrap-around
//                 columns are copies, the elevation padding rows are zeros -- so that EVERY tap of EVERY output
//                 position is "window origin + constant": no per-tap lookup, no padding special cases.
//   obase[P_OUT]  LDS row of the window origin of output position pos;   toff [NTAPS] row offset of a tap.
// Measured on this chip (tools/ubench/mfma_mix.hip): every non-MFMA instruction in the loop costs matrix-pipe time,
// and the 4 MFMAs of a tile are fastest as a back-to-back dependent chain (153 TF pure, 124 TF when the compiler
// interleaves accumulators); hence one ds_read_b128 + one v_add per tile and a pinned tile-major order.
template <int NCHUNK, int NTAPS, int P_IN, int P_LDS, int P_OUT, int COUT, int G, bool RELU, int NW, int NPW>
struct ConvCfg {
    static constexpr int CT = NW * 64;    // threads
    static constexpr int M = G * P_OUT;
    static constexpr int MT = (M + 15) / 16;
    static constexpr int NT = (COUT + 15) / 16;
    static constexpr int WN = NT / NPW;       // NPW column tiles per wave: one A read feeds 4*NPW MFMAs
    static constexpr int WM = NW / WN;
    static constexpr int TPW = (MT + WM - 1) / WM;
    static constexpr int ROWS = G * P_LDS;
    static constexpr int BUF_FLOATS = ROWS * ROWF;
    static constexpr int NLD = (G * P_IN * 4 + CT - 1) / CT;
    static constexpr int EPI_FLOATS = 16 * ROWF;              // per-wave 16-row x 16-channel transposition scratch (epilogue)
    // the scratch normally has its own LDS; the 972-row CostNet slab leaves no room, there it aliases the slab buffer
    // that has just been consumed (and a barrier follows the epilogue)
    static constexpr bool EPI_ALIAS = (size_t)2 * BUF_FLOATS * 4 + (size_t)NW * EPI_FLOATS * 4 > 160 * 1024;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF_FLOATS * 4 + (EPI_ALIAS ? 0 : (size_t)NW * EPI_FLOATS * 4);
    // taps whose B fragments are all requested before the next slab's loads go out (see the tail of the tap loop)
    static constexpr int KT = NTAPS - 1 < 2 ? NTAPS - 1 : (TPW > 10 ? 1 : 2);
    // register budget: two co-resident workgroups (4 waves/SIMD, <= 128 VGPRs) when LDS allows two
    // register budget: as many co-resident workgroups as the LDS allows (they desynchronise and cover each other's
    // prologue / barrier / epilogue phases), capped where the accumulator tile would spill
    static constexpr int WG_LDS = (int)(160 * 1024 / LDS_BYTES);
    static constexpr int WPS_WANT = WG_LDS * NW / 4;                       // waves per SIMD the LDS would admit
    static constexpr int WPS_CAP = TPW * NPW <= 5 ? 6 : (TPW * NPW <= 10 ? (NW == 4 ? 5 : 4) : 2);
    static constexpr int MINW = WPS_WANT < WPS_CAP ? (WPS_WANT < 1 ? 1 : WPS_WANT) : WPS_CAP;
    // cylindrical 3x3 layers: tap offsets are compile-time constants -> the tap loop is unrolled and the offset rides in the
    // ds_read's immediate field (no address add per tile)
    static constexpr bool CYLG = NTAPS == 9 && P_IN == BX_EA && P_LDS == (BX_ELE + 2) * (BX_AZI + 2);
    // un-padded 3x3 layers on a square VW x VW map (CostNet's k(3,1,3) layers): same treatment, row stride VW
    static constexpr int isq(int v) { int r = 0; while ((r + 1) * (r + 1) <= v) ++r; return r; }
    static constexpr int VW = isq(P_IN);
    static constexpr bool VALG = NTAPS == 9 && P_LDS == P_IN && VW * VW == P_IN && (VW - 2) * (VW - 2) == P_OUT;
    static constexpr bool ST9 = CYLG || VALG;                  // static 9-tap schedule
    static constexpr int TAPW = CYLG ? BX_AZI + 2 : VW;        // LDS rows between two tap rows
    static_assert(NT % NPW == 0 && WN <= NW && NW % WN == 0, "waves must tile the output channels");
    static_assert(LDS_BYTES <= 160 * 1024, "slab double buffer exceeds the LDS");
    static_assert(NTAPS <= 64, "tap offsets live in one lane each");
};

template <int NCHUNK, int NTAPS, int P_IN, int P_LDS, int P_OUT, int COUT, int G, bool RELU, int NW, int NPW>
__global__ __launch_bounds__(NW * 64, (ConvCfg<NCHUNK, NTAPS, P_IN, P_LDS, P_OUT, COUT, G, RELU, NW, NPW>::MINW)) void conv_kernel(
    const float* __restrict__ in, const int32_t* __restrict__ units_dev, int max_units, const float* __restrict__ W,
    const float* __restrict__ bias, const int32_t* __restrict__ lrow, const int32_t* __restrict__ lrow2,
    const int32_t* __restrict__ obase, const int32_t* __restrict__ toff, float* __restrict__ out,
    const int32_t* __restrict__ skip, long long* __restrict__ dbg)
{
    if (skip && *skip) return;
    using C = ConvCfg<NCHUNK, NTAPS, P_IN, P_LDS, P_OUT, COUT, G, RELU, NW, NPW>;
    constexpr int CT = C::CT;
    // optional cycle stamps (BX_BALL_DEBUG): 16 workgroups x {t0, prologue, [taps done, barrier passed] per chunk, end}
    long long t0_ = 0;
    const bool tr_ = dbg != nullptr && (blockIdx.x % 151) == 0 && blockIdx.x / 151 < 16 && threadIdx.x == 0;
    long long* td_ = dbg + (blockIdx.x / 151) * 32;
    if (tr_) { t0_ = __builtin_readcyclecounter(); td_[0] = t0_; }
#define CV_TR(k) do { if (tr_) td_[k] = __builtin_readcyclecounter() - t0_; } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* buf = reinterpret_cast<float*>(smem);

    int units = max_units;
    if (units_dev) { int u = *units_dev; units = u < max_units ? u : max_units; }
    const int ngroups = (units + G - 1) / G;
    if ((int)blockIdx.x >= ngroups) return;
    int grp = blockIdx.x;            // persistent walk: groups blockIdx.x, blockIdx.x + gridDim.x, ...
    int u0 = grp * G;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % C::WN, wm = wave / C::WN;
    const int wm_u = __builtin_amdgcn_readfirstlane(wm);     // SGPR copy: branches on it are scalar
    const int li = lane & 15, kk = lane >> 4;

    // ---- both slab buffers start as zeros (the halo rows stay zero for the whole kernel)
    for (int i = tid; i < 2 * C::BUF_FLOATS / 4; i += CT) reinterpret_cast<float4*>(buf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- staging geometry of this thread's float4 pieces (constant over the chunks)
    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4 st[C::NLD];
    // piece geometry (LDS destinations, source row) is fixed for the whole kernel: keep it in registers -- recomputing it
    // per chunk put two dependent table loads in front of every LDS hand-off
    int dst1[C::NLD], dst2[C::NLD];
#pragma unroll
    for (int q = 0; q < C::NLD; ++q) {
        const int f = tid + q * CT;
        const int row = f >> 2, part = f & 3;
        const int g = row / P_IN, p = row - g * P_IN;
        dst1[q] = -1; dst2[q] = -1;
        if (row < G * P_IN) {
            dst1[q] = (g * P_LDS + lrow[p]) * ROWF + part * 4;
            const int r2 = lrow2[p];
            if (r2 >= 0) dst2[q] = (g * P_LDS + r2) * ROWF + part * 4;
        }
    }
    auto gload = [&](int cc, int ub) {
        int tq = tid;
        asm volatile("" : "+v"(tq));      // source addresses are recomputed (pure integer math, no table loads): fewer live registers
#pragma unroll
        for (int q = 0; q < C::NLD; ++q) {
            const int f = tq + q * CT;
            const int row = f >> 2, part = f & 3;
            const int g = row / P_IN, p = row - g * P_IN;
            st[q] = (row < G * P_IN && ub + g < units)
                        ? in4[((size_t)(ub + g) * NCHUNK + cc) * P_IN * 4 + (size_t)p * 4 + part]
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lwrite = [&](int b) {
        float* d = buf + (size_t)b * C::BUF_FLOATS;
#pragma unroll
        for (int q = 0; q < C::NLD; ++q) {
            if (dst1[q] >= 0) *reinterpret_cast<float4*>(d + dst1[q]) = st[q];
            if (dst2[q] >= 0) *reinterpret_cast<float4*>(d + dst2[q]) = st[q];
        }
    };

    // ---- A-operand byte offset of every tile row owned by this lane (window origin), tap offsets one per lane
    int abase[C::TPW];
#pragma unroll
    for (int t = 0; t < C::TPW; ++t) {
        const int m = (wm + t * C::WM) * 16 + li;
        int r = 0;                                       // rows beyond M compute garbage that is never stored
        if (m < C::M) { const int g = m / P_OUT, pos = m - g * P_OUT; r = g * P_LDS + obase[pos]; }
        abase[t] = (r * ROWF + kk * 4) * 4;
    }
    const int toffv = lane < NTAPS ? toff[lane] * (ROWF * 4) : 0;

    gload(0, u0);
    __syncthreads();          // zero fill complete before the first slab lands on top of it
    lwrite(0);
    __syncthreads();

    // ---- a wave owns NPW column tiles of 16 channels; accumulators start at the (BN-folded) bias
    const int n0 = wn * NPW * 16;
    bool colok[NPW];
    float bvs[NPW];
    f32x4 acc[C::TPW][NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        colok[j] = (n0 + j * 16 + li) < COUT;
        bvs[j] = colok[j] ? bias[n0 + j * 16 + li] : 0.0f;
    }

    // B fragments: device layout [chunk*tap][column tile][lane][4] (bx_load_weights) -> ONE 16-byte load per lane and tap
    const float4* wbase = reinterpret_cast<const float4*>(W) + (size_t)(wn * NPW) * 64 + lane;
    auto loadB = [&](int ct, float (&b)[NPW][4]) {
#ifdef BX_EXP_NOB
        if (ct != 0) return;
#endif
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const float4 v = wbase[((size_t)ct * C::NT + j) * 64];
            b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
        }
    };
    float bc[NPW][4], bn[NPW][4];
    loadB(0, bc);
    if constexpr (C::ST9) loadB(1, bn);
    else {
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) bn[j][i] = 0.f;
    }
    // drain the prologue loads here: otherwise the waitcnt pass sees them pending on the loop's entry edge and
    // waits for the NEWEST loads (the counters are in-order) in front of every tap's first MFMAs
    __builtin_amdgcn_s_waitcnt(0);
    CV_TR(1);

    const int slot = 4 * (li & 3) + (li >> 2);
    int sl = 0;                      // running slab counter: LDS buffer = sl & 1
  for (;;) {
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
        for (int t = 0; t < C::TPW; ++t) acc[t][j] = (f32x4){bvs[j], bvs[j], bvs[j], bvs[j]};
    const int grp_next = grp + (int)gridDim.x;
#pragma unroll 1
    for (int cc = 0; cc < NCHUNK; ++cc, ++sl) {
        const char* lb = reinterpret_cast<const char*>(buf + (size_t)(sl & 1) * C::BUF_FLOATS);
        // one tap: per tile ONE address add + ONE ds_read_b128 (issued two tiles ahead) + 4 back-to-back MFMAs
        // per-chunk tile base = slab buffer + window origin; a tap adds a constant (immediate field of the ds_read when the
        // geometry is compile-time, one v_add otherwise)
        const char* ab[C::TPW];
#pragma unroll
        for (int t = 0; t < C::TPW; ++t) ab[t] = lb + abase[t];
        auto do_tap = [&](int tp, const float (&b_)[NPW][4]) {
            const int tb = C::ST9 ? ((tp / 3) * C::TAPW + tp % 3) * (ROWF * 4) : __builtin_amdgcn_readlane(toffv, tp);
            f32x4 a[C::TPW];
            a[0] = *reinterpret_cast<const f32x4*>(ab[0] + tb);
            if (C::TPW > 1) a[C::TPW > 1 ? 1 : 0] = *reinterpret_cast<const f32x4*>(ab[C::TPW > 1 ? 1 : 0] + tb);
#pragma unroll
            for (int t = 0; t < C::TPW; ++t) {
                if (t + 2 < C::TPW) a[t + 2 < C::TPW ? t + 2 : 0] = *reinterpret_cast<const f32x4*>(ab[t + 2 < C::TPW ? t + 2 : 0] + tb);
                __builtin_amdgcn_sched_barrier(0);
                // the last tile slot of a wave may lie beyond the map (MT not a multiple of WM: the 32-channel layers waste 2 of 20
                // slots): a wave-uniform branch skips its MFMAs, the matrix pipe goes to the co-resident waves instead
                if (C::MT % C::WM == 0 || t + 1 < C::TPW || wm_u + t * C::WM < C::MT) {
#pragma unroll
                    for (int j = 0; j < NPW; ++j) {
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b_[j][0], acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b_[j][1], acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b_[j][2], acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b_[j][3], acc[t][j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (C::ST9) {
            // Unrolled 9-tap schedule, B fragments requested TWO taps ahead (one tap of MFMAs is not always longer than an
            // L2 round trip under load); the fragments of the next chunk's taps 0 and 1 and then the next slab go out at
            // tap 6, so nothing issued after the slab load is waited for before the LDS hand-off (vmcnt retires in order).
            float b2[NPW][4], b3[NPW][4], b4[NPW][4], b5[NPW][4], b6[NPW][4], b7[NPW][4], b8[NPW][4], n0b[NPW][4], n1b[NPW][4];
            const int cb = cc * NTAPS;
            loadB(cb + 2, b2); do_tap(0, bc);
            loadB(cb + 3, b3); do_tap(1, bn);
            loadB(cb + 4, b4); do_tap(2, b2);
            loadB(cb + 5, b5); do_tap(3, b3);
            loadB(cb + 6, b6); do_tap(4, b4);
            loadB(cb + 7, b7); do_tap(5, b5);
            loadB(cb + 8, b8);
            {
                int c0 = cb + 9, c1 = cb + 10;
                if (c0 >= NCHUNK * NTAPS) c0 -= NCHUNK * NTAPS;       // first taps of the next group's chunk 0
                if (c1 >= NCHUNK * NTAPS) c1 -= NCHUNK * NTAPS;
                loadB(c0, n0b); loadB(c1, n1b);
            }
#ifndef BX_EXP_NOGLOAD
            if (cc + 1 < NCHUNK) gload(cc + 1, u0);
            else if (grp_next < ngroups) gload(0, grp_next * G);
#endif
            do_tap(6, b6);
            do_tap(7, b7);
#ifndef BX_EXP_NOGLOAD
            if (cc + 1 < NCHUNK || grp_next < ngroups) lwrite((sl + 1) & 1);
#endif
            do_tap(8, b8);
#pragma unroll
            for (int j = 0; j < NPW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) { bc[j][i] = n0b[j][i]; bn[j][i] = n1b[j][i]; }
        } else {
        // head taps [0, T0): the B fragment of the next tap is fetched while the current one is on the matrix cores
        constexpr int KT = C::KT, T0 = NTAPS - 1 - KT;
#pragma unroll 1
        for (int tp = 0; tp < T0; ++tp) {
            loadB(cc * NTAPS + tp + 1, bn);
            do_tap(tp, bc);
#pragma unroll
            for (int j = 0; j < NPW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) bc[j][i] = bn[j][i];
        }
        // tail taps [T0, NTAPS): vmcnt retires IN ORDER, so a B fetch issued after the slab load would wait for the
        // slab's HBM round trip.  Every remaining B fragment of this chunk AND the first one of the next chunk are
        // requested first, THEN the next slab's loads go out; nothing waits on them until the LDS hand-off below.
        {
            float bt[KT + 1][NPW][4];
#pragma unroll
            for (int i = 0; i <= KT; ++i) {
                int ctn = cc * NTAPS + T0 + 1 + i;
                if (ctn >= NCHUNK * NTAPS) ctn -= NCHUNK * NTAPS;   // harmless extra fetch after the last chunk
                loadB(ctn, bt[i]);
            }
            // next slab: the next chunk of this group, or chunk 0 of this workgroup's NEXT group (its prologue disappears)
#ifndef BX_EXP_NOGLOAD
            if (cc + 1 < NCHUNK) gload(cc + 1, u0);
            else if (grp_next < ngroups) gload(0, grp_next * G);
#endif
            do_tap(T0, bc);
#pragma unroll
            for (int i = 0; i + 1 < KT; ++i) do_tap(T0 + 1 + i, bt[i]);
            // hand the next slab to LDS BEFORE the last tap: the other buffer has been free since the previous barrier, so
            // the LDS writes overlap the other waves' MFMAs and the barrier below only waits for the last tap
#ifndef BX_EXP_NOGLOAD
            if (cc + 1 < NCHUNK || grp_next < ngroups) lwrite((sl + 1) & 1);
#endif
            if (KT > 0) do_tap(T0 + KT, bt[KT > 0 ? KT - 1 : 0]);
#pragma unroll
            for (int j = 0; j < NPW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) bc[j][i] = bt[KT][j][i];
        }
        }
        CV_TR(2 + 2 * cc);
#ifndef BX_EXP_NOGLOAD
        __syncthreads();
#endif
        CV_TR(3 + 2 * cc);
    }

    // ---- epilogue of this group: ReLU, store in chunk-slot order (the next group's first slab is already in LDS).
    //      A tile (16 rows x 16 channels) is 1 KiB contiguous in the output map; the accumulator layout (lane = channel,
    //      4 rows per lane) is turned into row-major through a wave-private LDS scratch so that every lane issues ONE
    //      16-byte store (the 4-byte scattered stores of the first version cost 3 % of the stack).
    {
        float* scr = (C::EPI_ALIAS ? buf + (size_t)((sl - 1) & 1) * C::BUF_FLOATS : buf + 2 * C::BUF_FLOATS) + wave * C::EPI_FLOATS;
        int lrow16 = lane >> 2;
        asm volatile("" : "+v"(lrow16));   // keeps the per-tile store addresses out of the loop-invariant hoisting (VGPR budget)
        const int lpart = lane & 3;
#pragma unroll
        for (int t = 0; t < C::TPW; ++t) {
            const int mt = wm + t * C::WM;
            if (mt >= C::MT) continue;
            const int m = mt * 16 + lrow16;
            const int g = m / P_OUT, pos = m - g * P_OUT;
            const bool ok = m < C::M && u0 + g < units;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[t][j][r];
                    if (RELU) v = v > 0.0f ? v : 0.0f;
                    scr[(kk * 4 + r) * ROWF + slot] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float4 v4 = *reinterpret_cast<const float4*>(scr + lrow16 * ROWF + lpart * 4);
                if (ok)
                    *reinterpret_cast<float4*>(out + (((size_t)(u0 + g) * C::NT + wn * NPW + j) * P_OUT + pos) * 16 + lpart * 4) = v4;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (C::EPI_ALIAS) __syncthreads();
    }
    CV_TR(2 + 2 * NCHUNK);
    grp = grp_next;
    if (grp >= ngroups) break;
    u0 = grp * G;
  }
}

// ---------------------------------------------------------------- CostNet layer 0 on the implicit cost volume
// cost[c][n][k][l] = S[c][k+1][(l-n) mod 20] - T[c][k+1][l]   (models/BUFFERX.py:59-65), valid 3x3x3 conv 32->32.
constexpr int CV_D = BX_AZI, CV_H = BX_ELE - 2, CV_W = BX_AZI;       // 20 x 5 x 20
constexpr int CV_DO = CV_D - 2, CV_HO = CV_H - 2, CV_WO = CV_W - 2;  // 18 x 3 x 18
constexpr int CV_POUT = CV_DO * CV_HO * CV_WO;                        // 972
constexpr int CT = 512;                                              // threads of the cost-volume kernel
constexpr int CV_ROWF = 36;                                           // 32 + 4 pad floats per LDS row

// One workgroup per match.  S (source equivariant map, elevation rows 1..5) is staged with a +-2 column wrap-around
// halo ([5][24] rows) and T as [5][20] rows, both channels-last in chunk-slot order.  With e = (l - n) mod 20 per output
// row, the A operand of tap (a,b,c) is  S_ext[k+b][e + 2 + c - a] - T[k+b][l+c]:  "row base + constant", so each tile
// costs two ds_read_b128 (immediate offsets), four v_sub and -- a wave owns both 16-channel column tiles -- EIGHT MFMAs.
constexpr int CV_WE = CV_W + 4;                                       // 24 columns of the wrapped S map

__global__ __launch_bounds__(CT, 2) void cost_l1_kernel(const float* __restrict__ s_equi, const float* __restrict__ t_equi,
                                                        const int32_t* __restrict__ s_mids, const int32_t* __restrict__ t_mids,
                                                        const int32_t* __restrict__ m_dev, int max_m, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out, const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    constexpr int MT = (CV_POUT + 15) / 16;  // 61
    constexpr int WM = 8, TPW = (MT + WM - 1) / WM;   // 8 row tiles per wave, both column tiles
    constexpr int NTAPS = 27;
    constexpr int ROWB = CV_ROWF * 4;                 // bytes per LDS row (32 channels + pad)
    __shared__ __attribute__((aligned(16))) float sS[CV_H * CV_WE * CV_ROWF];
    __shared__ __attribute__((aligned(16))) float sT[CV_H * CV_W * CV_ROWF];

    int m = *m_dev;
    m = m < max_m ? m : max_m;
    const int u = blockIdx.x;
    if (u >= m) return;
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int wm_u = __builtin_amdgcn_readfirstlane(wm);
    const int li = lane & 15, kk = lane >> 4;

    const float* sp = s_equi + ((size_t)s_mids[u] * BX_EA + BX_AZI) * 32;  // elevation rows 1..5
    const float* tp_ = t_equi + ((size_t)t_mids[u] * BX_EA + BX_AZI) * 32;
    for (int f = tid; f < CV_H * CV_W * 32; f += CT) {
        const int row = f >> 5, c = f & 31;
        const int k = row / CV_W, l = row - k * CV_W;
        const int sl = (c & 16) + 4 * (c & 3) + ((c & 15) >> 2);
        const float sv = sp[f];
        sS[(k * CV_WE + l + 2) * CV_ROWF + sl] = sv;
        if (l >= CV_W - 2) sS[(k * CV_WE + l + 2 - CV_W) * CV_ROWF + sl] = sv;      // columns -2, -1
        if (l < 2) sS[(k * CV_WE + l + 2 + CV_W) * CV_ROWF + sl] = sv;              // columns 20, 21
        sT[row * CV_ROWF + sl] = tp_[f];
    }
    __syncthreads();

    f32x4 acc[TPW][2];
    int bS[TPW], bT[TPW];
    {
        const float bv0 = bias[li], bv1 = bias[16 + li];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            acc[t][0] = (f32x4){bv0, bv0, bv0, bv0};
            acc[t][1] = (f32x4){bv1, bv1, bv1, bv1};
            int mrow = (wm + t * WM) * 16 + li;
            if (mrow >= CV_POUT) mrow = CV_POUT - 1;
            const int n = mrow / (CV_HO * CV_WO), rem = mrow - n * (CV_HO * CV_WO);
            const int k = rem / CV_WO, l = rem - k * CV_WO;
            int e = l - n;
            e = e < 0 ? e + CV_W : e;
            bS[t] = ((k * CV_WE + e + 2) * CV_ROWF + kk * 4) * 4;
            bT[t] = ((k * CV_W + l) * CV_ROWF + kk * 4) * 4;
        }
    }
    const char* cS = reinterpret_cast<const char*>(sS);
    const char* cT = reinterpret_cast<const char*>(sT);
    const float4* w4 = reinterpret_cast<const float4*>(W) + lane;
    float4 bq0 = w4[0], bq1 = w4[64];
    // Software pipeline over (tap, tile): the S / T rows of the NEXT tile are requested before the 8 MFMAs of the current one, so
    // the wave never sits on an LDS round trip between two MFMA groups (one workgroup = 2 waves per SIMD is all the registers
    // admit: a wave has to cover its own latencies).  Round 1 read the rows right in front of their MFMAs, into the
    // accumulator's scratch registers (s_nop 7 + lgkmcnt(0) per tile): 73.5 % matrix-pipe occupancy.
    f32x4 psv = *reinterpret_cast<const f32x4*>(cS + bS[0]);
    f32x4 ptv = *reinterpret_cast<const f32x4*>(cT + bT[0]);
#pragma unroll 1
    for (int cc = 0; cc < 2; ++cc) {
#pragma unroll 1
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int bc = 0; bc < 9; ++bc) {
                const int b = bc / 3, c = bc % 3;
                const int tp = a * 9 + bc;
                // B fragments of the next tap (wraps harmlessly after the last one)
                int nx = cc * NTAPS + tp + 1;
                nx = nx < 2 * NTAPS ? nx : 0;
                const float4 nq0 = w4[(size_t)nx * 128], nq1 = w4[(size_t)nx * 128 + 64];
                const int offS = (b * CV_WE + c) * ROWB + cc * 64;
                const int offT = (b * CV_W + c) * ROWB + cc * 64;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const f32x4 av = psv - ptv;
                    // operands of the tile after this one: next tile of the tap, or tile 0 of the next tap (next a: the S column
                    // moves one to the left; next chunk: back to the first column, channels 16..31)
                    if (t + 1 < TPW) {
                        psv = *reinterpret_cast<const f32x4*>(cS + bS[t + 1 < TPW ? t + 1 : 0] + offS);
                        ptv = *reinterpret_cast<const f32x4*>(cT + bT[t + 1 < TPW ? t + 1 : 0] + offT);
                    } else if (bc < 8) {
                        constexpr int dummy = 0; (void)dummy;
                        const int nb = (bc + 1) / 3, nc = (bc + 1) % 3;
                        psv = *reinterpret_cast<const f32x4*>(cS + bS[0] + (nb * CV_WE + nc) * ROWB + cc * 64);
                        ptv = *reinterpret_cast<const f32x4*>(cT + bT[0] + (nb * CV_W + nc) * ROWB + cc * 64);
                    } else {
                        const int ncc = a < 2 ? cc : (cc + 1) & 1;                 // after the last tap: wraps to the start (unused)
                        const int sh = a < 2 ? -ROWB : 2 * ROWB;
                        psv = *reinterpret_cast<const f32x4*>(cS + bS[0] + sh + ncc * 64);
                        ptv = *reinterpret_cast<const f32x4*>(cT + bT[0] + ncc * 64);
                    }
                    if (t + 1 == TPW && wm_u + t * WM >= MT) continue;   // tile slot beyond the 61 row tiles (waves 5..7): scalar branch
                    __builtin_amdgcn_sched_barrier(0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bq0.x, acc[t][0], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bq0.y, acc[t][0], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bq0.z, acc[t][0], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bq0.w, acc[t][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);       // two dependent chains back to back, not interleaved (ubench: 153 vs 124 TF)
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bq1.x, acc[t][1], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bq1.y, acc[t][1], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bq1.z, acc[t][1], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bq1.w, acc[t][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                bq0 = nq0; bq1 = nq1;
            }
            // next a: the S column moves one to the left
#pragma unroll
            for (int t = 0; t < TPW; ++t) bS[t] -= ROWB;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) bS[t] += 3 * ROWB;
    }
    const int slot = 4 * (li & 3) + (li >> 2);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int mt = wm + t * WM;
        if (mt >= MT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mrow = mt * 16 + kk * 4 + r;
            if (mrow < CV_POUT) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = acc[t][j][r];
                    v = v > 0.0f ? v : 0.0f;
                    out[(((size_t)u * 2 + j) * CV_POUT + mrow) * 16 + slot] = v;
                }
            }
        }
    }
}

template <int NCHUNK, int NTAPS, int P_IN, int P_LDS, int P_OUT, int COUT, int G, bool RELU, int NW = 8, int NPW = 1>
int launch_conv(bx_ctx* c, int net, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, const int32_t* units_dev, int max_units,
                float* out, const int32_t* skip, long long* dbg = nullptr)
{
    using C = ConvCfg<NCHUNK, NTAPS, P_IN, P_LDS, P_OUT, COUT, G, RELU, NW, NPW>;
    constexpr int CT = C::CT;
    if (L.nchunk != NCHUNK || L.ntaps != NTAPS || L.p_in != P_IN || L.p_lds != P_LDS || L.p_out != P_OUT || L.cout != COUT ||
        (L.relu != 0) != RELU) {
        bx_set_error("conv layer geometry mismatch (%d %d %d %d %d %d)", L.nchunk, L.ntaps, L.p_in, L.p_lds, L.p_out, L.cout);
        return BX_ERR_STATE;
    }
    auto k = conv_kernel<NCHUNK, NTAPS, P_IN, P_LDS, P_OUT, COUT, G, RELU, NW, NPW>;
    // Function attribute and occupancy are per DEVICE: they are kept in the context (one context = one device), not in
    // process-wide statics, so contexts on several devices of one process each set them up for their own device.
    int& cap = c->conv_cap[net][layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        // persistent workgroups: as many as are co-resident, each walking its unit groups (BX_CONV_PERSIST=0: one group each;
        // BX_CONV_PERSIST_CAP=n: at most n workgroups -- test hook that forces the group walk at small unit counts)
        cap = 1 << 30;
        if (c->conv_persist) {
            int occ = 0;
            BX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, CT, C::LDS_BYTES));
            if (occ >= 1) cap = occ * c->n_cu;
            if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
        }
    }
    int grid = (max_units + G - 1) / G;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k, dim3(grid), dim3(CT), C::LDS_BYTES, s, in, units_dev, max_units, L.W, L.b, L.lrow, L.lrow2, L.obase, L.toff,
                       out, skip, dbg);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
}  // namespace

// units per workgroup of the smallest CostNet layers (6: 8x8 -> 6x6, 7: 6x6 -> 4x4, 8: 4x4 -> 2x2, 9: 2x2 -> 1).  Round 6: 16 / 32 / 128 -> 8 / 8 / 16
// for layers 7..9: more, smaller workgroups (27.5 / 11.9 / 9.1 -> 17.7 / 7.8 / 6.0 us per launch at 1 400 matches, rocprofv3)
#ifndef BX_TAIL_G6
#define BX_TAIL_G6 8
#endif
#ifndef BX_TAIL_G7
#define BX_TAIL_G7 8
#define BX_TAIL_G8 8
#define BX_TAIL_G9 16
#endif

int bxk_conv(bx_ctx* c, hipStream_t s, int net, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    constexpr int CYL = (BX_ELE + 2) * (BX_AZI + 2);   // 198 LDS rows per unit: cylindrical map + halo
    if (max_units < 1) return BX_OK;                   // an empty launch is not an error in any form (bx_desc_net(K = 0), bx_conv_layer(units = 0))
    if (net == 0 && c->use_wino == 3) {
        const int rcm = bxk_wino43m(c, s, layer, in, units_dev, max_units, out);                          // mixed F(4x4) / F(3x4) tiles: every layer
        if (rcm >= 0) return rcm;
    }
    if (net == 0 && c->use_wino == 2) {
        const int rc43 = bxk_wino43(c, s, layer, in, units_dev, max_units, out);                          // F(4x4, 3x3): every layer
        if (rc43 >= 0) return rc43;
        // -1 = "not served" (a device-side unit count): the direct kernels below take it
    }
    if (net == 0 && c->use_wino == 1) {
        const int rcw = bxk_wino(c, s, layer, in, units_dev, max_units, out);                             // F(2x2, 3x3): layers 0..5
        if (rcw >= 0) return rcw;
    }
    if (net == 0) {
        const ConvLayerDev& L = c->desc[layer];
        // Configuration by measurement (tools/gpu_conv.sh, K = 5000): 8 waves; 2 units per workgroup for the 64- and
        // 32-channel layers (9 / 5 accumulator tiles per wave), 1 unit for the 128-channel layers.  Variants tried and
        // found within +-2 %: 4-wave workgroups, two column tiles per wave (NPW = 2), a persistent group walk,
        // pinned accumulator interleaving (10 % slower).  Re-measured on the final kernel (127.4 TFLOP/s stand-alone) for the two
        // 128-channel layers: NPW = 2 with 8 waves / G = 2 (127.5) and with 4 waves (127.6) -- halving the A-operand LDS reads
        // changes nothing, so the LDS read rate is not what holds the stack at 0.81 of the f32 MFMA peak.
        switch (layer) {
            //                     NCHUNK taps P_IN P_LDS P_OUT COUT G  RELU
            case 0: return launch_conv<3, 9, 140, CYL, 140, 64, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 1: return launch_conv<4, 9, 140, CYL, 140, 64, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip,
                                                                          getenv("BX_BALL_DEBUG") ? c->ball_dbg : nullptr);
            case 2: return launch_conv<4, 9, 140, CYL, 140, 128, 1, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 3: return launch_conv<8, 9, 140, CYL, 140, 128, 1, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 4: return launch_conv<8, 9, 140, CYL, 140, 64, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 5: return launch_conv<4, 9, 140, CYL, 140, 64, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 6: return launch_conv<4, 9, 140, CYL, 140, 32, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 7: return launch_conv<2, 9, 140, CYL, 140, 32, 2, false>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
        }
    } else if (net == 1) {
        if (c->use_wino_pose == 2) {
            const int rcw = bxk_wino43v(c, s, layer, in, units_dev, max_units, out);      // layers 1..5: valid F(4x4, 3x3)
            if (rcw >= 0) return rcw;
        } else if (c->use_wino_pose == 1) {
            const int rcw = bxk_wino_pose(c, s, layer, in, units_dev, max_units, out);    // layers 1..5: valid F(2x2, 3x3)
            if (rcw >= 0) return rcw;
        }
        const ConvLayerDev& L = c->pose[layer];
        switch (layer) {
            case 1: return launch_conv<2, 27, 972, 972, 256, 64, 1, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 2: return launch_conv<4, 9, 256, 256, 196, 64, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 3: return launch_conv<4, 9, 196, 196, 144, 128, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 4: return launch_conv<8, 9, 144, 144, 100, 128, 2, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 5: return launch_conv<8, 9, 100, 100, 64, 64, 4, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 6: return launch_conv<4, 9, 64, 64, 36, 64, BX_TAIL_G6, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 7: return launch_conv<4, 9, 36, 36, 16, 32, BX_TAIL_G7, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 8: return launch_conv<2, 9, 16, 16, 4, 32, BX_TAIL_G8, true>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
            case 9: return launch_conv<2, 4, 4, 4, 1, 20, BX_TAIL_G9, false>(c, net, layer, s, L, in, units_dev, max_units, out, c->skip);
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
