// tools/experimental/k_wino43h.hip -- NOT part of the library (round-4 experiment, kept for the record; profiles/r04_wino43_variants.txt).
// Bit-exact against bxo_conv_wino43 (the 24 conv tests of tests/test_gpu_stages.py pass with it), two workgroups per CU confirmed by
// hipOccupancyMaxActiveBlocksPerMultiprocessor -- and 12-15 % SLOWER than k_wino43.hip (layers 0..5 at K = 5000: 2 450-2 500 us against
// 2 145-2 170 us).  Two measurements say why: (1) with every weight fragment read from one L1-resident kilobyte (wrong results, timing
// only) it runs exactly as fast as k_wino43.hip (2 158 vs 2 162 us) -- one MFMA row tile per fragment doubles the fragment stream from
// L2 to ~19 TB/s at the target speed, and that stream costs 12 %; (2) a second resident workgroup adds NOTHING (256 or 512 workgroups:
// same time): VALU and LDS instructions of one wave do not run under the MFMAs of the other wave of the SIMD, they take issue time
// from the matrix pipe, so transform / output time is paid in full whichever workgroup spends it.  To build it: add it to SRCS,
// declare bxk_wino43h and bx_ctx::wino43h_cap[BX_NDESC] in bx_common.h and call it from bxk_wino43 for layers 0..5.
//
// k_wino43h.hip -- the 64-column layers of Cylindrical_Net in the Winograd F(4x4, 3x3) form, TWO workgroups per CU.
//
// Same arithmetic as k_wino43.hip (contract oracle/bx_oracle.c::bxo_conv_wino43, GPU == oracle bit for bit; reference
// models/patchnet.py:49-84, padding utils/common.py:265-310), other decomposition.  k_wino43.hip holds 32 tile rows x 36 planes of one
// 16-channel chunk in LDS (145 KB: one workgroup per CU) and its eight waves walk transform -> MFMA -> output in lock step, so the
// matrix pipe idles while they transform (mfma_busy 0.50).  Here a workgroup item is three BANDS -- a band = one tile row of a unit =
// five 4 x 4 output tiles, 15 of the 16 rows of ONE MFMA row tile -- so slab (3 x 6 input rows) + V planes (36 x 16 rows) are 78 KB and
// two workgroups share a CU; they drift apart and one's transform / output runs under the other's MFMAs.
//  * workgroup = 4 waves (one per SIMD), wave = one 16-column tile of the 64, ALL 36 planes: 144 accumulator VGPRs, and the whole output
//    transform A^T M A is lane-local (no exchange between waves, no LDS, no barrier): nu pass of the six xi rows (144 -> 96 live values),
//    then the xi pass column by column, 16 stores of 16 bytes per lane and item.
//  * one MFMA row tile per weight fragment (k_wino43.hip: two), so the B-fragment stream from L2 doubles per output; it stays a raw
//    buffer load ring (four planes) with wave-uniform offsets.  Two planes alternate on the matrix pipe (dependent MFMAs on one
//    accumulator issue every 40 cycles, independent ones every 32).
//  * slab: bands alternate between the upper (input rows -1 .. 4) and the lower (3 .. 8) half of a unit; with an EVEN number of
//    workgroups along x a workgroup's slot i always holds the same half, so the rows beyond the map are zeroed once and the piece table
//    (16-byte pieces, requested a chunk ahead, written / re-requested inside the MFMA loop) is per-thread constant.  The two bands of a
//    unit read two input rows twice: 13.5 instead of 10.5 rows per three bands.
//  * transform: thread (tile row, channel) as in k_wino43.hip (256 threads = 16 x 16).
#include "wino43_common.h"
#include <cstdio>
#include <cstdlib>

namespace {
using namespace w43;
constexpr int WP = BX_AZI + 2;                   // slab columns (wrap-around halo)
constexpr int TC4 = BX_AZI / 4;                  // tiles of a band
constexpr int GB = 3, ROWSH = GB * TC4;          // three bands = 15 tile rows of the 16
constexpr int RP3 = WP * ROWF + 4;               // slab row pitch in floats (444)
constexpr int BP = 6 * RP3 + 8;                  // band pitch: (BP - 4 * 4 * ROWF) % 32 == 16, the tile rows either side of a band boundary sit 16 banks apart
constexpr int VPLH = 16 * ROWF;                  // floats per V plane
constexpr int CTH = 256;
constexpr int NPU = BX_EA * 4;                   // 16-byte pieces of one (unit, chunk) map
constexpr int NLDH = 5;                          // pieces per thread: 14 rows x 80 pieces / 256 threads
constexpr size_t W43H_LDS = (size_t)(GB * BP + NPL * VPLH) * 4;   // 32 064 + 46 080 B
static_assert(W43H_LDS <= 80 * 1024 && (BP * 4) % 16 == 0 && (BP - 16 * ROWF) % 32 == 16 && 14 * BX_AZI * 4 <= NLDH * CTH && NPL % 4 == 0 && BX_ELE == 7,
              "two workgroups per CU, 16-byte slab rows, bank spread, piece table, B ring, 2 bands per unit");

// ---- output transform of one wave's 16 x 16 accumulator tile x 36 planes, lane-local.  Accumulator register r of lane (li, kk) is
// M[plane][tile row li][slot 4 kk + r].  Expression order = wino43_send / wino43_finish of wino43_common.h (the contract).
template <int NT, bool RELU>
__device__ __forceinline__ void wino43h_output(const f32x4 (&acc)[NPL], int lane, int ug, int units, int ctile, const float4 b4, float* __restrict__ out)
{
    const int li = lane & 15, kk = lane >> 4;
    float rr[6][4][4];                                      // [xi][j][r]: nu pass
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m0 = acc[x * 6 + 0][r], m1 = acc[x * 6 + 1][r], m2 = acc[x * 6 + 2][r], m3 = acc[x * 6 + 3][r],
                        m4 = acc[x * 6 + 4][r], m5 = acc[x * 6 + 5][r];
            const float p = m1 + m2, q = m1 - m2, s = m3 + m4, t = m3 - m4;
            rr[x][0][r] = (m0 + p) + s;
            rr[x][1][r] = fmaf(2.0f, t, q);
            rr[x][2][r] = fmaf(4.0f, s, p);
            rr[x][3][r] = fmaf(8.0f, t, q) + m5;
        }
    const int slot = li / TC4, tc = li - slot * TC4;
    const int gb = ug * GB + slot, u = gb >> 1, b = gb & 1;
    const bool live = li < ROWSH && u < units;
    float* ou = out + ((size_t)(u * NT + ctile) * BX_EA + (4 * b) * BX_AZI + 4 * tc) * 16 + 4 * kk;
    const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float y[4][4];                                      // [output row of the tile][r]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float r0 = rr[0][j][r], r1 = rr[1][j][r], r2 = rr[2][j][r], r3 = rr[3][j][r], r4 = rr[4][j][r], r5 = rr[5][j][r];
            const float p12 = r1 + r2, q12 = r1 - r2, p34 = r3 + r4, q34 = r3 - r4;
            y[0][r] = ((r0 + p12) + p34) + ba[r];
            y[1][r] = (q12 + 2.0f * q34) + ba[r];
            y[2][r] = (p12 + 4.0f * p34) + ba[r];
            y[3][r] = (q12 + fmaf(8.0f, q34, r5)) + ba[r];
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i][r] = y[i][r] > 0.f ? y[i][r] : 0.f;
            }
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < 3 || b == 0)                        // the 8th output row does not exist
                    __builtin_nontemporal_store((f32x4){y[i][0], y[i][1], y[i][2], y[i][3]}, reinterpret_cast<f32x4*>(ou + (i * BX_AZI + j) * 16));
        }
    }
}

template <int NCHUNK, int COUT, bool RELU>
__global__ __launch_bounds__(CTH, 2) void wino43h_kernel(const float* __restrict__ in, int units, const float* __restrict__ U,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    constexpr int NT = COUT / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    float* Vp = slab + GB * BP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctg = (int)blockIdx.y * 4 + wave;
    const int li = lane & 15, kk = lane >> 4;
    const int ngroups = (2 * units + GB - 1) / GB;
    if ((int)blockIdx.x >= ngroups) return;
    const int par = (int)blockIdx.x & 1;            // gridDim.x is even (or 1): slot i of this workgroup always holds a band of half (par + i) & 1

    for (int i = tid; i < (int)(W43H_LDS / 16); i += CTH) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- slab traffic: per-thread constants.  Slot i holds input rows h0 .. h0 + nrow - 1 of its unit at band rows rr0 ..:
    // upper half (b = 0): h = 0 .. 4 at band rows 1 .. 5 (band row 0 = the zero row above the map);
    // lower half (b = 1): h = 3 .. 6 at band rows 0 .. 3 (band rows 4, 5 = the zero rows below it)
    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4 st[NLDH];
    int lpc[NLDH];                                  // piece inside the (unit, chunk) map (10 bits) | slot << 10 | halo code << 12 | slab offset / 4 << 14; -1 = none
#pragma unroll
    for (int q = 0; q < NLDH; ++q) {
        int f = tid + q * CTH;
        lpc[q] = -1;
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            const int b = (par + i) & 1, nrow = b ? 4 : 5, np = nrow * BX_AZI * 4;
            if (f >= 0 && f < np) {
                const int ri = f / (BX_AZI * 4), rem = f - ri * (BX_AZI * 4), w = rem >> 2, part = rem & 3;
                const int h = b ? 3 + ri : ri, rrow = b ? ri : ri + 1;
                const int halo = w == 0 ? 1 : (w == BX_AZI - 1 ? 2 : 0);
                const int dst = i * BP + rrow * RP3 + (w + 1) * ROWF + part * 4;
                lpc[q] = ((h * BX_AZI + w) * 4 + part) | (i << 10) | (halo << 12) | ((dst >> 2) << 14);
                f = -1;
            } else if (f >= 0) f -= np;
        }
    }
    auto gload1 = [&](int q, int ug_, int cc_) {   // streamed once: non-temporal
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (lpc[q] >= 0) {
            const int u = (ug_ * GB + ((lpc[q] >> 10) & 3)) >> 1;
            if (u < units) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in4 + ((size_t)u * NCHUNK + cc_) * NPU + (lpc[q] & 1023)));
        }
        st[q] = make_float4(v.x, v.y, v.z, v.w);
    };
    auto lwrite1 = [&](int q) {
        if (lpc[q] >= 0) {
            float* d = slab + (lpc[q] >> 14) * 4;
            *reinterpret_cast<float4*>(d) = st[q];
            const int halo = (lpc[q] >> 12) & 3;
            if (halo != 0) *reinterpret_cast<float4*>(d + (halo == 1 ? BX_AZI * ROWF : -BX_AZI * ROWF)) = st[q];
        }
    };

    // ---- transform role: (tile row tR = tid / 16, channel slot tid % 16); the 16 lanes of the padding row idle
    const int tR = tid >> 4;
    const bool tact = tR < ROWSH;
    const int tRc = tact ? tR : ROWSH - 1;
    const int tsl = tRc / TC4, ttc = tRc - tsl * TC4;
    const float* wsrc = slab + tsl * BP + (4 * ttc) * ROWF + (tid & 15);
    float* vdst = Vp + tRc * ROWF + (tid & 15);
    auto transform = [&]() {
        if (!tact) return;
        float t[6][6];                              // t[xi][j]: B^T d down column j
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float o[6];
            bt6s(wsrc[j * ROWF], wsrc[RP3 + j * ROWF], wsrc[2 * RP3 + j * ROWF], wsrc[3 * RP3 + j * ROWF], wsrc[4 * RP3 + j * ROWF], wsrc[5 * RP3 + j * ROWF], o);
#pragma unroll
            for (int x = 0; x < 6; ++x) t[x][j] = o[x];
        }
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            float o[6];
            bt6s(t[x][0], t[x][1], t[x][2], t[x][3], t[x][4], t[x][5], o);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) vdst[(x * 6 + nu) * VPLH] = o[nu];
        }
    };

    const float* bq = bias + ctg * 16 + kk;         // slots 4 kk .. 4 kk + 3 hold the logical channels kk, 4 + kk, 8 + kk, 12 + kk (read at output time)
    // weight fragments [chunk * 36 + plane][column tile][lane][4] (bxk_wino43_weights): raw buffer loads, wave-uniform offsets
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, NCHUNK * NPL * NT * 1024, 0x00020000);
    const int ubase = ctg * 1024;
    const int ulane = lane * 16;
    auto bload = [&](int q) {
#ifdef BX_W43_WFAKE       // timing experiment: every fragment load hits the same L1-resident kilobyte (results are wrong)
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, (q & 1) * 1024, 0));
#else
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, ubase + q * (NT * 1024), 0));
#endif
        return make_float4(v.x, v.y, v.z, v.w);
    };
    const char* abase = reinterpret_cast<const char*>(Vp) + (li * ROWF + kk * 4) * 4;

    f32x4 acc[NPL];
    // weight fragments in a ring of FOUR planes: the two slots a pair of planes has just used are refilled behind its MFMAs with the
    // planes four ahead (two pair-iterations of slack)
    float4 bring[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) bring[p] = bload(p);

    int ug = blockIdx.x;
    const int gstep = (int)gridDim.x;
    int lg = ug, lc = 0;                            // the (group, chunk) the NEXT request fetches
    auto ladv = [&]() { if (++lc == NCHUNK) { lc = 0; lg += gstep; } };
#pragma unroll
    for (int q = 0; q < NLDH; ++q) gload1(q, lg, lc);
    ladv();
    __syncthreads();                 // zero fill complete
#pragma unroll
    for (int q = 0; q < NLDH; ++q) lwrite1(q);
    bool st_live = lg < ngroups;
    if (st_live) {
#pragma unroll
        for (int q = 0; q < NLDH; ++q) gload1(q, lg, lc);
        ladv();
    }

    for (;;) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ugn = ug + gstep;
#pragma unroll 1
        for (int cc = 0; cc < NCHUNK; ++cc) {
            __syncthreads();         // the slab of this chunk is complete; every wave is done with the V planes of the chunk before
            transform();
            __syncthreads();         // V complete; the slab is free
            const bool st_was = st_live;
            st_live = lg < ngroups;
            const int lgq = lg, lcq = lc;
            if (st_live) ladv();
            const int cn = cc + 1 == NCHUNK ? 0 : cc + 1;
            f32x4 ar0 = *reinterpret_cast<const f32x4*>(abase);
            f32x4 ar1 = *reinterpret_cast<const f32x4*>(abase + VPLH * 4);
#pragma unroll
            for (int p = 0; p < NPL; p += 2) {
                // the slab is free during the MFMA phase: piece q goes to the slab (requested a whole chunk ago) and is re-requested
                if (p / 2 < NLDH) {
                    if (st_was) lwrite1(p / 2);
                    if (st_live) gload1(p / 2, lgq, lcq);
                }
                const f32x4 a0 = ar0, a1 = ar1;
                if (p + 2 < NPL) {
                    ar0 = *reinterpret_cast<const f32x4*>(abase + ((p + 2) * VPLH) * 4);
                    ar1 = *reinterpret_cast<const f32x4*>(abase + ((p + 3) * VPLH) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[p % 4].x, a0.x, acc[p], 0, 0, 0);
                acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[(p + 1) % 4].x, a1.x, acc[p + 1], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[p % 4].y, a0.y, acc[p], 0, 0, 0);
                acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[(p + 1) % 4].y, a1.y, acc[p + 1], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[p % 4].z, a0.z, acc[p], 0, 0, 0);
                acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[(p + 1) % 4].z, a1.z, acc[p + 1], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[p % 4].w, a0.w, acc[p], 0, 0, 0);
                acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bring[(p + 1) % 4].w, a1.w, acc[p + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                bring[p % 4] = p + 4 < NPL ? bload(cc * NPL + p + 4) : bload(cn * NPL + p + 4 - NPL);
                bring[(p + 1) % 4] = p + 5 < NPL ? bload(cc * NPL + p + 5) : bload(cn * NPL + p + 5 - NPL);
            }
        }
        wino43h_output<NT, RELU>(acc, lane, ug, units, ctg, make_float4(bq[0], bq[4], bq[8], bq[12]), out);
        ug = ugn;
        if (ug >= ngroups) break;
    }
}

template <int NCHUNK, int COUT, bool RELU>
int launch_wino43h(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, int units, float* out)
{
    if (L.nchunk != NCHUNK || L.cout != COUT || (L.relu != 0) != RELU || !L.Wwino43) {
        bx_set_error("winograd F(4x4) layer %d: geometry mismatch (%d chunks, %d channels)", layer, L.nchunk, L.cout);
        return BX_ERR_STATE;
    }
    auto k = wino43h_kernel<NCHUNK, COUT, RELU>;
    int& cap = c->wino43h_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W43H_LDS));
        cap = 2 * c->n_cu / (COUT / 64);
        if (c->conv_cap_override > 0 && 2 * c->conv_cap_override < cap) cap = 2 * c->conv_cap_override;
        if (cap < 2) cap = 2;
        cap &= ~1;
        if (getenv("BX_W43H_OCC")) {
            int nb = -1;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), CTH, W43H_LDS);
            fprintf(stderr, "wino43h layer %d: occupancy %d workgroups per CU (err %d), LDS %zu, cap %d\n", layer, nb, (int)e, W43H_LDS, cap);
        }
    }
    int grid = (2 * units + GB - 1) / GB;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    if (grid > 1) grid &= ~1;           // even: a workgroup's slots keep their unit half over its whole walk
    hipLaunchKernelGGL(k, dim3(grid, COUT / 64), dim3(CTH), W43H_LDS, s, in, units, L.Wwino43, L.b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
}  // namespace

// the 64-column layers (0 .. 5) in the two-workgroups-per-CU decomposition; -1: not served (k_wino43.hip's kernel takes the layer)
int bxk_wino43h(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (units_dev || max_units < 1) return -1;
    const ConvLayerDev& L = c->desc[layer];
    switch (layer) {
        case 0: return launch_wino43h<3, 64, true>(c, layer, s, L, in, max_units, out);
        case 1: return launch_wino43h<4, 64, true>(c, layer, s, L, in, max_units, out);
        case 2: return launch_wino43h<4, 128, true>(c, layer, s, L, in, max_units, out);
        case 3: return launch_wino43h<8, 128, true>(c, layer, s, L, in, max_units, out);
        case 4: return launch_wino43h<8, 64, true>(c, layer, s, L, in, max_units, out);
        case 5: return launch_wino43h<4, 64, true>(c, layer, s, L, in, max_units, out);
    }
    return -1;
}
