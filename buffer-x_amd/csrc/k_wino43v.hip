// k_wino43v.hip -- CostNet layers 1..5 as VALID Winograd F(4x4, 3x3) convolutions on the f32 matrix cores (round 4;
// bx_params.pose_conv_form = BX_POSE_CONV_WINOGRAD43).
//
// Reference: models/patchnet.py:197-201 -- Conv3d k(3,3,3) on [18][3][18], then k(3,1,3) on [16][1][16] ... [10][1][10], un-padded: 2-D valid
// 3 x 3 convolutions over the two azimuth-like axes (n, l); layer 1 folds its three k rows into the channel dimension (6 effective
// 16-channel chunks e = chunk * 3 + k).  A D x D map gives (D - 2)^2 outputs = T x T output tiles of 4 x 4, T = ceil((D - 2) / 4); the 6 x 6
// window of tile (tr, tc) starts at (4 tr, 4 tc), rows / columns >= D read as zero (D = 16, 12), outputs >= D - 2 are dropped: 36
// multiplications per 16 outputs against 16 per 4 of the F(2x2, 3x3) form of k_wino.hip (0.56x the MFMA work; 0.73x / 0.81x for D = 16 / 12).
// Arithmetic contract: oracle/bx_oracle.c::bxo_conv_wino43_valid (the transforms and the accumulation order of bxo_conv_wino43);
// GPU == oracle bit for bit.
//
// Kernel = the round-4 structure of k_wino43.hip with valid geometry: workgroup = 8 waves = 64 output channels of G units (G x T^2 <= 32
// tile rows = two MFMA row tiles: G = 2, 2, 3, 3, 8 for D = 18 .. 10), the unit count is the device-side match count; the slab
// (G x (4 T + 2)^2 positions, rows beyond D stay zero) is fed by 16-byte pieces requested a whole chunk ahead and written inside the
// MFMA loop; one-channel transform (36 ds_read_b32, shared column pass, 36 ds_write_b32); swapped MFMA operands, buffer-load B ring,
// two-round output exchange with 16-byte stores (the last tile column / row of D = 16, 12 stores its existing outputs only).
#include "wino43_common.h"
#include <cstdlib>

namespace {
using namespace w43;

template <int NE, int FOLD, int COUT, int D, int G, bool RELU>
struct GeoV {
    static constexpr int NCH = NE / FOLD;                    // real 16-channel chunks of the input map
    static constexpr int DO = D - 2, T = (DO + 3) / 4, NTU = T * T, ROWS = G * NTU;
    static constexpr int SD = 4 * T + 2;                     // slab rows = columns (the window of the last tile ends at 4 (T - 1) + 5)
    static constexpr int RP = SD * ROWF + 4, UP = SD * RP;   // row / unit pitch in floats (4 rows = 16 banks mod 32: SD is even)
    static constexpr int PIN = D * D, PIN3 = D * FOLD * D;   // positions of an effective chunk / of a real chunk
    static constexpr int NPU = PIN * 4, NPIECE = G * NPU, NLD = (NPIECE + CT - 1) / CT;
    static constexpr size_t LDS = (size_t)(G * UP + NPL * VPL4 + 64) * 4;   // slab | V planes | the workgroup's 64 bias values
    static_assert(NE % FOLD == 0 && ROWS <= VR4 && SD >= D && SD % 2 == 0 && LDS <= 160 * 1024 && 2 * NLD + 1 <= NPH && (RP * 4) % 16 == 0,
                  "tile rows fit two MFMA row tiles, slab covers the map, LDS, slab traffic fits the plane loop");
};

template <int NE, int FOLD, int COUT, int D, int G, bool RELU>
__global__ __launch_bounds__(CT, 2) void wino43v_kernel(const float* __restrict__ in, const int32_t* __restrict__ units_dev, int max_units,
                                                        const float* __restrict__ U, const float* __restrict__ bias, float* __restrict__ out,
                                                        const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    using GE = GeoV<NE, FOLD, COUT, D, G, RELU>;
    constexpr int NT = COUT / 16, NCW = 4, NLD = GE::NLD, RP = GE::RP, UP = GE::UP, T = GE::T, NTU = GE::NTU, DO = GE::DO;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    float* Vp = slab + G * UP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, ctl = wave >> 1;
    const int ctg = (int)blockIdx.y * NCW + ctl;
    const int li = lane & 15, kk = lane >> 4;
    int units = max_units;
    if (units_dev) { const int u = *units_dev; units = u < max_units ? u : max_units; }
    const int ngroups = (units + G - 1) / G;
    if ((int)blockIdx.x >= ngroups) return;

    for (int i = tid; i < (int)(GE::LDS / 16); i += CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- slab traffic: piece f = (unit g of the group, position p = n D + l, 16-byte part): source inside the group's
    //      [G][NCH][D FOLD D][16] floats (position (n FOLD + k) D + l of real chunk c2 for the effective chunk e = c2 FOLD + k), destination
    float4 st[NLD];
    int lsrc[NLD], ldst[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int f = tid + q * CT;
        lsrc[q] = -1; ldst[q] = 0;
        if (f < GE::NPIECE) {
            const int g = f / GE::NPU, fr = f - g * GE::NPU;
            const int p = fr >> 2, part = fr & 3;
            const int n = p / D, l = p - n * D;
            lsrc[q] = (g * GE::NCH * GE::PIN3 + n * FOLD * D + l) * 4 + part;
            ldst[q] = g * UP + n * RP + l * ROWF + part * 4;
        }
    }
    // raw buffer loads (round 5): lane offset lsrc * 16 (a lane constant) + the (group, chunk) offset in an SGPR -- no 64-bit VALU address per
    // piece and chunk, no address registers; pieces of units that do not exist read as zeros (explicit predicate: the hardware range
    // check does not see the SGPR offset)
    const long long in_bytes = (long long)max_units * GE::NCH * GE::PIN3 * 64;
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)(in_bytes < 0x7fffffffLL ? in_bytes : 0x7fffffffLL), 0x00020000);
    auto gload1 = [&](int q, int ug_, int e_) {
        const int c2 = e_ / FOLD, k = e_ - c2 * FOLD;
        const int soff = ((ug_ * G * GE::NCH + c2) * GE::PIN3 + k * D) * 64;
        const int lim = (units - ug_ * G) * GE::NCH * GE::PIN3 * 4;     // pieces of units that do not exist read as zeros
        const f32x4 v = (lsrc[q] >= 0 && lsrc[q] < lim) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, lsrc[q] * 16, soff, 2 /* nt */)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        st[q] = make_float4(v.x, v.y, v.z, v.w);
    };
    auto lwrite1 = [&](int q) {
        if (lsrc[q] >= 0) *reinterpret_cast<float4*>(slab + ldst[q]) = st[q];
    };

    // ---- transform role: (tile row tR = 4 wave + lane / 16, channel slot lane % 16)
    const int tR = 4 * wave + (lane >> 4);
    const bool tact = tR < GE::ROWS;
    const int tRc = tact ? tR : GE::ROWS - 1;
    const int tg = tRc / NTU, tt = tRc - tg * NTU, ttr = tt / T, ttc = tt - ttr * T;
    const float* wsrc = slab + tg * UP + (4 * ttr) * RP + (4 * ttc) * ROWF + (lane & 15);
    float* vdst = Vp + tRc * ROWF + (lane & 15);
    auto transform = [&]() {
        if (!tact) return;
        float t[6][6];                              // t[xi][j]: B^T d down column j
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float o[6];
            bt6s(wsrc[j * ROWF], wsrc[RP + j * ROWF], wsrc[2 * RP + j * ROWF], wsrc[3 * RP + j * ROWF], wsrc[4 * RP + j * ROWF], wsrc[5 * RP + j * ROWF], o);
#pragma unroll
            for (int x = 0; x < 6; ++x) t[x][j] = o[x];
        }
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            float o[6];
            bt6s(t[x][0], t[x][1], t[x][2], t[x][3], t[x][4], t[x][5], o);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) vdst[(x * 6 + nu) * VPL4] = o[nu];
        }
    };

    // (the bias of the workgroup's column tiles lives in LDS as [column tile][kk][r] = slot kk + 4 r, behind the V planes: output phase)
    const long long out_bytes = (long long)max_units * NT * (DO * DO) * 64;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(out_bytes < 0x7fffffffLL ? out_bytes : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, NE * NPL * NT * 1024, 0x00020000);
    const int ubase = ((half * NPH) * NT + ctg) * 1024;
    const int ulane = lane * 16;
    auto bload = [&](int q) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, ubase + q * (NT * 1024), 0));
        return make_float4(v.x, v.y, v.z, v.w);
    };
    const char* abase = reinterpret_cast<const char*>(Vp) + ((half * NPH * VR4 + li) * ROWF + kk * 4) * 4;

    f32x4 acc[NPH][RT4];
    float4 bring[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bring[p] = bload(p);

    int ug = blockIdx.x;
    const int gstep = (int)gridDim.x;
    int lg = ug, lc = 0;                            // the (group, effective chunk) the NEXT request fetches
    auto ladv = [&]() { if (++lc == NE) { lc = 0; lg += gstep; } };
#pragma unroll
    for (int q = 0; q < NLD; ++q) gload1(q, lg, lc);
    ladv();
    __syncthreads();                 // zero fill complete
    if (tid < 64) Vp[NPL * VPL4 + (tid >> 4) * 16 + (tid & 3) * 4 + ((tid & 15) >> 2)] = bias[(int)blockIdx.y * 64 + tid];
#pragma unroll
    for (int q = 0; q < NLD; ++q) lwrite1(q);
    bool st_live = lg < ngroups;

    for (;;) {
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int rt = 0; rt < RT4; ++rt) acc[p][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ugn = ug + gstep;
#pragma unroll 1
        for (int cc = 0; cc < NE; ++cc) {
            __syncthreads();         // the slab of this chunk is complete; every wave is done with the V planes of the chunk before
            // the pieces of the NEXT chunk are requested here, a transform and a whole MFMA phase before they are written to the slab (loads and
            // the weight ring share one in-order counter: a request between ring loads holds the plane three ahead until it returns)
            st_live = lg < ngroups;
            if (st_live) {
#pragma unroll
                for (int q = 0; q < NLD; ++q) gload1(q, lg, lc);
                ladv();
            }
            __builtin_amdgcn_sched_barrier(0);
            transform();
            __syncthreads();         // V complete; the slab is free
            const int cn = cc + 1 == NE ? 0 : cc + 1;
            f32x4 ar[3];
            ar[0] = *reinterpret_cast<const f32x4*>(abase);
            ar[1] = *reinterpret_cast<const f32x4*>(abase + (16 * ROWF) * 4);
#pragma unroll
            for (int p = 0; p < NPH; ++p) {
                const float4 bqq = bring[p % 3];
                bring[p % 3] = p + 3 < NPH ? bload(cc * NPL + p + 3) : bload(cn * NPL + p + 3 - NPH);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < RT4; ++rt) {
                    const int s0 = p * RT4 + rt, s2 = s0 + 2;
                    if (s2 < NPH * RT4) ar[s2 % 3] = *reinterpret_cast<const f32x4*>(abase + (((s2 / RT4) * VR4 + (s2 % RT4) * 16) * ROWF) * 4);
                    const f32x4 a = ar[s0 % 3];
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.x, a.x, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.y, a.y, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.z, a.z, acc[p][rt], 0, 0, 0);
                    acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.w, a.w, acc[p][rt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the slab writes of the next chunk, all at once behind the last plane (one wait for the loads in flight instead of one per piece)
            if (st_live) {
#pragma unroll
                for (int q = 0; q < NLD; ++q) lwrite1(q);
            }
        }
        __syncthreads();             // every wave is done with the V planes: their bytes carry the exchange now
        // the lane constants of the output phase (exchange addresses, tile coordinates, bias) are derived here, every group: the empty asm
        // keeps hipcc from hoisting them out of the group loop, where they lived -- spilled -- across the MFMA pipeline (k_wino43.hip)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int li = lane_o & 15, kk = lane_o >> 4;
        float4* ex = reinterpret_cast<float4*>(Vp);         // output exchange [wave][8][lane]
        float4* mine = ex + (wave * 8) * 64 + lane_o;
        const float4* theirs = ex + ((wave ^ 1) * 8) * 64 + lane_o;
        const float4 b4 = *reinterpret_cast<const float4*>(Vp + NPL * VPL4 + ctl * 16 + kk * 4);
#pragma unroll
        for (int rt = 0; rt < RT4; ++rt) {
            f32x2 A[2][4], B[2][4];             // [register pair][j]
            if (half == 0) wino43_send<0>(acc, rt, A, B, mine);
            else wino43_send<1>(acc, rt, A, B, mine);
            __syncthreads();
            const int R = rt * 16 + li;
            const int g = R / NTU, t = R - g * NTU, tr = t / T, tc = t - tr * T;
            const int u = ug * G + g;
            const int i0 = 2 * half;                        // this half's output rows of a tile: i0, i0 + 1
            const bool live = R < GE::ROWS && u < units && 4 * tr + i0 < DO;
            const bool second_row = 4 * tr + i0 + 1 < DO;
            const int nj = DO - 4 * tc;                     // output columns of this tile that exist (>= 4: all)
            const unsigned jmask = nj >= 4 ? 15u : (1u << (nj > 0 ? nj : 0)) - 1u;
            const int voff = (((g * NT) * (DO * DO) + (4 * tr + i0) * DO + 4 * tc) * 16 + 4 * kk) * 4;
            const int soff = ((ug * G * NT + ctg) * (DO * DO) * 16) * 4;
            // raw buffer stores (32-bit offsets, no 64-bit address registers).  The group's offset is ADDED INTO the lane offset and the
            // instruction's SGPR offset field stays 0 -- see k_wino43.hip::wino43_output for the hazard this avoids
            const int vo = voff + soff;
            auto store = [&](int off, const f32x4 v) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, vo + off * 4, 0, 2 /* nt */);
            };
            if (half == 0) wino43_finish<0, RELU>(A, B, theirs, b4, store, live, second_row, DO * 16, jmask);
            else wino43_finish<1, RELU>(A, B, theirs, b4, store, live, second_row, DO * 16, jmask);
            __syncthreads();                                // the exchange is free again (next row tile / next group's V planes)
        }
        ug = ugn;
        if (ug >= ngroups) break;
    }
}

template <int NE, int FOLD, int COUT, int D, int G, bool RELU>
int launch_wino43v(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    using GE = GeoV<NE, FOLD, COUT, D, G, RELU>;
    if (L.nchunk * FOLD != NE || L.cout != COUT || (L.relu != 0) != RELU || !L.Wwino43 || L.ntaps != 9 * FOLD) {
        bx_set_error("winograd F(4x4) CostNet layer %d: geometry mismatch (%d chunks, %d taps, %d channels)", layer, L.nchunk, L.ntaps, L.cout);
        return BX_ERR_STATE;
    }
    // 32-bit byte offsets inside the kernel (raw buffer resources): 2 GiB or more of input or output maps (~17 k matches on layer 1) is
    // "not served" -- the direct kernels (64-bit addressing) take the layer
    if (!w43::fits_i32((long long)max_units * GE::NCH * GE::PIN3 * 64) || !w43::fits_i32((long long)max_units * (COUT / 16) * (GE::DO * GE::DO) * 64)) return -1;
    auto k = wino43v_kernel<NE, FOLD, COUT, D, G, RELU>;
    int& cap = c->wino43v_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GE::LDS));
        cap = c->n_cu / (COUT / 64);
        if (cap < 1) cap = 1;
        if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
    }
    int grid = (max_units + G - 1) / G;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k, dim3(grid, COUT / 64), dim3(CT), GE::LDS, s, in, units_dev, max_units, L.Wwino43, L.b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
}  // namespace

// CostNet layer 1..5 in the F(4x4, 3x3) form; -1 for the other layers (caller falls back to the direct kernels)
int bxk_wino43v(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (max_units < 1) return -1;
    const ConvLayerDev& L = c->pose[layer];
    switch (layer) {
        //                            NE FOLD COUT  D  G
        case 1: return launch_wino43v<6, 3, 64, 18, 2, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 2: return launch_wino43v<4, 1, 64, 16, 2, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 3: return launch_wino43v<4, 1, 128, 14, 3, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 4: return launch_wino43v<8, 1, 128, 12, 3, true>(c, layer, s, L, in, units_dev, max_units, out);
        case 5: return launch_wino43v<8, 1, 64, 10, 8, true>(c, layer, s, L, in, units_dev, max_units, out);
    }
    return -1;
}
