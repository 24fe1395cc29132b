// k_wino43m.hip -- Cylindrical_Net layers as MIXED-tile Winograd convolutions on the f32 matrix cores (bx_params.desc_conv_form = winograd43m).
//
// Reference: models/patchnet.py:49-84, padding utils/common.py:265-310 (as k_wino43.hip).  The all-F(4x4) form cuts the 7 map rows into
// two tile rows of 4 and multiplies an 8th output row that does not exist: 1/8 of the MFMA rows of tile row 1.  Here tile row 0 (output
// rows 0..3) is the F(4x4, 3x3) tile of k_wino43.hip, bit for bit, and tile row 1 (rows 4..6) is an F(3x4, 3x3) tile: a 5 x 6 window,
// 30 planes instead of 36, F(3, 3) on the points {1, -1, 1/2, -1/2, inf} down the rows (its fp32 error against a binary64 convolution
// is 0.94 x that of F(4, 3) on the same rows: tests/study_wino43m_error.py).  66 instead of 72 planes per column block = -8.3 % MFMAs.
// Arithmetic contract: oracle/bx_oracle.c::bxo_conv_wino43m; GPU == oracle bit for bit.
//
// Kernel = k_wino43.hip's (read that header first) with these differences:
//  * ITEM = 16 consecutive COLUMN BLOCKS (unit, tile column) of the layer: MFMA row tile 0 holds their F(4x4) tiles, row tile 1 their F(3x4)
//    tiles -- tile rows r and r + 16 of an item are the two tiles of one column block (same unit slot of the slab, same output columns).
//    16 blocks = 3.2 units: at most four unit slots, the slab of the other kernel.
//  * transform: waves 0..3 own the F(4x4) tiles (unchanged), waves 4..7 the F(3x4) tiles: 30 window reads (rows h = 3..7), B3^T down the
//    five rows (bt5s), B^T along the six columns, 30 planes.
//  * MFMA phase: the two row tiles of a plane multiply DIFFERENT filter transforms (U_A = G g G^T, U_B = G3 g G^T), so a plane step loads
//    two fragments (interleaved in memory: one address stream) and the F(3x4) row tile has no planes xi = 5: the wave half that owns
//    xi 3..5 of the F(4x4) tiles owns xi 3, 4 of the F(3x4) ones and skips the MFMAs of its last six plane steps on row tile 1
//    (120 instead of 144 MFMAs per chunk).  The two waves of a SIMD are (ct, half) and (ct + 2, half): the halves of the column tiles
//    2, 3 are SWAPPED (he = half ^ (ct >> 1)) so that every SIMD carries one 144- and one 120-MFMA wave.
//  * output: row tile 0 as k_wino43.hip; row tile 1 through wino43m_send / wino43m_finish (half 0 stores the map rows 4, 5, half 1 row 6).
// MEASURED (profiles/r06_wino43m.txt; K = 5000, us per stack, one box, alternating): 2 286-2 312 against 2 331-2 338 for k_wino43.hip = -1.5 %,
// not the -6 % the MFMA count promises: without the half swap (every SIMD at 288 MFMAs, as the other kernel) this kernel is +1.7 % --
// the second fragment stream (34 instead of 17 loads inside a chunk's MFMA stream) and the packed slab constants cost that much -- and
// the swap recovers 3.7 %, less than 8.3 % because the 144-MFMA wave finishes its last 24 MFMAs alone on the SIMD (dependent chains of
// four at the 40-cycle latency).  NOT the default form: a second arithmetic for 1.5 % is not worth it; kept as a tested alternative.
#include "wino43_common.h"
#include <cstdlib>
#include <type_traits>
#include <vector>

#ifndef BX_W43M_RING
#define BX_W43M_RING 2
#endif

namespace {
using namespace w43;
constexpr int WP = BX_AZI + 2;                   // slab columns (wrap-around halo)
constexpr int HP = BX_ELE + 3;                   // slab rows h = -1 .. 8
constexpr int TC4 = BX_AZI / 4;                  // 5 column blocks per unit
constexpr int G4 = 4, BLK = 16;                  // unit slots of the slab; column blocks of an item
constexpr int HPU = HP - 1;                      // rows a slot owns (h = -1 .. 7)
constexpr int RP3 = WP * ROWF + 4, UP3 = HPU * RP3;
constexpr int SLABF = G4 * UP3 + RP3;
constexpr size_t W43M_LDS = (size_t)(SLABF + NPL * VPL4 + 64) * 4;
static_assert(NPH % 3 == 0 && BX_AZI % 4 == 0 && BX_ELE == 7 && W43M_LDS <= 160 * 1024 && (RP3 * 4) % 16 == 0 && 8 * 8 * 64 * 16 <= NPL * VPL4 * 4,
              "geometry, LDS, 16-byte slab rows, output exchange inside the V planes");

// ---- output transform: row tile 0 = the F(4x4) tiles (wino43_send / wino43_finish), row tile 1 = the F(3x4) tiles
template <int NT, bool RELU, int he>
__device__ __forceinline__ void wino43m_output(const f32x4 (&acc)[NPH][RT4], float* Vp, bool cw, int wave, int lane, int u0, int ioff, int units,
                                               int ctile, const __amdgpu_buffer_rsrc_t ors)
{
    float4* ex = reinterpret_cast<float4*>(Vp);             // [wave][8][lane]
    asm volatile("" : "+v"(lane));                          // (lane constants re-derived per group: see k_wino43.hip)
    const float4 b4 = *reinterpret_cast<const float4*>(Vp + NPL * VPL4 + (wave >> 1) * 16 + (lane >> 4) * 4);
    float4* mine = ex + (wave * 8) * 64 + lane;
    const float4* theirs = ex + ((wave ^ 1) * 8) * 64 + lane;
#pragma unroll
    for (int rt = 0; rt < RT4; ++rt) {
        f32x2 A[2][4], B[2][4];                 // [register pair][j]
        if (cw) {
            if (rt == 0) { if (he == 0) wino43_send<0>(acc, rt, A, B, mine); else wino43_send<1>(acc, rt, A, B, mine); }
            else { if (he == 0) wino43m_send<0>(acc, rt, A, B, mine); else wino43m_send<1>(acc, rt, A, B, mine); }
        }
        __syncthreads();
        if (cw) {
            // the tile coordinates are the same for both row tiles (tile rows r and r + 16 are one column block), but they are derived
            // AGAIN behind each exchange barrier from an opaque copy of the lane id: computed once they are live -- in fact spilled --
            // across the send of the other row tile and hipcc loses the whole register schedule of the kernel (250+ spilled VGPRs)
            int lq = lane;
            asm volatile("" : "+v"(lq));
            const int blk = ioff + (lq & 15);                       // column block inside the item's unit window
            const int g = blk / TC4, tc = blk - g * TC4;
            const int u = u0 + g;
            const bool live = u < units;
            const int kk = lq >> 4;
            // first output row of this half in this tile: F(4x4) tile rows 2 he, 2 he + 1; F(3x4) tile: half 0 the map rows 4, 5, half 1 row 6
            const int row0 = rt == 0 ? 2 * he : 4 + 2 * he;
            // (raw buffer stores with soffset = 0: the store-data hazard note of k_wino43.hip)
            const int voff = (((u * NT + ctile) * BX_EA + row0 * BX_AZI + 4 * tc) * 16 + 4 * kk) * 4;
            auto store = [&](int off, const f32x4 v) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, voff + off * 4, 0, 2 /* nt: streamed once */);
            };
            if (rt == 0) {
                if (he == 0) wino43_finish<0, RELU>(A, B, theirs, b4, store, live, true, BX_AZI * 16, 15u);
                else wino43_finish<1, RELU>(A, B, theirs, b4, store, live, true, BX_AZI * 16, 15u);
            } else {
                if (he == 0) wino43m_finish<0, RELU>(A, B, theirs, b4, store, live, BX_AZI * 16);
                else wino43m_finish<1, RELU>(A, B, theirs, b4, store, live, BX_AZI * 16);
            }
        }
        __syncthreads();                                    // the exchange is free again (next row tile / next group's V planes)
    }
}

template <int NCHUNK, int COUT, int CW, bool RELU>
__global__ __launch_bounds__(CT, 2) void wino43m_kernel(const float* __restrict__ in, int units, const float* __restrict__ U,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    constexpr int NT = COUT / 16, NCW = CW / 16;    // column tiles of the layer / of a workgroup
    constexpr int NPU = BX_EA * 4, NPIECE = G4 * NPU, NLD = (NPIECE + CT - 1) / CT;
    static_assert(NLD * 2 + 1 <= NPH && (NCW == 4 || NCW == 2), "slab traffic fits the plane loop; 8 or 4 compute waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* slab = reinterpret_cast<float*>(smem);
    float* Vp = slab + SLABF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, ctl = wave >> 1;
    const bool cw = ctl < NCW;                      // compute wave (MFMAs + output); with CW = 32 waves 4..7 only stage and transform
    // effective half: which xi rows the wave owns.  The waves of a SIMD are w and w + 4 = (ctl, half) and (ctl + 2, half): swapping the
    // halves of the column tiles 2, 3 gives every SIMD one wave with 144 and one with 120 MFMAs per chunk
#ifndef BX_W43M_FLIP
#define BX_W43M_FLIP 1              // 0: no swap, 1: swap the halves of the column tiles 2, 3, 2: of the column tiles 1, 3 (timing experiments)
#endif
    const int he = (NCW == 4 && BX_W43M_FLIP == 1) ? (half ^ (ctl >> 1)) : ((NCW == 4 && BX_W43M_FLIP == 2) ? (half ^ (ctl & 1)) : half);
    const int ctg = (int)blockIdx.y * NCW + (cw ? ctl : 0);
    const int li = lane & 15, kk = lane >> 4;
    const int ngroups = (units * TC4 + BLK - 1) / BLK;              // items of 16 column blocks
    if ((int)blockIdx.x >= ngroups) return;

    for (int i = tid; i < (int)(W43M_LDS / 16); i += CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- slab traffic: as k_wino43.hip (four unit slots from the item's first unit)
    // per piece TWO lane constants: the source byte offset (-1: none) and {destination byte 17 bits | halo copy 2 bits} packed -- the three
    // separate constants of k_wino43.hip are 15 registers that this kernel (two transform paths, two fragment rings) does not have; spilled,
    // their reloads sat in front of every slab request with an s_waitcnt vmcnt(0) = a drain of the weight ring per piece (first build:
    // +25 % per layer).  The destination is unpacked where it is used (behind an opaque copy: hoisted, the unpacked values are spilled again).
    float4 st[NLD];
    int lsrc[NLD];
    unsigned lpk[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int f = tid + q * CT;
        lsrc[q] = -1; lpk[q] = 0u;
        if (f < NPIECE) {
            const int g = f / NPU, fr = f - g * NPU;
            const int p = fr >> 2, part = fr & 3;
            const int h = p / BX_AZI, w = p - h * BX_AZI;
            lsrc[q] = (g * NCHUNK * NPU + fr) * 16;
            const int dst = g * UP3 + (h + 1) * RP3 + (w + 1) * ROWF + part * 4;
            const int halo = w == 0 ? 1 : (w == BX_AZI - 1 ? 2 : 0);
            static_assert(SLABF * 4 < (1 << 17), "packed slab piece constants");
            lpk[q] = (unsigned)(dst * 4) | ((unsigned)halo << 29);
        }
    }
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((long long)units * NCHUNK * NPU * 16 < 0x7fffffffLL ? (long long)units * NCHUNK * NPU * 16 : 0x7fffffffLL), 0x00020000);
    auto gload1 = [&](int q, int ug_, int cc_) {
        const int u0_ = (ug_ * BLK) / TC4;         // first unit of the item's window (slot 0)
        const int soff = ((u0_ * NCHUNK + cc_) * NPU) * 16;
        const int lim = (units - u0_) * NCHUNK * NPU * 16;
        const f32x4 v = (lsrc[q] >= 0 && lsrc[q] < lim) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, lsrc[q], soff, 2 /* nt */)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        st[q] = make_float4(v.x, v.y, v.z, v.w);
    };
    auto lwrite1 = [&](int q) {
        if (lsrc[q] >= 0) {
            unsigned pk = lpk[q];
            asm volatile("" : "+v"(pk));
            char* d = smem + (pk & 0x1ffffu);
            *reinterpret_cast<float4*>(d) = st[q];
            const unsigned halo = pk >> 29;
            if (halo != 0) *reinterpret_cast<float4*>(d + (halo == 1 ? BX_AZI * ROWF * 4 : -BX_AZI * ROWF * 4)) = st[q];
        }
    };

    // ---- transform role: (tile row tR = 4 wave + lane / 16 of the item, channel slot lane % 16).  tR < 16 (waves 0..3): the F(4x4) tile of
    //      column block tR; tR >= 16 (waves 4..7): the F(3x4) tile of column block tR - 16 (window rows h = 3..7 = slab rows 4..8 of the slot)
    const int tR = 4 * wave + (lane >> 4);
    const bool tB = wave >= 4;                      // wave-uniform
    float* vdst = Vp + tR * ROWF + (lane & 15);
    const float* wsrc = slab;
    auto set_window = [&](int ioff_) {
        const int blk = ioff_ + (tR & 15);
        const int tg = blk / TC4, ttc = blk - tg * TC4;
        wsrc = slab + tg * UP3 + (tB ? 4 * RP3 : 0) + (4 * ttc) * ROWF + (lane & 15);
    };
    auto transform = [&]() {
        if (!tB) {
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
                for (int nu = 0; nu < 6; ++nu) vdst[(x * 6 + nu) * VPL4] = o[nu];
            }
        } else {
            float t[5][6];                              // t[xi][j]: B3^T d down column j (five window rows)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float o[5];
                bt5s(wsrc[j * ROWF], wsrc[RP3 + j * ROWF], wsrc[2 * RP3 + j * ROWF], wsrc[3 * RP3 + j * ROWF], wsrc[4 * RP3 + j * ROWF], o);
#pragma unroll
                for (int x = 0; x < 5; ++x) t[x][j] = o[x];
            }
#pragma unroll
            for (int x = 0; x < 5; ++x) {
                float o[6];
                bt6s(t[x][0], t[x][1], t[x][2], t[x][3], t[x][4], t[x][5], o);
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) vdst[(x * 6 + nu) * VPL4] = o[nu];
            }
        }
    };

    // weight fragments [chunk * 36 + plane][column tile][{U_A, U_B}][lane][4]: the wave's planes are he * 18 + 0..17
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, NCHUNK * NPL * NT * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long long)units * NT * BX_EA * 64 < 0x7fffffffLL ? (long long)units * NT * BX_EA * 64 : 0x7fffffffLL), 0x00020000);
    // Everything from here on is specialised on the wave's effective half (two copies of the item loop, one wave-uniform branch at the
    // top): which plane steps skip the F(3x4) row tile is then a compile-time property of the unrolled plane loop.  A run-time test
    // around MFMAs -- or a diamond of two specialised plane loops that merge again -- makes hipcc keep the accumulators of the two paths
    // in different registers (290-340 spilled VGPRs, vmcnt(0) drains inside the loop).  Both copies execute the same barriers.
    auto run = [&](auto HE) {
    constexpr int HEC = decltype(HE)::value;
    const int ubase = ((HEC * NPH) * NT + ctg) * 2048;
    const int ulane = lane * 16;
    auto bloadA = [&](int q) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, ubase + q * (NT * 2048), 0));
        return make_float4(v.x, v.y, v.z, v.w);
    };
    auto bloadB = [&](int q) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane + 1024, ubase + q * (NT * 2048), 0));
        return make_float4(v.x, v.y, v.z, v.w);
    };
    const char* abase = reinterpret_cast<const char*>(Vp) + ((HEC * NPH * VR4 + li) * ROWF + kk * 4) * 4;

    f32x4 acc[NPH][RT4];
    // fragment rings of TWO planes each (k_wino43.hip: one ring of three): two streams cost twice the registers, and a plane step of the
    // two waves of a SIMD (2 x 8 MFMAs = 512 cycles) already covers an L2 access
    constexpr int RING = BX_W43M_RING;
    static_assert(NPH % RING == 0, "plane q of the next chunk lands in the slot plane q is read from");
    float4 bringA[RING], bringB[RING];
#pragma unroll
    for (int p = 0; p < RING; ++p) { bringA[p] = bloadA(p); bringB[p] = bloadB(p); }

    int ug = blockIdx.x;
    const int gstep = (int)gridDim.x;
    int lg = ug, lc = 0;                            // the (group, chunk) the NEXT request fetches
    auto ladv = [&]() { if (++lc == NCHUNK) { lc = 0; lg += gstep; } };
#pragma unroll
    for (int q = 0; q < NLD; ++q) gload1(q, lg, lc);
    ladv();
    __syncthreads();                 // zero fill complete
    if (tid < CW) Vp[NPL * VPL4 + (tid >> 4) * 16 + (tid & 3) * 4 + ((tid & 15) >> 2)] = bias[(int)blockIdx.y * CW + tid];
#pragma unroll
    for (int q = 0; q < NLD; ++q) lwrite1(q);
    bool st_live = lg < ngroups;

    for (;;) {
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int rt = 0; rt < RT4; ++rt) acc[p][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ugn = ug + gstep;
        const int u0 = (ug * BLK) / TC4, ioff = ug * BLK - u0 * TC4;     // first unit of the item, offset of its first column block in it
        set_window(ioff);
#pragma unroll 1
        for (int cc = 0; cc < NCHUNK; ++cc) {
            __syncthreads();         // the slab of this chunk is complete; every wave is done with the V planes of the chunk before
            st_live = lg < ngroups;
            const bool st_was = st_live;
            if (st_live) {
#pragma unroll
                for (int q = 0; q < NLD; ++q) gload1(q, lg, lc);
                ladv();
            }
            __builtin_amdgcn_sched_barrier(0);
            transform();
            __syncthreads();         // V complete; the slab is free
            if (cw) {
                const int cn = cc + 1 == NCHUNK ? 0 : cc + 1;
                // the plane loop, specialised on the wave's effective half: which plane steps skip the F(3x4) row tile is then a compile-time
                // property of the unrolled loop (a run-time test around MFMAs that write accumulator registers costs hipcc its schedule:
                // 291 spilled VGPRs and 40 vmcnt(0) drains inside the loop in the first build)
                {
                    f32x4 ar[3];
                    ar[0] = *reinterpret_cast<const f32x4*>(abase);
                    ar[1] = *reinterpret_cast<const f32x4*>(abase + (16 * ROWF) * 4);
#pragma unroll
                    for (int p = 0; p < NPH; ++p) {
                        const float4 bqa = bringA[p % RING], bqb = bringB[p % RING];
                        // the slots are refilled BEFORE the plane's MFMAs (they read the copies)
                        bringA[p % RING] = p + RING < NPH ? bloadA(cc * NPL + p + RING) : bloadA(cn * NPL + p + RING - NPH);
                        bringB[p % RING] = p + RING < NPH ? bloadB(cc * NPL + p + RING) : bloadB(cn * NPL + p + RING - NPH);
                        // the F(3x4) row tile has no planes xi = 5: the half that owns xi 3..5 skips its last six plane steps there
                        const bool bact = p < 12 || HEC == 0;
                        if (NCW == 2) {
                            const f32x4 a0 = ar[0], a1 = ar[1];
                            if (p + 1 < NPH) {
                                ar[0] = *reinterpret_cast<const f32x4*>(abase + (((p + 1) * VR4) * ROWF) * 4);
                                ar[1] = *reinterpret_cast<const f32x4*>(abase + (((p + 1) * VR4 + 16) * ROWF) * 4);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (bact) {
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.x, a0.x, acc[p][0], 0, 0, 0);
                                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqb.x, a1.x, acc[p][1], 0, 0, 0);
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.y, a0.y, acc[p][0], 0, 0, 0);
                                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqb.y, a1.y, acc[p][1], 0, 0, 0);
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.z, a0.z, acc[p][0], 0, 0, 0);
                                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqb.z, a1.z, acc[p][1], 0, 0, 0);
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.w, a0.w, acc[p][0], 0, 0, 0);
                                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqb.w, a1.w, acc[p][1], 0, 0, 0);
                            } else {
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.x, a0.x, acc[p][0], 0, 0, 0);
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.y, a0.y, acc[p][0], 0, 0, 0);
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.z, a0.z, acc[p][0], 0, 0, 0);
                                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa.w, a0.w, acc[p][0], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            continue;
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rt = 0; rt < RT4; ++rt) {
                            const int s0 = p * RT4 + rt, s2 = s0 + 2;
                            if (s2 < NPH * RT4) ar[s2 % 3] = *reinterpret_cast<const f32x4*>(abase + (((s2 / RT4) * VR4 + (s2 % RT4) * 16) * ROWF) * 4);
                            const f32x4 a = ar[s0 % 3];
                            const float4 bq = rt == 0 ? bqa : bqb;
                            if (rt == 0 || bact) {
                                acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq.x, a.x, acc[p][rt], 0, 0, 0);
                                acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq.y, a.y, acc[p][rt], 0, 0, 0);
                                acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq.z, a.z, acc[p][rt], 0, 0, 0);
                                acc[p][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq.w, a.w, acc[p][rt], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                // the slab writes of the next chunk, all at once behind the last plane (one drain of the loads in flight per chunk)
                if (st_was) {
#pragma unroll
                    for (int q = 0; q < NLD; ++q) lwrite1(q);
                }
            } else {                 // CW = 32: the waves without a column tile carry the slab traffic only
#pragma unroll
                for (int q = 0; q < NLD; ++q) {
                    if (st_was) lwrite1(q);
                }
            }
        }
        __syncthreads();             // every wave is done with the V planes: their bytes carry the exchange now
        wino43m_output<NT, RELU, HEC>(acc, Vp, cw, wave, lane, u0, ioff, units, ctg, ors);
        ug = ugn;
        if (ug >= ngroups) break;
    }
    };
    if (he == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

template <int NCHUNK, int COUT, int CW, bool RELU>
int launch_wino43m(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, int units, float* out)
{
    if (L.nchunk != NCHUNK || L.cout != COUT || (L.relu != 0) != RELU || !L.Wwino43) {
        bx_set_error("mixed-tile winograd layer %d: geometry mismatch (%d chunks, %d channels)", layer, L.nchunk, L.cout);
        return BX_ERR_STATE;
    }
    // 32-bit byte offsets inside the kernel: 2 GiB or more of maps is "not served" (the direct kernels take it)
    if (!w43::fits_i32((long long)units * NCHUNK * BX_EA * 64) || !w43::fits_i32((long long)units * (COUT / 16) * BX_EA * 64)) return -1;
    auto k = wino43m_kernel<NCHUNK, COUT, CW, RELU>;
    int& cap = c->wino_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W43M_LDS));
        cap = c->n_cu / (COUT / CW);
        if (cap < 1) cap = 1;
        if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
    }
    int grid = (units * TC4 + BLK - 1) / BLK;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k, dim3(grid, COUT / CW), dim3(CT), W43M_LDS, s, in, units, L.Wwino43, L.b, out, c->skip);
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
void g5(const double g[3], double o[5])       // oracle/bx_oracle.c::wino43m_g5
{
    o[0] = ((g[0] + g[1]) + g[2]) / 6.0;
    o[1] = -((g[0] - g[1]) + g[2]) / 6.0;
    o[2] = -((4.0 * g[0] + 2.0 * g[1]) + g[2]) / 6.0;
    o[3] = ((4.0 * g[0] - 2.0 * g[1]) + g[2]) / 6.0;
    o[4] = g[2] / 4.0;
}
}  // namespace

// U_A = G g G^T (F(4x4) tiles) and U_B = G3 g G^T (F(3x4) tiles; its planes 30..35 do not exist: zeros) of every (chunk, channel, output
// channel) in binary64, rounded once, packed as MFMA fragments [chunk * 36 + plane][column tile][{A, B}][lane = kk*16 + li][4]
int bxk_wino43m_weights(const float* w /* [nchunk][9][16][cout] */, int nchunk, int cout, float** d_out)
{
    const int nt = cout / 16;
    std::vector<float> frag((size_t)nchunk * NPL * nt * 2 * 64 * 4, 0.0f);
    for (int cc = 0; cc < nchunk; ++cc)
        for (int ch = 0; ch < 16; ++ch)
            for (int o = 0; o < cout; ++o) {
                double g[3][3], GA[6][3], GB[5][3];
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw) g[kh][kw] = (double)w[(((size_t)cc * 9 + kh * 3 + kw) * 16 + ch) * cout + o];
                for (int kw = 0; kw < 3; ++kw) {
                    const double col[3] = {g[0][kw], g[1][kw], g[2][kw]};
                    double r6[6], r5[5];
                    g6(col, r6);
                    g5(col, r5);
                    for (int xi = 0; xi < 6; ++xi) GA[xi][kw] = r6[xi];
                    for (int xi = 0; xi < 5; ++xi) GB[xi][kw] = r5[xi];
                }
                const int kk = ch & 3, i = ch >> 2, t = o / 16, li = bx_chunk_slot(o % 16);
                for (int xi = 0; xi < 6; ++xi) {
                    double ua[6], ub[6];
                    g6(GA[xi], ua);
                    if (xi < 5) g6(GB[xi], ub);
                    for (int nu = 0; nu < 6; ++nu) {
                        const int pl = xi * 6 + nu;
                        const size_t base = ((size_t)(cc * NPL + pl) * nt + t) * 2;
                        frag[(((base + 0) * 4 + kk) * 16 + li) * 4 + i] = (float)ua[nu];
                        if (xi < 5) frag[(((base + 1) * 4 + kk) * 16 + li) * 4 + i] = (float)ub[nu];
                    }
                }
            }
    BX_HIP(hipMalloc(reinterpret_cast<void**>(d_out), frag.size() * sizeof(float)));
    BX_HIP(hipMemcpy(*d_out, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice));
    return BX_OK;
}

// layer of Cylindrical_Net in the mixed-tile form; -1 when this layer / unit count is not served (caller falls back)
int bxk_wino43m(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (units_dev || max_units < 1) return -1;
    const ConvLayerDev& L = c->desc[layer];
    switch (layer) {
        case 0: return launch_wino43m<3, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 1: return launch_wino43m<4, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 2: return launch_wino43m<4, 128, 64, true>(c, layer, s, L, in, max_units, out);
        case 3: return launch_wino43m<8, 128, 64, true>(c, layer, s, L, in, max_units, out);
        case 4: return launch_wino43m<8, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 5: return launch_wino43m<4, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 6: return launch_wino43m<4, 32, 32, true>(c, layer, s, L, in, max_units, out);
        case 7: return launch_wino43m<2, 32, 32, false>(c, layer, s, L, in, max_units, out);
    }
    return -1;
}
