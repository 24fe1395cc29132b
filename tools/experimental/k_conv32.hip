// k_conv32.hip -- Cylindrical_Net (reference models/patchnet.py:49-84; circular-azimuth / zero-elevation padding of
// utils/common.py:265-310) on v_mfma_f32_32x32x2_f32, gfx950.
//
// Same implicit GEMM and the same arithmetic as k_conv.hip -- rows = (unit, position), columns = output channels,
// K = (16-channel chunk, tap, channel), accumulation bias -> chunk -> tap -> channel, an exact k-ordered fp32 fmaf chain --
// but on 32 x 32 output tiles: one MFMA covers 32 positions x 32 channels x 2 input channels in 64 matrix-pipe cycles and its
// issue interval equals its dependent-accumulator latency, so ONE wave per SIMD keeps the pipe full with one accumulator chain.
//
// Structure (found with in-kernel cycle stamps and SQ_VALU_MFMA_BUSY_CYCLES, LABBOOK.md §2):
//   * On gfx950 the f32 MFMA runs at the f32 VECTOR rate.  Two waves that both stream f32 MFMAs on one SIMD reach only ~80 % of
//     the pipe (measured: 2 workgroups x 4 waves per CU, 81 % busy once their non-MFMA phases had been shrunk to nothing), one
//     wave alone ~95 %; and while a wave streams MFMAs, a co-resident wave's VALU instruction gets in only at an MFMA boundary
//     (~64 cycles of wall time per instruction).
//   * So a workgroup is 4 COMPUTE waves (one per SIMD: nothing but ds_read_b128 + MFMA + one barrier per chunk) and 4 LOADER
//     waves (one per SIMD: global -> registers -> LDS staging of the NEXT slab into the other buffer, all address arithmetic,
//     the group tickets), one workgroup per CU.  The loaders' few dozen instructions per chunk crawl between the compute
//     wave's MFMAs and still finish long before the 41 500-cycle chunk does.
//   * accumulators are initialised by one extra MFMA per tile (A = 1, B = bias, C = inline 0: fma(1, bias, 0) = bias exactly),
//     not by 144 v_mov; the epilogue addresses rows through an LDS offset table and costs one v_max (ReLU) per value.
//
// A compute wave owns ONE 32-channel column tile and 9 row tiles (144 accumulator registers): G = 2 units per group for the
// 128-channel layers (4 column tiles x 9 row tiles), G = 4 for the 64-channel ones (2 x 18).  The 16-channel slab of the G units
// is staged in LDS with its halo exactly as in k_conv.hip ((7+2) x (20+2) rows of 80 bytes per unit), so a tap is a compile-time
// byte offset in the ds_read.  Lane (i = l & 31, kk = l >> 5) reads position i's float4 #kk and #kk+2 of the row: with the
// chunk-slot order of the feature maps (slot 4*(c%4) + c/4) these are channels {kk, 4+kk, 8+kk, 12+kk} and {2+kk, 6+kk, 10+kk,
// 14+kk} -- the A operands of the eight MFMAs of a 16-channel chunk in natural channel order (MFMA j consumes channels 2j, 2j+1).
#include "bx_common.h"
#include <cstdlib>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ROWB = 80;                                   // bytes per LDS row (16 floats + 4 pad)
constexpr int CYL_W = BX_AZI + 2, CYL_ROWS = (BX_ELE + 2) * CYL_W;   // 22, 198
constexpr int UNIT_CHUNK_BYTES = BX_EA * 64;                // one unit's 16-channel map: 8960 bytes
constexpr int NO_DST = 0xffff;                              // staging table entry of a piece beyond the slab

template <int NCHUNK, int COUT, int G, bool RELU>
struct C32 {
    static constexpr int NWC = 4, NWL = 4;                  // compute / loader waves
    static constexpr int CT = (NWC + NWL) * 64, LT = NWL * 64;
    static constexpr int NCT = COUT / 32;                   // column tiles: 4 / 2
    static constexpr int NRG = NWC / NCT;                   // row groups: 1 / 2
    static constexpr int M = G * BX_EA;
    static constexpr int MT = (M + 31) / 32;
    static constexpr int RTW = MT / NRG;                    // row tiles per compute wave
    static constexpr int ROWS = G * CYL_ROWS;
    static constexpr int NF4 = G * BX_EA * 4;               // float4 pieces of a slab
    static constexpr int NLD = (NF4 + LT - 1) / LT;         // pieces per loader thread
    static constexpr int SLAB = ROWS * ROWB;
    // LDS map (bytes): slab 0 | slab 1 | mailbox (16) | dst1 u16[NLD*LT] | dst2 u16[NLD*LT] | unit u8[NLD*LT] | rowoff i32[MT*32] |
    //                  epilogue scratch 4 x 32 x 36 floats
    static constexpr int OFF_MAIL = 2 * SLAB;
    static constexpr int OFF_D1 = OFF_MAIL + 16, OFF_D2 = OFF_D1 + NLD * LT * 2, OFF_G = OFF_D2 + NLD * LT * 2;
    static constexpr int OFF_ROW = (OFF_G + NLD * LT + 15) / 16 * 16;
    static constexpr int OFF_SCR = OFF_ROW + MT * 32 * 4;
    static constexpr int SCR_WAVE = 32 * 36 * 4;
    static constexpr int LDS_BYTES = OFF_SCR + NWC * SCR_WAVE;
    static_assert(COUT % 32 == 0 && NWC % NCT == 0 && MT % NRG == 0, "compute waves must tile the output evenly");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(SLAB < NO_DST, "slab offsets travel as 16 bits");
    static_assert(NCHUNK >= 3, "the loaders run two slabs ahead of the ticket of the next group");
    static_assert(RELU, "the MFMA bias initialisation relies on the ReLU epilogue for the sign of a zero bias");
};

// Group walk: a workgroup starts with group blockIdx.x and then takes the next unclaimed group from a device-side ticket counter
// (ctr[0]; one returning atomic per group, ~1 us against >= 100 us of work per group): with 2 500 equal groups on 256 CUs a static
// stride walk leaves the CUs that got 10 groups running alone while the others (9) idle.  The last workgroup to leave resets the
// counters (ctr[1] counts departures), so launches need no memset in between.  A ragged last group is shifted back to the last G
// units: the units it shares with its neighbour are computed twice, with identical results (units >= G: launcher).
template <int NCHUNK, int COUT, int G, bool RELU>
__global__ __launch_bounds__(512, 2) void conv32_kernel(const float* __restrict__ in, int units, const float* __restrict__ W32,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        const int32_t* __restrict__ skip, int32_t* __restrict__ ctr,
                                                        long long* __restrict__ dbg)
{
    if (skip && *skip) return;
    using C = C32<NCHUNK, COUT, G, RELU>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ngroups = (units + G - 1) / G;
    if ((int)blockIdx.x >= ngroups) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= C::NWC;                     // wave-uniform role

    // ---- one-off set-up (all 512 threads): zero both slabs (halo rows stay zero for the whole kernel), staging tables, row table
    for (int i = tid; i < 2 * C::SLAB / 16 + 1; i += C::CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        unsigned short* d1 = reinterpret_cast<unsigned short*>(smem + C::OFF_D1);
        unsigned short* d2 = reinterpret_cast<unsigned short*>(smem + C::OFF_D2);
        unsigned char* ug = reinterpret_cast<unsigned char*>(smem + C::OFF_G);
        for (int f = tid; f < C::NLD * C::LT; f += C::CT) {
            const int row = f >> 2, part = f & 3;
            int a = NO_DST, b = NO_DST, g = 255;                // pieces beyond the slab: not written, source = piece 0
            if (f < C::NF4) {
                g = row / BX_EA;
                const int p = row - g * BX_EA, h = p / BX_AZI, w = p - h * BX_AZI;
                a = (g * CYL_ROWS + (h + 1) * CYL_W + (w + 1)) * ROWB + part * 16;
                b = a;                                          // no wrap copy: the same store twice
                if (w == 0) b = a + BX_AZI * ROWB;              // column 20 = column 0
                if (w == BX_AZI - 1) b = a - BX_AZI * ROWB;     // column -1 = column 19
            }
            d1[f] = (unsigned short)a; d2[f] = (unsigned short)b; ug[f] = (unsigned char)g;
        }
        // output byte offset of tile row m inside the group's output block (channel chunk 0): unit g, position pos
        int* ro = reinterpret_cast<int*>(smem + C::OFF_ROW);
        for (int m = tid; m < C::MT * 32; m += C::CT) {
            const int g = m / BX_EA, pos = m - g * BX_EA;
            ro[m] = m < C::M ? g * (COUT / 16) * UNIT_CHUNK_BYTES + pos * 64 : 0x7fffffff;
        }
    }
    int* mailbox = reinterpret_cast<int*>(smem + C::OFF_MAIL);
    auto unit0 = [&](int grp_) { return grp_ * G < units - G ? grp_ * G : units - G; };
    __syncthreads();          // zero fill + tables complete

    int grp = blockIdx.x, grp_next = ngroups;
    int s = 0;                // slab counter of this workgroup: slab s lives in buffer s & 1

    if (loader) {
        // =================================================================================== loader waves
        const int lt = tid - C::NWC * 64;
        const unsigned lt16 = (unsigned)lt * 16u;
        const char* inb = reinterpret_cast<const char*>(in);
        constexpr unsigned GS = (unsigned)(NCHUNK - 1) * UNIT_CHUNK_BYTES;   // extra bytes between the same chunk of consecutive units
        const unsigned short* d1 = reinterpret_cast<const unsigned short*>(smem + C::OFF_D1);
        const unsigned short* d2 = reinterpret_cast<const unsigned short*>(smem + C::OFF_D2);
        const unsigned char* ug = reinterpret_cast<const unsigned char*>(smem + C::OFF_G);
        float4 st[C::NLD];
        // source = scalar base + tid*16 + q*4096 + unit * stride (one v_mad per piece)
        auto gload = [&](int g_, int cc) {
            const char* base = inb + ((size_t)unit0(g_) * NCHUNK + cc) * UNIT_CHUNK_BYTES;
#pragma unroll
            for (int q = 0; q < C::NLD; ++q) {
                const unsigned g = ug[lt + q * C::LT];
                unsigned voff = lt16 + (unsigned)(q * C::LT * 16) + g * GS;
                if ((q + 1) * C::LT > C::NF4) voff = g == 255u ? 0u : voff;            // last piece only: beyond the slab -> piece 0
                st[q] = *reinterpret_cast<const float4*>(base + voff);
            }
        };
        auto lwrite = [&](int buf) {
            char* sb = smem + buf * C::SLAB;
#pragma unroll
            for (int q = 0; q < C::NLD; ++q) {
                const int a = d1[lt + q * C::LT], b = d2[lt + q * C::LT];
                if ((q + 1) * C::LT <= C::NF4 || a != NO_DST) {
                    *reinterpret_cast<float4*>(sb + a) = st[q];
                    *reinterpret_cast<float4*>(sb + b) = st[q];
                }
            }
        };
        // prologue: slab 0 in place, slab 1 in flight
        gload(grp, 0);
        lwrite(0);
        gload(grp, 1);
        __syncthreads();                                     // B_0
        for (;;) {
#pragma unroll 1
            for (int cc = 0; cc < NCHUNK; ++cc, ++s) {
                int ticket = 0;
                if (cc == 0 && lt == 0) ticket = atomicAdd(&ctr[0], 1);
                if (cc == 1) grp_next = (int)gridDim.x + __builtin_amdgcn_readfirstlane(*mailbox);
                // slab s+1 (in registers since the previous chunk) -> the buffer the compute waves left at the last barrier
                if (cc + 1 < NCHUNK || grp_next < ngroups) lwrite((s + 1) & 1);
                // slab s+2 -> registers
                if (cc + 2 < NCHUNK) gload(grp, cc + 2);
                else if (grp_next < ngroups) gload(grp_next, cc + 2 - NCHUNK);
                if (cc == 0 && lt == 0) *mailbox = ticket;   // the atomic has had the whole hand-off to return
                __syncthreads();                             // B_{s+1}
            }
            grp = grp_next;
            if (grp >= ngroups) break;
        }
        if (lt == 0) {
            const int gone = atomicAdd(&ctr[1], 1);
            if (gone == (ngroups < (int)gridDim.x ? ngroups : (int)gridDim.x) - 1) { ctr[0] = 0; ctr[1] = 0; __threadfence(); }
        }
        return;
    }

    // ======================================================================================= compute waves
    // optional cycle stamps (BX_BALL_DEBUG): 16 workgroups x 32 {t0, [taps done, barrier passed] per chunk of the SECOND group}
    const bool tr_ = dbg != nullptr && (blockIdx.x % 15) == 0 && blockIdx.x / 15 < 16 && tid == 0;
    long long* td_ = dbg + (blockIdx.x / 15) * 32;
    long long t0_ = 0;
    int gi_ = 0;
#define C32_TR(k) do { if (tr_ && gi_ == 1) td_[k] = __builtin_readcyclecounter() - t0_; } while (0)
    const int ct = wave % C::NCT, rg = wave / C::NCT;
    const int li = lane & 31, kk = lane >> 5;

    // window origin (byte address) of every tile row owned by this lane, in the CURRENT slab buffer
    int abase[C::RTW];
#pragma unroll
    for (int k = 0; k < C::RTW; ++k) {
        const int m = (rg + k * C::NRG) * 32 + li;
        int r = 0;                                          // rows beyond M compute garbage that is never stored
        if (m < C::M) { const int g = m / BX_EA, pos = m - g * BX_EA; const int h = pos / BX_AZI, w = pos - h * BX_AZI; r = g * CYL_ROWS + h * CYL_W + w; }
        abase[k] = r * ROWB + kk * 16;
    }
    int bufstep = C::SLAB;                                  // added to abase after every slab (sign flips)

    // B fragments: [chunk*9 + tap][column tile][lane][8]: element j = W[chunk][tap][2j + kk][ct*32 + li]
    const char* wb = reinterpret_cast<const char*>(W32) + (size_t)ct * 64 * 32;      // uniform (saddr); per lane: + lane * 32
    const unsigned lane32 = (unsigned)lane * 32u;
    constexpr size_t WSTEP = (size_t)C::NCT * 64 * 32;      // bytes per (chunk, tap)
    constexpr int NCT9 = NCHUNK * 9;
    auto ldB = [&](int nx, int half) { return *reinterpret_cast<const float4*>(wb + (size_t)nx * WSTEP + (lane32 + 16u * half)); };

    // bias as an MFMA: k = 0 lanes carry A = 1 and B = bias[column], k = 1 lanes zeros
    const float a_one = kk == 0 ? 1.0f : 0.0f;
    const float b_bias = kk == 0 ? bias[ct * 32 + li] : 0.0f;

    // epilogue addressing (per lane, constant): scratch write base, scratch read base, row-table base, output column offset
    const int scr_off = C::OFF_SCR + wave * C::SCR_WAVE;
    const int scol = (li >> 4) * 16 + 4 * (li & 3) + ((li & 15) >> 2);       // chunk-slot order (bx_chunk_slot)
    const int ep_w = scr_off + (4 * kk * 36 + scol) * 4;
    const int ep_r = scr_off + (lane >> 3) * 144 + (lane & 7) * 16;
    const int ep_t = C::OFF_ROW + (rg * 32 + (lane >> 3)) * 4;
    const int ep_c = (ct * 2 + ((lane & 7) >> 2)) * UNIT_CHUNK_BYTES + (lane & 3) * 16;

    float4 b0 = ldB(0, 0), b1 = ldB(0, 1);
    f32x16 acc[C::RTW];
    __syncthreads();                                         // B_0: slab 0 in place
    for (;;) {
        if (tr_ && gi_ == 1) { t0_ = __builtin_readcyclecounter(); td_[0] = t0_; }
        // accumulators = bias: one MFMA per tile with the inline constant 0 as C (the asm keeps the nine from being merged into
        // one MFMA plus 144 register copies)
#pragma unroll
        for (int k = 0; k < C::RTW; ++k) {
            float a1 = a_one;
            asm volatile("" : "+v"(a1));
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b_bias, z, 0, 0, 0);
        }
#pragma unroll 1
        for (int cc = 0; cc < NCHUNK; ++cc, ++s) {
            if (cc == 1) grp_next = (int)gridDim.x + __builtin_amdgcn_readfirstlane(*mailbox);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                int nx = cc * 9 + tp + 1;
                if (nx >= NCT9) nx = 0;                     // first tap of the next group's first chunk
                const float4 n0 = ldB(nx, 0), n1 = ldB(nx, 1);
                const int tb = ((tp / 3) * CYL_W + tp % 3) * ROWB;
                // software pipeline over the wave's row tiles: the two ds_read_b128 of tile k+1 are in flight while the eight
                // MFMAs of tile k (512 matrix-pipe cycles) run; sched_barrier keeps the compiler from hoisting every read of the
                // tap to the top (18 x 4 registers) or interleaving the accumulator chains
                f32x4 lo = *reinterpret_cast<const f32x4*>(smem + abase[0] + tb);
                f32x4 hi = *reinterpret_cast<const f32x4*>(smem + abase[0] + tb + 32);
#pragma unroll
                for (int k = 0; k < C::RTW; ++k) {
                    f32x4 nlo = lo, nhi = hi;
                    if (k + 1 < C::RTW) {
                        nlo = *reinterpret_cast<const f32x4*>(smem + abase[k + 1 < C::RTW ? k + 1 : 0] + tb);
                        nhi = *reinterpret_cast<const f32x4*>(smem + abase[k + 1 < C::RTW ? k + 1 : 0] + tb + 32);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.x, b0.x, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.x, b0.y, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.y, b0.z, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.y, b0.w, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.z, b1.x, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.z, b1.y, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo.w, b1.z, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi.w, b1.w, acc[k], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    lo = nlo; hi = nhi;
                }
                b0 = n0; b1 = n1;
            }
            // the other buffer next
#pragma unroll
            for (int k = 0; k < C::RTW; ++k) abase[k] += bufstep;
            bufstep = -bufstep;
            C32_TR(1 + 2 * cc);
            if (cc + 1 == NCHUNK) {
                // ---- epilogue of the group: ReLU, then every wave turns its 32 x 32 tiles (accumulator layout: lane = column li,
                //      register r = row (r & 3) + 8 (r >> 2) + 4 kk) into rows of 2 chunks x 16 slots in its 32 x 36-float scratch
                //      and stores 16 bytes per lane: a wave instruction writes 8 complete 128-byte position records.
                char* ob = reinterpret_cast<char*>(out) + (size_t)unit0(grp) * (COUT / 16) * UNIT_CHUNK_BYTES;
#pragma unroll
                for (int k = 0; k < C::RTW; ++k) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[k][r];
                        // ReLU as ONE v_max_f32 (the C expression costs a canonicalising second one); max(NaN, 0) = 0 and
                        // max(-0, +0) = +0, like  v > 0 ? v : 0
                        asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(v));
                        *reinterpret_cast<float*>(smem + ep_w + ((r & 3) + 8 * (r >> 2)) * 144) = v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int roff = *reinterpret_cast<const int*>(smem + ep_t + (k * C::NRG * 32 + 8 * e) * 4);
                        const float4 v4 = *reinterpret_cast<const float4*>(smem + ep_r + e * 8 * 144);
                        // rows beyond M exist only in the wave's last tile
                        if (k + 1 < C::RTW || C::M % 32 == 0 || roff != 0x7fffffff) *reinterpret_cast<float4*>(ob + (unsigned)(roff + ep_c)) = v4;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            __syncthreads();                                 // B_{s+1}: this slab is free, the next one is in place
            C32_TR(2 + 2 * cc);
        }
        ++gi_;
        grp = grp_next;
        if (grp >= ngroups) break;
    }
}

template <int NCHUNK, int COUT, int G, bool RELU>
int launch32(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, int units, float* out)
{
    using C = C32<NCHUNK, COUT, G, RELU>;
    if (L.nchunk != NCHUNK || L.ntaps != 9 || L.p_in != BX_EA || L.p_out != BX_EA || L.cout != COUT || (L.relu != 0) != RELU || !L.W32) {
        bx_set_error("conv32 layer geometry mismatch (%d %d %d %d %d)", L.nchunk, L.ntaps, L.p_in, L.p_out, L.cout);
        return BX_ERR_STATE;
    }
    auto k = conv32_kernel<NCHUNK, COUT, G, RELU>;
    int& cap = c->conv32_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        cap = c->n_cu;                                       // one workgroup per CU
        if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
    }
    int grid = (units + G - 1) / G;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::CT), C::LDS_BYTES, s, in, units, L.W32, L.b, out, c->skip, c->conv_ctr + 2 * layer,
                       (layer == 3 && getenv("BX_BALL_DEBUG")) ? c->ball_dbg : nullptr);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
}  // namespace

// Desc (Cylindrical_Net) layer `layer` on the 32x32x2 f32 MFMA; same contract as bxk_conv(net = 0).  Returns -1 when the call is
// not served here (the caller then uses the 16x16x4 kernel): a device-side unit count, fewer than 4 units, and layers 6 and 7
// (32 output channels = ONE column tile: four compute waves cannot split 9 or 18 row tiles evenly; the last layer has no ReLU).
int bxk_conv32(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (units_dev || max_units < 4 || layer >= 6) return -1;
    // the group walk draws its tickets from ONE counter pair per (context, layer): the latency form runs the source and the target
    // chain of a pair concurrently on two streams, whose launches of a layer would share it -- the stateless 16x16x4 kernels serve it
    if (c->p.keypoint_tiles > 1) return -1;
    const ConvLayerDev& L = c->desc[layer];
    switch (layer) {
        //                      NCHUNK COUT G  RELU
        case 0: return launch32<3, 64, 4, true>(c, layer, s, L, in, max_units, out);
        case 1: return launch32<4, 64, 4, true>(c, layer, s, L, in, max_units, out);
        case 2: return launch32<4, 128, 2, true>(c, layer, s, L, in, max_units, out);
        case 3: return launch32<8, 128, 2, true>(c, layer, s, L, in, max_units, out);
        case 4: return launch32<8, 64, 4, true>(c, layer, s, L, in, max_units, out);
        case 5: return launch32<4, 64, 4, true>(c, layer, s, L, in, max_units, out);
    }
    return -1;
}
