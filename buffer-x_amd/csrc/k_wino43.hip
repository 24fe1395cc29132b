// k_wino43.hip -- Cylindrical_Net layers as Winograd F(4x4, 3x3) convolutions on the f32 matrix cores (the default form: bx_params.desc_conv_form).
//
// Reference: models/patchnet.py:49-84 (the eight Conv2d + BatchNorm + ReLU blocks), padding utils/common.py:265-310 (azimuth wraps,
// elevation zero-pads).  The 7 x 20 map is cut into 2 x 5 output tiles of 4 x 4 (the 8th output row does not exist); per tile and channel
// the 6 x 6 input window d becomes V = B^T d B, the channel contraction is THIRTY-SIX independent GEMMs M[xi][nu] = V[xi][nu] U[xi][nu]
// (rows = tiles, K = input channels, columns = output channels) on v_mfma_f32_16x16x4_f32, and Y = A^T M A (+ bias, ReLU) folds them
// into the 16 outputs: 36 multiplications per 16 outputs against 9 per output of the direct form -- 10 tile rows x 36 planes per unit =
// 0.29x the direct form's MFMA work.  B^T holds 4, -5, 2, G 1/4, 1/6, 1/24, A^T up to 8: the error against a binary64 convolution is
// ~2.4x the direct fp32 form's (rms; tests/study_wino43_error.py), still fp32-grade; the parity sweep (tests/test_gpu_sweep.py) and the
// reference-minted fixtures are the judge.  Arithmetic contract: oracle/bx_oracle.c::bxo_conv_wino43; GPU == oracle bit for bit.
//
// Round-4 kernel (profiles/r04_wino43_variants.txt has the measurements behind every choice):
//  * workgroup = 8 waves = CW output channels (64; 32 for the two 32-channel layers) of one ITEM = 32 consecutive tile rows of the
//    layer's (unit, tile) sequence = two full MFMA row tiles (round 5; rounds 3-4: three whole units = 30 of the 32 rows -- see the
//    constants below: 6.7 % fewer items, +1.6 % pairs/s; the MFMAs no longer issued on two padding rows also lower the ISSUED-flop
//    roofline fraction by 6 % at equal time: bench.py prices 360 instead of 384 plane-rows per unit);
//    compute wave (column tile ct, half) owns the eighteen planes of the xi rows 3 half .. 3 half + 2: 36 accumulator tiles = 144 VGPRs.
//    With CW = 32 only waves 0..3 (one per SIMD) stream MFMAs; the others still stage and transform.
//  * slab: the (unit, chunk) maps arrive as 16-byte pieces (up to five per thread, non-temporal) in the LDS slab (four unit slots of 9 x 22
//    positions with the wrap-around columns copied and the rows beyond the map left zero).  Loads and the weight ring share ONE in-order counter
//    (vmcnt), so the pieces of the next chunk are requested right after the chunk's first barrier -- a transform and a whole MFMA phase
//    before they are needed, never between two ring loads -- and written to the slab all at once behind the last plane (the write's
//    wait for its piece is a wait for every load in flight: one such drain per chunk, under the tail of the MFMAs).
//  * transform: lane (tile row tR = 4 wave + lane / 16 of the item, channel slot lane % 16) -- every one of the 512 threads owns one
//    (tile row, channel) of the chunk: 36 ds_read_b32 of its 6 x 6 window from the slab (row pitch 22 x 20 + 4 floats), B^T d down
//    the six columns and along the six rows on scalar f32 (bt6s: 144 + 19 VALU), 36 ds_write_b32 into the V planes (round 3 re-read
//    the window once per xi: 253 KB of slab reads per chunk instead of 69 KB).  A channel-PAIR form (ds_read_b64, 72 v_pk_*_f32,
//    ds_write_b64, three xi rows per wave) was built in round 5, is bit-exact and measured 2.6 % slower
//    (profiles/r05_wino43_variants.txt): it is NOT in the library; its helpers were removed from wino43_common.h in round 6.
//  * MFMA phase: SWAPPED operands (A = weight fragment, B = V rows), so accumulator register r of lane (li, kk) is
//    M[tile row rt * 16 + li][slot 4 kk + r]: a lane owns four contiguous output slots of one tile.  B fragments in a ring of three
//    planes through raw buffer loads (descriptor + wave-uniform offset in SGPRs: no VALU address arithmetic between the MFMAs).
//  * output: nu pass and the half's partial xi sums lane-local; half 0 finishes the output rows 0, 1 of every tile, half 1 the rows 2, 3:
//    each sends the two partials the other needs as float4 through a lane-linear LDS exchange (the bytes of the V planes) and stores
//    Y = (P_0 + P_1) + bias as 16-byte pieces: 16 ds_write_b128 + 16 ds_read_b128 + 16 stores per lane and group, four barriers
//    (round 3: 128 ds_write_b32 + 32 ds_read_b128, eight barriers).  Round 5: the lane-local arithmetic runs on v_pk_*_f32 over the
//    register pairs (r, r + 1) of the MFMA results (no shuffles: wino43_common.h), a half keeps only the two sums its own rows need
//    across the exchange, the stores are raw buffer stores (32-bit lane offset + SGPR group offset), the slab pieces raw buffer loads,
//    and the phase's lane constants + the bias (LDS) are re-derived per group instead of living -- spilled -- across the MFMA pipeline:
//    2 spilled VGPRs instead of 8-10, stack -1.3 % (profiles/r05_wino43_variants.txt).
// Measured and NOT kept: the window straight from global memory (36 dword loads per thread: the texture addresser needs ~16 cycles per
// wave-instruction of four 64-byte segments -- 1 800-4 700 cycles of blocked issue per chunk), a start stagger of the workgroups (no
// change: the output phase is not a chip-wide burst), temporal output stores (no change).
#include "wino43_common.h"
#include <cstdlib>
#include <vector>

#ifndef BX_W43_EARLYREQ
#define BX_W43_EARLYREQ 1          // slab pieces of the next chunk requested BEFORE the transform and written late in the MFMA loop (0: the round-4 first form)
#endif
#ifndef BX_W43_SWAPXY
#define BX_W43_SWAPXY 0            // 1: 128-column layers with the column block as the fast grid dimension (XCD parity = column block): measured +-0 (r06h)
#endif
#ifndef BX_W43_STAMP
#define BX_W43_STAMP 0             // 1: instrumented build (tools/build_variant.sh): s_memtime phase stamps of workgroup (0, 0) into bx_debug_read
#endif

namespace {
using namespace w43;
constexpr int WP = BX_AZI + 2;                   // slab columns (wrap-around halo)
constexpr int HP = BX_ELE + 3;                   // slab rows h = -1 .. 8 (tile row 1 reaches two rows below the map)
constexpr int TR4 = (BX_ELE + 3) / 4, TC4 = BX_AZI / 4, NT4 = TR4 * TC4;   // 2 x 5 = 10 tiles
// Round 5: a workgroup ITEM is 32 consecutive tile rows of the layer's (unit, tile) sequence -- both MFMA row tiles full -- instead of three
// whole units (30 of the 32 rows): 6.7 % fewer items for the same work per item.  32 tile rows start at an even tile of a unit (32 = 2 mod
// 10) and touch at most FOUR units, so the slab has four unit slots; a slot keeps 9 of the 10 window rows (h = -1 .. 7: its tenth, the second
// zero row below the map, is the next slot's zero row h = -1, and one spare zero row follows the last slot), which is what lets four slots
// fit beside the V planes.  A unit that straddles two items is staged by both (input traffic x 1.25 on average; the layers sit at ~20 %
// of the HBM rate).
constexpr int G4 = 4, ROWS4 = VR4;               // unit slots of the slab; tile rows of an item = 32
constexpr int HPU = HP - 1;                      // rows a slot owns
constexpr int RP3 = WP * ROWF + 4, UP3 = HPU * RP3;                        // slab row / slot pitch in floats: 444, 3 996
constexpr int SLABF = G4 * UP3 + RP3;            // + the spare zero row
constexpr size_t W43_LDS = (size_t)(SLABF + NPL * VPL4 + 64) * 4;          // 65 712 + 92 160 B + the workgroup's 64 bias values
static_assert(NPH % 3 == 0 && BX_AZI % 4 == 0 && W43_LDS <= 160 * 1024 && (RP3 * 4) % 16 == 0 && 8 * 8 * 64 * 16 <= NPL * VPL4 * 4,
              "geometry, LDS, 16-byte slab rows, output exchange inside the V planes");

// ---- output transform.  nu pass and the half's partial xi sums lane-local (wino43_send), one exchange round per MFMA row tile.
template <int NT, bool RELU>
__device__ __forceinline__ void wino43_output(const f32x4 (&acc)[NPH][RT4], float* Vp, bool cw, int half, int wave, int lane, int u0, int ioff, int units,
                                              int ctile, const __amdgpu_buffer_rsrc_t ors)
{
    float4* ex = reinterpret_cast<float4*>(Vp);             // [wave][8][lane]
    // the lane constants of this phase (exchange addresses, tile coordinates, bias) are derived HERE from the lane id, every group: the
    // empty asm keeps hipcc from hoisting them out of the group loop, where they were live (in fact: spilled to scratch and reloaded
    // -- a reload waits behind every load and store in flight) across the whole MFMA pipeline for ~20 integer instructions per group;
    // the bias comes from LDS for the same reason
    asm volatile("" : "+v"(lane));
    const float4 b4 = *reinterpret_cast<const float4*>(Vp + NPL * VPL4 + (wave >> 1) * 16 + (lane >> 4) * 4);
    const int li = lane & 15, kk = lane >> 4;
    float4* mine = ex + (wave * 8) * 64 + lane;
    const float4* theirs = ex + ((wave ^ 1) * 8) * 64 + lane;
#pragma unroll
    for (int rt = 0; rt < RT4; ++rt) {
        f32x2 A[2][4], B[2][4];                 // [register pair][j]
        if (cw) {
            if (half == 0) wino43_send<0>(acc, rt, A, B, mine);
            else wino43_send<1>(acc, rt, A, B, mine);
        }
        __syncthreads();
        if (cw) {
            const int R = ioff + rt * 16 + li;          // tile row inside the item's unit window
            const int g = R / NT4, t = R - g * NT4, tr = t / TC4, tc = t - tr * TC4;
            const int u = u0 + g;
            const bool live = u < units;
            const int i0 = 2 * half;                        // this half's output rows of a tile: i0, i0 + 1
            const bool second_row = 4 * tr + i0 + 1 < BX_ELE;
            // raw buffer stores: ONE 32-bit offset per lane and row tile -- no 64-bit address registers held (or spilled) across the
            // kernel, no 64-bit VALU address arithmetic in the output phase.
            // HAZARD (found the hard way, round 5): the group's offset must NOT go into the instruction's SGPR offset field.  hipcc 7.2
            // inserts no wait state between a buffer_store_dwordx4 WITH an SGPR offset and a VALU instruction that overwrites the
            // store's data registers (LLVM's rule: "the >64-bit store-data hazard only exists without an soffset register"), and on
            // gfx950 the store then reads half-overwritten data: wino43v_kernel<6, 3, 64, 18, 2> stored wrong values for four lanes of one
            // output column (tests/test_gpu_stages.py::test_pose_conv_layer_exact[1]).  With soffset = 0 the compiler sees the hazard
            // and spaces the instructions; the group offset costs one v_add per row tile.
            const int voff = (((u * NT + ctile) * BX_EA + (4 * tr + i0) * BX_AZI + 4 * tc) * 16 + 4 * kk) * 4;
            auto store = [&](int off, const f32x4 v) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, voff + off * 4, 0, 2 /* nt: streamed once */);
            };
            if (half == 0) wino43_finish<0, RELU>(A, B, theirs, b4, store, live, second_row, BX_AZI * 16, 15u);
            else wino43_finish<1, RELU>(A, B, theirs, b4, store, live, second_row, BX_AZI * 16, 15u);
        }
        __syncthreads();                                    // the exchange is free again (next row tile / next group's V planes)
    }
}

template <int NCHUNK, int COUT, int CW, bool RELU>
__global__ __launch_bounds__(CT, 2) void wino43_kernel(const float* __restrict__ in, int units, const float* __restrict__ U,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       const int32_t* __restrict__ skip, long long* __restrict__ dbg)
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
    // block = (item walker bxi, column block byi).  With two column blocks per layer (COUT = 128) the COLUMN block is the fast grid dimension:
    // workgroup -> XCD placement follows the linear block id modulo 8, so even XCDs then hold column block 0 and odd XCDs column block 1,
    // and an XCD's L2 keeps HALF of the layer's U fragments (1.2 instead of 2.4 MB for the 128 -> 128 layer) beside the streamed maps
    constexpr bool SWAPXY = BX_W43_SWAPXY && (COUT / CW == 2);
    const int bxi = SWAPXY ? (int)blockIdx.y : (int)blockIdx.x, byi = SWAPXY ? (int)blockIdx.x : (int)blockIdx.y;
    const int ctg = byi * NCW + (cw ? ctl : 0);
    const int li = lane & 15, kk = lane >> 4;
    const int ngroups = (units * NT4 + ROWS4 - 1) / ROWS4;          // items of 32 tile rows
    if (bxi >= ngroups) return;

    for (int i = tid; i < (int)(W43_LDS / 16); i += CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- slab traffic: per-thread constants (piece inside the group's [3][NCHUNK][140][16] floats, destination, halo copy)
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
            ldst[q] = g * UP3 + (h + 1) * RP3 + (w + 1) * ROWF + part * 4;
            lhalo[q] = w == 0 ? BX_AZI * ROWF : (w == BX_AZI - 1 ? -BX_AZI * ROWF : 0);
        }
    }
    // raw buffer loads: lane offset lsrc * 16 (a lane constant) + the (group, chunk) offset in an SGPR; pieces of units that do not exist
    // read as zeros (explicit predicate: the hardware range check does not see the SGPR offset)
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((long long)units * NCHUNK * NPU * 16 < 0x7fffffffLL ? (long long)units * NCHUNK * NPU * 16 : 0x7fffffffLL), 0x00020000);
    auto gload1 = [&](int q, int ug_, int cc_) {   // streamed once: non-temporal, so that the activations do not push the B fragments out of L2
        const int u0_ = (ug_ * ROWS4) / NT4;       // first unit of the item's window (slot 0)
        const int soff = ((u0_ * NCHUNK + cc_) * NPU) * 16;
        const int lim = (units - u0_) * NCHUNK * NPU;
        const f32x4 v = (lsrc[q] >= 0 && lsrc[q] < lim) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, lsrc[q] * 16, soff, 2 /* nt */)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        st[q] = make_float4(v.x, v.y, v.z, v.w);
    };
    auto lwrite1 = [&](int q) {
        if (lsrc[q] >= 0) {
            float* d = slab + ldst[q];
            *reinterpret_cast<float4*>(d) = st[q];
            if (lhalo[q] != 0) *reinterpret_cast<float4*>(d + lhalo[q]) = st[q];
        }
    };

    // ---- transform role: (tile row tR = 4 wave + lane / 16 of the item, channel slot lane % 16): all 512 threads have a tile row
    const int tR = 4 * wave + (lane >> 4);
    float* vdst = Vp + tR * ROWF + (lane & 15);
    const float* wsrc = slab;                       // window origin of this thread's tile: set per item (the unit window moves)
    auto set_window = [&](int ioff_) {
        const int gr = ioff_ + tR;
        const int tg = gr / NT4, tt = gr - tg * NT4, ttr = tt / TC4, ttc = tt - ttr * TC4;
        wsrc = slab + tg * UP3 + (4 * ttr) * RP3 + (4 * ttc) * ROWF + (lane & 15);
    };
    auto transform = [&]() {
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
            bt6s(t[x][0], t[x][1], t[x][2], t[x][3], t[x][4], t[x][5], o);      // along the row: the six planes (xi, 0..5)
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) vdst[(x * 6 + nu) * VPL4] = o[nu];
        }
    };

    // (the bias of the workgroup's column tiles lives in LDS as [column tile][kk][r] = slot kk + 4 r, behind the V planes: wino43_output)
    // B fragments [chunk * 36 + plane][column tile][lane][4]; this wave's planes are half * 18 + 0..17.  Raw buffer loads: descriptor +
    // wave-uniform byte offset in SGPRs (SALU arithmetic), one 32-bit lane offset -- no VALU address arithmetic between the MFMAs
    // (with global_load hipcc rebuilt a 64-bit VGPR address per plane: 30 VALU instructions inside every chunk's MFMA stream)
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, NCHUNK * NPL * NT * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long long)units * NT * BX_EA * 64 < 0x7fffffffLL ? (long long)units * NT * BX_EA * 64 : 0x7fffffffLL), 0x00020000);
    const int ubase = ((half * NPH) * NT + ctg) * 1024;
    const int ulane = lane * 16;
    auto bload = [&](int q) {
        // (whole-vector bit cast: hipcc 7.2 narrows the load to ONE dword when the components of the u32x4 result are bit-cast one by one)
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, ubase + q * (NT * 1024), 0));
        return make_float4(v.x, v.y, v.z, v.w);
    };
    const char* abase = reinterpret_cast<const char*>(Vp) + ((half * NPH * VR4 + li) * ROWF + kk * 4) * 4;

    f32x4 acc[NPH][RT4];
    // B fragments in a ring of THREE planes (18 planes per wave: the ring size must divide it so that plane q of the next chunk lands in
    // the slot plane q is read from): plane p + 3 is requested right before the MFMAs of plane p
    float4 bring[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bring[p] = bload(p);

    // (group, chunk) walk of the slab pipeline: the slab holds the chunk being transformed, st the pieces of the chunk after it,
    // requests go out for the one after that
    int ug = bxi;
    const int gstep = SWAPXY ? (int)gridDim.y : (int)gridDim.x;
    int lg = ug, lc = 0;                            // the (group, chunk) the NEXT request fetches
    auto ladv = [&]() { if (++lc == NCHUNK) { lc = 0; lg += gstep; } };
#pragma unroll
    for (int q = 0; q < NLD; ++q) gload1(q, lg, lc);
    ladv();
    __syncthreads();                 // zero fill complete
    if (tid < CW) Vp[NPL * VPL4 + (tid >> 4) * 16 + (tid & 3) * 4 + ((tid & 15) >> 2)] = bias[byi * CW + tid];
#pragma unroll
    for (int q = 0; q < NLD; ++q) lwrite1(q);
    bool st_live = lg < ngroups;
#if !BX_W43_EARLYREQ
    if (st_live) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) gload1(q, lg, lc);
        ladv();
    }
#endif
#if BX_W43_STAMP
    // phase stamps (instrumented build only), cycles summed over the kernel: 0 MFMA phase (incl. the slab traffic inside it), 2 barrier A,
    // 3 transform + V stores, 4 barrier B, 5 (unused), 6 = groups, 7 output transform
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
    int st_in = 0;
#define BX_STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); if ((k) == 6) { st_acc[0] += t_ - st_last; st_acc[6] += 1; } else if ((k) == 0) { if (st_in) st_acc[0] += t_ - st_last; st_in = 1; } else st_acc[k] += t_ - st_last; if ((k) == 7) st_in = 0; st_last = t_; } while (0)
#else
#define BX_STAMP(k) do { } while (0)
#endif

    for (;;) {
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int rt = 0; rt < RT4; ++rt) acc[p][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ugn = ug + gstep;
        const int u0 = (ug * ROWS4) / NT4, ioff = ug * ROWS4 - u0 * NT4;     // first unit of the item, tile offset of its first row in it
        set_window(ioff);
#pragma unroll 1
        for (int cc = 0; cc < NCHUNK; ++cc) {
            BX_STAMP(0);
            __syncthreads();         // the slab of this chunk is complete; every wave is done with the V planes of the chunk before
            BX_STAMP(2);
#if BX_W43_EARLYREQ
            // the pieces of the NEXT chunk are requested here, a transform and a whole MFMA phase before they are written to the slab: loads
            // and the weight ring share one in-order counter, so a request issued between ring loads (the first form: planes 2, 4, 6, 8)
            // held the plane three ahead until the HBM access returned
            st_live = lg < ngroups;
            const bool st_was = st_live;
            if (st_live) {
#pragma unroll
                for (int q = 0; q < NLD; ++q) gload1(q, lg, lc);
                ladv();
            }
            __builtin_amdgcn_sched_barrier(0);
            const int lgq = 0, lcq = 0;
            (void)lgq; (void)lcq;
#endif
            transform();
            BX_STAMP(3);
            __syncthreads();         // V complete; the slab is free
            BX_STAMP(4);
#if !BX_W43_EARLYREQ
            const bool st_was = st_live;
            st_live = lg < ngroups;
            const int lgq = lg, lcq = lc;
            if (st_live) ladv();
#endif
            if (cw) {
                const int cn = cc + 1 == NCHUNK ? 0 : cc + 1;
                f32x4 ar[3];
                ar[0] = *reinterpret_cast<const f32x4*>(abase);
                ar[1] = *reinterpret_cast<const f32x4*>(abase + (16 * ROWF) * 4);
#pragma unroll
                for (int p = 0; p < NPH; ++p) {
                    const float4 bqq = bring[p % 3];
                    // the slot is refilled BEFORE the plane's MFMAs (they read the copy): four planes of look-ahead from a ring of three
                    bring[p % 3] = p + 3 < NPH ? bload(cc * NPL + p + 3) : bload(cn * NPL + p + 3 - NPH);
                    if (NCW == 2) {
                        // ONE compute wave per SIMD: four back-to-back MFMAs on one accumulator would issue at the 40-cycle dependent
                        // latency instead of every 32 cycles (with two waves per SIMD the sibling fills the gap), so the two row tiles
                        // of the plane alternate
#if !BX_W43_EARLYREQ
                        if (p >= 1 && p < 1 + 2 * NLD) {
                            const int q = (p - 1) >> 1;
                            if (((p - 1) & 1) == 0) { if (st_was) lwrite1(q); }
                            else if (st_live) gload1(q, lgq, lcq);
                        }
#endif
                        const f32x4 a0 = ar[0], a1 = ar[1];
                        if (p + 1 < NPH) {
                            ar[0] = *reinterpret_cast<const f32x4*>(abase + (((p + 1) * VR4) * ROWF) * 4);
                            ar[1] = *reinterpret_cast<const f32x4*>(abase + (((p + 1) * VR4 + 16) * ROWF) * 4);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.x, a0.x, acc[p][0], 0, 0, 0);
                        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.x, a1.x, acc[p][1], 0, 0, 0);
                        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.y, a0.y, acc[p][0], 0, 0, 0);
                        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.y, a1.y, acc[p][1], 0, 0, 0);
                        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.z, a0.z, acc[p][0], 0, 0, 0);
                        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.z, a1.z, acc[p][1], 0, 0, 0);
                        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.w, a0.w, acc[p][0], 0, 0, 0);
                        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqq.w, a1.w, acc[p][1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    // the slab is free during the MFMA phase: piece q goes to the slab behind plane 2 q + 1 (requested a whole chunk ago),
                    // its register is re-requested behind plane 2 q + 2
#if !BX_W43_EARLYREQ
                    if (p >= 1 && p < 1 + 2 * NLD) {
                        const int q = (p - 1) >> 1;
                        if (((p - 1) & 1) == 0) { if (st_was) lwrite1(q); }
                        else if (st_live) gload1(q, lgq, lcq);
                    }
#endif
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
#if BX_W43_EARLYREQ
                // the slab writes of the next chunk, all at once behind the last plane: a slab write must know its piece has arrived, and
                // with one in-order counter for every load that is a wait for the ring loads in flight too -- one such wait here, under the
                // tail of the MFMAs, instead of one per piece inside the loop
                if (st_was) {
#pragma unroll
                    for (int q = 0; q < NLD; ++q) lwrite1(q);
                }
#endif
            } else {                 // CW = 32: the waves without a column tile carry the slab traffic only
#pragma unroll
                for (int q = 0; q < NLD; ++q) {
                    if (st_was) lwrite1(q);
#if !BX_W43_EARLYREQ
                    if (st_live) gload1(q, lgq, lcq);
#endif
                }
            }
        }
        BX_STAMP(6);
        __syncthreads();             // every wave is done with the V planes: their bytes carry the exchange now
        wino43_output<NT, RELU>(acc, Vp, cw, half, wave, lane, u0, ioff, units, ctg, ors);
        BX_STAMP(7);
        ug = ugn;
        if (ug >= ngroups) break;
    }
#if BX_W43_STAMP
    if (dbg && bxi == 0 && byi == 0 && lane == 0 && (wave == 0 || wave == 3)) {
        long long* d = dbg + (wave ? 8 : 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = (long long)st_acc[i];
    }
#endif
}
#undef BX_STAMP

template <int NCHUNK, int COUT, int CW, bool RELU>
int launch_wino43(bx_ctx* c, int layer, hipStream_t s, const ConvLayerDev& L, const float* in, int units, float* out)
{
    if (L.nchunk != NCHUNK || L.cout != COUT || (L.relu != 0) != RELU || !L.Wwino43) {
        bx_set_error("winograd F(4x4) layer %d: geometry mismatch (%d chunks, %d channels)", layer, L.nchunk, L.cout);
        return BX_ERR_STATE;
    }
    // the kernel addresses its input and output through raw buffer resources with 32-bit byte offsets: a map set of 2 GiB or more
    // (~30 k units on the 128-channel layers) is "not served" -- the direct kernels (64-bit addressing) take it
    if (!w43::fits_i32((long long)units * NCHUNK * BX_EA * 64) || !w43::fits_i32((long long)units * (COUT / 16) * BX_EA * 64)) return -1;
    auto k = wino43_kernel<NCHUNK, COUT, CW, RELU>;
    int& cap = c->wino_cap[layer];
    if (cap == 0) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W43_LDS));
        cap = c->n_cu / (COUT / CW);
        if (cap < 1) cap = 1;
        if (c->conv_cap_override > 0 && c->conv_cap_override < cap) cap = c->conv_cap_override;
    }
    int grid = (units * NT4 + ROWS4 - 1) / ROWS4;
    if (grid <= 0) return BX_OK;
    if (grid > cap) grid = cap;
    const bool swapxy = BX_W43_SWAPXY && COUT / CW == 2;
    hipLaunchKernelGGL(k, swapxy ? dim3(COUT / CW, grid) : dim3(grid, COUT / CW), dim3(CT), W43_LDS, s, in, units, L.Wwino43, L.b, out, c->skip,
                       BX_W43_STAMP ? reinterpret_cast<long long*>(c->ball_dbg) + 16 * layer : nullptr);
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

// U = G g G^T (F(4x4, 3x3)) of every (chunk, channel, output channel) in binary64, rounded once, packed as MFMA fragments
// [chunk * 36 + plane][column tile][lane = kk*16 + li][4], element i = U[plane][chunk][kk + 4 i][channel of slot li]
int bxk_wino43_weights(const float* w /* [nchunk][9 * fold][16][cout] */, int nchunk, int fold, int cout, float** d_out)
{
    // fold = 3: CostNet layer 1, whose three k rows become input channels (tap = (a * 3 + b) * 3 + d; effective chunk e = chunk * 3 + b)
    const int nt = cout / 16, ntaps = 9 * fold;
    std::vector<float> frag((size_t)nchunk * fold * NPL * nt * 64 * 4, 0.0f);
    for (int cc = 0; cc < nchunk; ++cc)
      for (int b = 0; b < fold; ++b)
        for (int ch = 0; ch < 16; ++ch)
            for (int o = 0; o < cout; ++o) {
                double g[3][3], Gg[6][3];
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw)
                        g[kh][kw] = (double)w[(((size_t)cc * ntaps + (fold == 3 ? (kh * 3 + b) * 3 + kw : kh * 3 + kw)) * 16 + ch) * cout + o];
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
                        // column of the fragment = the channel's OUTPUT SLOT: the fragment is the A operand of the MFMA, so accumulator rows
                        // (4 kk + r) are contiguous slots of the output map
                        const int pl = xi * 6 + nu, kk = ch & 3, i = ch >> 2, t = o / 16, li = bx_chunk_slot(o % 16);
                        frag[((((size_t)((cc * fold + b) * NPL + pl) * nt + t) * 4 + kk) * 16 + li) * 4 + i] = (float)u6[nu];
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
        case 0: return launch_wino43<3, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 1: return launch_wino43<4, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 2: return launch_wino43<4, 128, 64, true>(c, layer, s, L, in, max_units, out);
        case 3: return launch_wino43<8, 128, 64, true>(c, layer, s, L, in, max_units, out);
        case 4: return launch_wino43<8, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 5: return launch_wino43<4, 64, 64, true>(c, layer, s, L, in, max_units, out);
        case 6: return launch_wino43<4, 32, 32, true>(c, layer, s, L, in, max_units, out);
        case 7: return launch_wino43<2, 32, 32, false>(c, layer, s, L, in, max_units, out);
    }
    return -1;
}
