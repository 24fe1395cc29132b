// wino43_common.h -- pieces shared by the two Winograd F(4x4, 3x3) kernels (k_wino43.hip: Cylindrical_Net, circular / zero-padded 7 x 20
// maps; k_wino43v.hip: CostNet layers 1..5, valid D x D maps): the input transform of a 6-vector and the two halves of the output
// transform of the swapped-operand form.  Arithmetic contract: oracle/bx_oracle.c::bxo_conv_wino43 / bxo_conv_wino43_valid.
#pragma once
#include "bx_common.h"

namespace w43 {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWF = 20;                         // floats per LDS row (16 + 4 pad)
constexpr int RT4 = 2, VR4 = RT4 * 16, VPL4 = VR4 * ROWF;   // two MFMA row tiles = 32 tile rows per workgroup item
constexpr int NPL = 36, NPH = 18;                // planes, planes per wave half
constexpr int CT = 512;
// the F(4x4) kernels address maps with 32-bit byte offsets (lane offset + SGPR group offset, num_records clamped to 2^31 - 1): the
// launchers refuse -- return -1, "not served" -- anything that does not fit, with a margin for the largest in-item offset
inline bool fits_i32(long long bytes) { return bytes < 0x7fffffffLL - (1LL << 24); }

// ---- packed f32 (round 5).  The vector ALU and the f32 MFMA are ONE resource on this chip (profiles/r05_mfma_coissue.txt: a VALU
// instruction costs its 4-5 issue cycles whichever wave issues it), so what the transforms cost is their instruction count, and a
// v_pk_*_f32 does two lanes' worth for the price of one -- but only when its operands already sit in aligned register pairs (hipcc's own
// SLP packing paid for the pairs with v_mov and gained nothing).  The OUTPUT transform is written on explicit two-element vectors whose
// pairs are free: two accumulator registers r, r + 1 of one MFMA result.  (The input transform stays scalar: its channel-pair form
// measured slower, profiles/r05_wino43_variants.txt.)  Element by element the expressions are those of the contract (oracle/bx_oracle.c::bxo_conv_wino43); a - b is written
// fma(-1, b, a) -- the same single rounding, signed zeros included -- because hipcc scalarises a packed subtraction.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk2(float k) { return (f32x2){k, k}; }
__device__ __forceinline__ f32x2 pfma(float k, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(pk2(k), b, c); }
__device__ __forceinline__ f32x2 psub(f32x2 a, f32x2 b) { return __builtin_elementwise_fma(pk2(-1.0f), b, a); }
__device__ __forceinline__ f32x2 lo2(const f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(const f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f32x4 cat2(const f32x2 a, const f32x2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }

// the six results of B^T on a 6-vector (contract: bxo_conv_wino43; t3 / t4 as fmaf(+-2, d3 - d1, c): 2 x is exact, so the rounding is
// that of c +- e): the INPUT transform of both kernels, scalar f32
__device__ __forceinline__ void bt6s(float d0, float d1, float d2, float d3, float d4, float d5, float (&o)[6])
{
    o[0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    const float a = fmaf(-4.0f, d2, d4), b = fmaf(-4.0f, d1, d3);
    o[1] = a + b;
    o[2] = a - b;
    const float c = d4 - d2, s = d3 - d1;
    o[3] = fmaf(2.0f, s, c);
    o[4] = fmaf(-2.0f, s, c);
    o[5] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
}

// ---- output transform, lane-local part.  Accumulator register r of a lane = output slot 4 kk + r.  nu pass (A^T along a row's six
// planes) and the half's partial xi sums; a half SENDS the two partials the other half's output rows need and KEEPS only the two sums
// its own rows need (A = P[0] of its first row, B = P[.] of its second): 32 live registers across the exchange barrier instead of the
// 48 of the round-4 form (ua, ub, uc kept whole), which is what pushed the kernel over its 256 VGPRs in the output phase -- and every
// spilled register is a scratch reload per group that waits behind all loads and stores in flight.
//   half 0 (xi 0..2, output rows 0, 1):  keeps A = r_0 + (r_1 + r_2), B = r_1 - r_2;       sends r_1 + r_2, r_1 - r_2
//   half 1 (xi 3..5, output rows 2, 3):  keeps A = 4 (r_3 + r_4), B = fma(8, r_3 - r_4, r_5); sends r_3 + r_4, 2 (r_3 - r_4)
// packed form: the pairs (0, 1), (2, 3) of an MFMA result are aligned register pairs, so everything runs on v_pk_*_f32 without shuffles
template <int HALF>
__device__ __forceinline__ void wino43_send(const f32x4 (&acc)[NPH][RT4], int rt, f32x2 (&A)[2][4], f32x2 (&B)[2][4], float4* mine)
{
    f32x4* mine4 = reinterpret_cast<f32x4*>(mine);
    f32x2 s0[2][4], s1[2][4];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        f32x2 rr[3][4];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            f32x2 m[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) m[k] = rp == 0 ? lo2(acc[x * 6 + k][rt]) : hi2(acc[x * 6 + k][rt]);
            const f32x2 p = m[1] + m[2], q = psub(m[1], m[2]), s = m[3] + m[4], t = psub(m[3], m[4]);
            rr[x][0] = (m[0] + p) + s;
            rr[x][1] = pfma(2.0f, t, q);
            rr[x][2] = pfma(4.0f, s, p);
            rr[x][3] = pfma(8.0f, t, q) + m[5];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (HALF == 0) {
                const f32x2 ub = rr[1][j] + rr[2][j], uc = psub(rr[1][j], rr[2][j]);
                A[rp][j] = rr[0][j] + ub; B[rp][j] = uc;
                s0[rp][j] = ub; s1[rp][j] = uc;
            } else {
                const f32x2 ua = rr[0][j] + rr[1][j], ub = psub(rr[0][j], rr[1][j]);
                A[rp][j] = pk2(4.0f) * ua; B[rp][j] = pfma(8.0f, ub, rr[2][j]);
                s0[rp][j] = ua; s1[rp][j] = pk2(2.0f) * ub;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mine4[(2 * j) * 64] = cat2(s0[0][j], s0[1][j]);
        mine4[(2 * j + 1) * 64] = cat2(s1[0][j], s1[1][j]);
    }
}

template <int HALF, bool RELU, class ST>
__device__ __forceinline__ void wino43_finish(const f32x2 (&A)[2][4], const f32x2 (&B)[2][4], const float4* theirs,
                                              const float4 b4, ST&& store, bool live, bool second_row, int row_stride, unsigned jmask)
{
    const f32x2 ba[2] = {(f32x2){b4.x, b4.y}, (f32x2){b4.z, b4.w}};
    const f32x4* theirs4 = reinterpret_cast<const f32x4*>(theirs);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 g0 = theirs4[(2 * j) * 64], g1 = theirs4[(2 * j + 1) * 64];
        f32x2 y0[2], y1[2];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const f32x2 g0p = rp == 0 ? lo2(g0) : hi2(g0), g1p = rp == 0 ? lo2(g1) : hi2(g1);
            // Y = (P_0 + P_1) + bias: half 0 holds P_0 and receives P_1, half 1 the other way round (the sum commutes bit for bit)
            y0[rp] = (HALF == 0 ? A[rp][j] + g0p : g0p + A[rp][j]) + ba[rp];
            y1[rp] = (HALF == 0 ? B[rp][j] + g1p : g1p + B[rp][j]) + ba[rp];
            if (RELU) {                                     // no packed f32 maximum on gfx950: v_max_f32 per element
                y0[rp] = (f32x2){y0[rp].x > 0.f ? y0[rp].x : 0.f, y0[rp].y > 0.f ? y0[rp].y : 0.f};
                y1[rp] = (f32x2){y1[rp].x > 0.f ? y1[rp].x : 0.f, y1[rp].y > 0.f ? y1[rp].y : 0.f};
            }
        }
        if (live && ((jmask >> j) & 1u)) {                  // jmask: output columns of the tile that exist (valid maps: the last tile column)
            store(j * 16, cat2(y0[0], y0[1]));
            if (second_row) store(row_stride + j * 16, cat2(y1[0], y1[1]));
        }
    }
}

// ---- mixed tiles (k_wino43m.hip, bx_params.desc_conv_form = winograd43m): the F(3x4, 3x3) tiles of the map rows 4..6.  F(3, 3) on the
// points {1, -1, 1/2, -1/2, inf} down the rows (contract: oracle/bx_oracle.c::bxo_conv_wino43m): B3^T of a 5-vector, and the two halves of
// the output transform.  Wave half 0 holds the planes xi = 0..2, half 1 the planes xi = 3, 4:
//   half 0: p = r_0 + r_1, q = r_0 - r_1: keeps A = p + r_2 (output row 0), B = fma(.5, r_2, q) (row 1); sends fma(.25, r_2, p) (row 2)
//   half 1: keeps A = fma(.25, r_3, r_4) (row 2); sends r_3 (row 0) and -.5 r_3 (row 1)
__device__ __forceinline__ void bt5s(float d0, float d1, float d2, float d3, float d4, float (&o)[5])
{
    o[0] = fmaf(4.0f, d2 + d3, -(d0 + d1));
    o[1] = fmaf(4.0f, d3 - d2, d0 - d1);
    const float a = d2 - d0, b = d3 - d1;
    o[2] = fmaf(2.0f, b, a);
    o[3] = fmaf(2.0f, b, -a);
    o[4] = fmaf(4.0f, d4, fmaf(-5.0f, d2, d0));
}

template <int HALF>
__device__ __forceinline__ void wino43m_send(const f32x4 (&acc)[NPH][RT4], int rt, f32x2 (&A)[2][4], f32x2 (&B)[2][4], float4* mine)
{
    f32x4* mine4 = reinterpret_cast<f32x4*>(mine);
    constexpr int NX = HALF == 0 ? 3 : 2;
    f32x2 s0[2][4], s1[2][4];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        f32x2 rr[NX][4];
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            f32x2 m[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) m[k] = rp == 0 ? lo2(acc[x * 6 + k][rt]) : hi2(acc[x * 6 + k][rt]);
            const f32x2 p = m[1] + m[2], q = psub(m[1], m[2]), s = m[3] + m[4], t = psub(m[3], m[4]);
            rr[x][0] = (m[0] + p) + s;
            rr[x][1] = pfma(2.0f, t, q);
            rr[x][2] = pfma(4.0f, s, p);
            rr[x][3] = pfma(8.0f, t, q) + m[5];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (HALF == 0) {
                const f32x2 p = rr[0][j] + rr[1][j], q = psub(rr[0][j], rr[1][j]);
                A[rp][j] = p + rr[NX - 1][j];
                B[rp][j] = pfma(0.5f, rr[NX - 1][j], q);
                s0[rp][j] = pfma(0.25f, rr[NX - 1][j], p);
                s1[rp][j] = s0[rp][j];
            } else {
                A[rp][j] = pfma(0.25f, rr[0][j], rr[1][j]);
                B[rp][j] = A[rp][j];
                s0[rp][j] = rr[0][j];
                s1[rp][j] = pk2(-0.5f) * rr[0][j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mine4[(2 * j) * 64] = cat2(s0[0][j], s0[1][j]);
        if (HALF == 1) mine4[(2 * j + 1) * 64] = cat2(s1[0][j], s1[1][j]);
    }
}

template <int HALF, bool RELU, class ST>
__device__ __forceinline__ void wino43m_finish(const f32x2 (&A)[2][4], const f32x2 (&B)[2][4], const float4* theirs, const float4 b4, ST&& store,
                                               bool live, int row_stride)
{
    const f32x2 ba[2] = {(f32x2){b4.x, b4.y}, (f32x2){b4.z, b4.w}};
    const f32x4* theirs4 = reinterpret_cast<const f32x4*>(theirs);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 g0 = theirs4[(2 * j) * 64];
        f32x4 g1 = g0;
        if (HALF == 0) g1 = theirs4[(2 * j + 1) * 64];
        f32x2 y0[2], y1[2];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const f32x2 g0p = rp == 0 ? lo2(g0) : hi2(g0), g1p = rp == 0 ? lo2(g1) : hi2(g1);
            // Y = (P_0 + P_1) + bias: half 0 holds P_0 of its two rows and receives P_1, half 1 holds P_1 of the third row and receives P_0
            y0[rp] = (HALF == 0 ? A[rp][j] + g0p : g0p + A[rp][j]) + ba[rp];
            y1[rp] = (B[rp][j] + g1p) + ba[rp];
            if (RELU) {
                y0[rp] = (f32x2){y0[rp].x > 0.f ? y0[rp].x : 0.f, y0[rp].y > 0.f ? y0[rp].y : 0.f};
                y1[rp] = (f32x2){y1[rp].x > 0.f ? y1[rp].x : 0.f, y1[rp].y > 0.f ? y1[rp].y : 0.f};
            }
        }
        if (live) {
            store(j * 16, cat2(y0[0], y0[1]));
            if (HALF == 0) store(row_stride + j * 16, cat2(y1[0], y1[1]));
        }
    }
}

}  // namespace w43
