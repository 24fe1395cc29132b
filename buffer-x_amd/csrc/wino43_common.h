// wino43_common.h -- pieces shared by the two Winograd F(4x4, 3x3) kernels (k_wino43.hip: Cylindrical_Net, circular / zero-padded 7 x 20
// maps; k_wino43v.hip: CostNet layers 1..5, valid D x D maps): the input transform of a 6-vector and the two halves of the output
// transform of the swapped-operand form.  Arithmetic contract: oracle/bx_oracle.c::bxo_conv_wino43 / bxo_conv_wino43_valid.
#pragma once
#include "bx_common.h"

namespace w43 {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWF = 20;                         // floats per LDS row (16 + 4 pad)
constexpr int RT4 = 2, VR4 = RT4 * 16, VPL4 = VR4 * ROWF;   // two MFMA row tiles = 32 tile rows per workgroup item
constexpr int NPL = 36, NPH = 18;                // planes, planes per wave half
constexpr int CT = 512;

// the six results of B^T on a 6-vector (contract: bxo_conv_wino43; t3 / t4 as fmaf(+-2, d3 - d1, c): 2 x is exact, so the rounding is
// that of c +- e)
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

template <int HALF>
__device__ __forceinline__ void wino43_send(const f32x4 (&acc)[NPH][RT4], int rt, float (&ua)[4][4], float (&ub)[4][4], float (&uc)[4][4], float4* mine)
{
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
            if (HALF == 0) { ua[r][j] = rr[0][j]; ub[r][j] = rr[1][j] + rr[2][j]; uc[r][j] = rr[1][j] - rr[2][j]; }
            else           { ua[r][j] = rr[0][j] + rr[1][j]; ub[r][j] = rr[0][j] - rr[1][j]; uc[r][j] = rr[2][j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (HALF == 0) {                                    // P_0[2] = r_1 + r_2, P_0[3] = r_1 - r_2
            mine[(2 * j) * 64] = make_float4(ub[0][j], ub[1][j], ub[2][j], ub[3][j]);
            mine[(2 * j + 1) * 64] = make_float4(uc[0][j], uc[1][j], uc[2][j], uc[3][j]);
        } else {                                            // P_1[0] = r_3 + r_4, P_1[1] = 2 (r_3 - r_4)
            mine[(2 * j) * 64] = make_float4(ua[0][j], ua[1][j], ua[2][j], ua[3][j]);
            mine[(2 * j + 1) * 64] = make_float4(2.0f * ub[0][j], 2.0f * ub[1][j], 2.0f * ub[2][j], 2.0f * ub[3][j]);
        }
    }
}

template <int HALF, bool RELU>
__device__ __forceinline__ void wino43_finish(const float (&ua)[4][4], const float (&ub)[4][4], const float (&uc)[4][4], const float4* theirs,
                                              const float4 b4, float* ou, bool live, bool second_row, int row_stride, unsigned jmask)
{
    const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 g0 = theirs[(2 * j) * 64], g1 = theirs[(2 * j + 1) * 64];
        const float g0a[4] = {g0.x, g0.y, g0.z, g0.w}, g1a[4] = {g1.x, g1.y, g1.z, g1.w};
        float y0[4], y1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p0, p1, q0, q1;                           // (P_0, P_1) of the half's two output rows
            if (HALF == 0) { p0 = ua[r][j] + ub[r][j]; p1 = g0a[r]; q0 = uc[r][j]; q1 = g1a[r]; }
            else           { p0 = g0a[r]; p1 = 4.0f * ua[r][j]; q0 = g1a[r]; q1 = fmaf(8.0f, ub[r][j], uc[r][j]); }
            y0[r] = (p0 + p1) + ba[r];
            y1[r] = (q0 + q1) + ba[r];
            if (RELU) { y0[r] = y0[r] > 0.f ? y0[r] : 0.f; y1[r] = y1[r] > 0.f ? y1[r] : 0.f; }
        }
        if (live && ((jmask >> j) & 1u)) {                  // jmask: output columns of the tile that exist (valid maps: the last tile column)
            __builtin_nontemporal_store((f32x4){y0[0], y0[1], y0[2], y0[3]}, reinterpret_cast<f32x4*>(ou + j * 16));
            if (second_row) __builtin_nontemporal_store((f32x4){y1[0], y1[1], y1[2], y1[3]}, reinterpret_cast<f32x4*>(ou + row_stride + j * 16));
        }
    }
}

}  // namespace w43
