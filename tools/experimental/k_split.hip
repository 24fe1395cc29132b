// k_split.hip -- MEASUREMENT ONLY (BX_EXP_SPLIT_CONV=1; never a default, never the headline): the 128 -> 128 Cylindrical_Net layer
// (models/patchnet.py:76, layer 3 of the stack) as a SPLIT-PRECISION convolution on the bf16 matrix cores.
//
// Every fp32 activation and weight is cut into three bf16 pieces by truncation, x = hi + mid + lo EXACTLY (8 + 8 + 8 = the 24
// mantissa bits of binary32; the residuals x - hi and (x - hi) - mid are exact in binary32), and the product a * w is replaced by
// its six leading partial products
//       a_lo w_hi + a_hi w_lo + a_mid w_mid + a_mid w_hi + a_hi w_mid + a_hi w_hi
// (dropped: mid*lo, lo*mid, lo*lo <= 2^-24 |a w| each), every piece product exact in the fp32 accumulator of
// v_mfma_f32_16x16x32_bf16, the accumulation itself in fp32 like the exact kernels'.  16x the f32 MFMA rate x 6 products per
// multiplication = 6/16 of the matrix-pipe time of the direct f32 form.  The review of round 2 asked for this number to be MEASURED
// against the exact kernels (tools/split_precision.py: per-element error against a binary64 convolution next to the direct f32
// and the Winograd f32 forms; descriptor / ind / pose deltas on the reference-minted fixtures); the shipped path stays exact f32.
//
// Kernel (simple, not tuned): one workgroup = 8 waves = the 8 column tiles of a unit, persistent walk over the units; per
// 32-channel block the unit's map (with its cylindrical halo, 9 x 22 rows) is staged in LDS as three bf16 planes (rows of 32 bf16 +
// 16 bytes of padding: conflict-free 16-byte A reads); per tap a wave holds the three B fragments of its column tile (pre-split on the
// host) and walks the 9 row tiles: three ds_read_b128 + six MFMAs each.
#include "bx_common.h"
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SROWB = 80;                                // bytes per slab row and piece: 32 bf16 + 16 pad
constexpr int SROWS = (BX_ELE + 2) * (BX_AZI + 2);      // 198
constexpr int SPLANE = SROWS * SROWB;                    // bytes per piece plane
constexpr int SCT = 512;
constexpr int NRT = (BX_EA + 15) / 16;                   // 9 row tiles (140 positions)

__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo)
{
    const uint32_t bh = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(bh);            // exact
    const uint32_t bm = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(bm);           // exact
    hi = bh >> 16; mid = bm >> 16; lo = __float_as_uint(r2) >> 16;
}

template <int NKB, int COUT, bool RELU>
__global__ __launch_bounds__(SCT, 2) void split_conv_kernel(const float* __restrict__ in, int units, const uint4* __restrict__ Wsp,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    static_assert(COUT == 128, "one workgroup = 8 column tiles");
    constexpr int NCHUNK = NKB * 2, NT = COUT / 16;
    constexpr int NPIECE = BX_EA * 8, NLD = (NPIECE + SCT - 1) / SCT;
    __shared__ __attribute__((aligned(16))) char slab[3 * SPLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    for (int i = tid; i < 3 * SPLANE / 16; i += SCT) reinterpret_cast<uint4*>(slab)[i] = make_uint4(0, 0, 0, 0);

    int arow[NRT];                                       // byte offset of the window origin of this lane's row of every row tile
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        int p = rt * 16 + li;
        p = p < BX_EA ? p : BX_EA - 1;
        const int h = p / BX_AZI, w = p - h * BX_AZI;
        arow[rt] = (h * (BX_AZI + 2) + w) * SROWB + kk * 16;
    }
    const float bv = bias[ct * 16 + li];
    const float4* in4 = reinterpret_cast<const float4*>(in);

    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        f32x4 acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            __syncthreads();                             // every wave is done with the slab of the block before
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int f = tid + q * SCT;
                if (f < NPIECE) {
                    const int pos = f >> 3, cin = (f >> 2) & 1, part = f & 3;
                    const float4 v = in4[(((size_t)u * NCHUNK + 2 * kb + cin) * BX_EA + pos) * 4 + part];
                    uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
                    split3(v.x, h0, m0, l0); split3(v.y, h1, m1, l1); split3(v.z, h2, m2, l2); split3(v.w, h3, m3, l3);
                    const uint2 ph = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
                    const uint2 pm = make_uint2(m0 | (m1 << 16), m2 | (m3 << 16));
                    const uint2 pl = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
                    const int h = pos / BX_AZI, w = pos - h * BX_AZI;
                    char* d = slab + ((h + 1) * (BX_AZI + 2) + (w + 1)) * SROWB + (cin * 16 + part * 4) * 2;
                    *reinterpret_cast<uint2*>(d) = ph;
                    *reinterpret_cast<uint2*>(d + SPLANE) = pm;
                    *reinterpret_cast<uint2*>(d + 2 * SPLANE) = pl;
                    if (w == 0 || w == BX_AZI - 1) {
                        char* e = w == 0 ? d + BX_AZI * SROWB : d - BX_AZI * SROWB;
                        *reinterpret_cast<uint2*>(e) = ph;
                        *reinterpret_cast<uint2*>(e + SPLANE) = pm;
                        *reinterpret_cast<uint2*>(e + 2 * SPLANE) = pl;
                    }
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const uint4* wp = Wsp + ((((size_t)kb * 9 + tap) * NT + ct) * 3) * 64 + lane;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, wp[0]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, wp[64]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, wp[128]);
                const int toff = ((tap / 3) * (BX_AZI + 2) + tap % 3) * SROWB;
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    const char* a = slab + arow[rt] + toff;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a));
                    const bf16x8 am = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a + SPLANE));
                    const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a + 2 * SPLANE));
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[rt], 0, 0, 0);
                }
            }
        }
        float* ou = out + ((size_t)u * NT + ct) * BX_EA * 16 + (4 * (li & 3) + (li >> 2));
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = rt * 16 + kk * 4 + r;
                if (p < BX_EA) {
                    float y = acc[rt][r] + bv;
                    if (RELU) y = y > 0.0f ? y : 0.0f;
                    ou[p * 16] = y;
                }
            }
    }
}
}  // namespace

// the three bf16 pieces of every weight as B fragments [32-channel block][tap][column tile][piece][lane = kk*16 + li][8]: element j of
// lane (li, kk) = piece of w[k = kk*8 + j][column li], k = (chunk inside the block)*16 + slot, slot s = channel (s & 3)*4 + (s >> 2)
int bxk_split_weights(const float* w /* [nchunk][9][16][cout] */, int nchunk, int cout, void** d_out)
{
    const int nkb = nchunk / 2, nt = cout / 16;
    std::vector<uint16_t> frag((size_t)nkb * 9 * nt * 3 * 64 * 8, 0);
    for (int kb = 0; kb < nkb; ++kb)
        for (int tap = 0; tap < 9; ++tap)
            for (int t = 0; t < nt; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int li = lane & 15, kk = lane >> 4, k = kk * 8 + j;
                        const int chunk = 2 * kb + (k >> 4), s = k & 15, ch = (s & 3) * 4 + (s >> 2);
                        const float x = w[(((size_t)chunk * 9 + tap) * 16 + ch) * cout + t * 16 + li];
                        uint32_t b; float f;
                        memcpy(&b, &x, 4);
                        const uint32_t bh = b & 0xFFFF0000u;
                        memcpy(&f, &bh, 4);
                        const float r1 = x - f;
                        memcpy(&b, &r1, 4);
                        const uint32_t bm = b & 0xFFFF0000u;
                        memcpy(&f, &bm, 4);
                        const float r2 = r1 - f;
                        memcpy(&b, &r2, 4);
                        const size_t base = ((((size_t)kb * 9 + tap) * nt + t) * 3) * 64 * 8 + (size_t)lane * 8 + j;
                        frag[base] = (uint16_t)(bh >> 16);
                        frag[base + 64 * 8] = (uint16_t)(bm >> 16);
                        frag[base + 2 * 64 * 8] = (uint16_t)(b >> 16);
                    }
    BX_HIP(hipMalloc(d_out, frag.size() * sizeof(uint16_t)));
    BX_HIP(hipMemcpy(*d_out, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return BX_OK;
}

// layer 3 of Cylindrical_Net in the split-precision form; -1 for every other layer / a device-side unit count
int bxk_split_conv(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out)
{
    if (layer != 3 || units_dev || max_units < 1 || !c->desc[3].Wsplit) return -1;
    const ConvLayerDev& L = c->desc[3];
    int grid = 2 * c->n_cu;
    if (grid > max_units) grid = max_units;
    hipLaunchKernelGGL((split_conv_kernel<4, 128, true>), dim3(grid), dim3(SCT), 0, s, in, max_units,
                       reinterpret_cast<const uint4*>(L.Wsplit), L.b, out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
