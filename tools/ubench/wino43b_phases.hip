// micro-benchmark 5 (round 6): the PHASES of a Winograd-domain 3 x bf16 split form of the F(4x4, 3x3) Cylindrical_Net kernel, as
// skeletons with the real instruction mix and the real LDS / register / L2 footprints -- before anybody writes the kernel.
//
// The form: V = B^T d B in fp32 as today, then every V element is split into three bf16 parts
// (hi = rn(v), mid = rn(v - hi), lo = rn(v - hi - mid): 8 + 8 + 8 significand bits, v = hi + mid + lo to 2^-26), the pre-split filter
// transform U likewise, and a plane's channel contraction is SIX bf16 MFMAs (hh, hm, mh, hl, lh, mm) with fp32 accumulation instead of
// the f32 MFMAs: 6 / 16 of today's matrix-pipe time.  What that costs elsewhere is what this file measures, one phase per launch:
//
//   M  the MFMA phase of one (item, chunk): a wave owns a 32-column tile x 9 planes (acc 144 VGPRs, as today); per plane 3 U fragments
//      (buffer_load_dwordx4 from the layer's 3.5 MB split U -- L2) + 3 V fragments (ds_read_b128) + 6 v_mfma_f32_32x32x16_bf16;
//   S  the same with the slab traffic of the real kernel beside it (5 non-temporal 16-byte loads per thread and chunk + LDS writes);
//   T  the transform phase: wave = xi half, lane = (tile row, channel pair): 30 ds_read_b64, B^T d B for three xi rows on two channels, the
//      three-way split (v_cvt_pk_bf16_f32 + shift / and + v_sub_f32), 54 ds_write_b32;
//   A  all of it, barrier-separated as the kernel would be (T -> barrier -> M + S -> barrier).
// Geometry of Cylindrical_Net layer 3 (128 -> 128: 8 chunks, 2 column blocks of 64, 1 563 items of 32 tile rows at 5 000 units),
// one 512-thread workgroup per CU (LDS: V 110 592 B + slab 51 504 B), persistent walk over the items.  Prints cycles per (item, chunk)
// from the wall clock at the measured clock, and the projected layer time = phases + the output phase of today's kernel (stamps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wino43b_phases wino43b_phases.hip && ./wino43b_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CT = 512, NCHUNK = 8, NPL = 36, NT32 = 4 /* 128 columns */, ROWF = 20, RP3 = 22 * ROWF + 4;
constexpr int V_BYTES = NPL * 3 * 1024, SLAB_ROWS = 4 * 7 + 1, SLAB_BYTES = SLAB_ROWS * RP3 * 4;
constexpr int LDS_BYTES = V_BYTES + SLAB_BYTES + 256;

__device__ __forceinline__ unsigned cvt_pk(float a, float b)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void bt_lo(float d0, float d1, float d2, float d3, float d4, float& o0, float& o1, float& o2)
{
    o0 = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    const float a = fmaf(-4.0f, d2, d4), b = fmaf(-4.0f, d1, d3);
    o1 = a + b; o2 = a - b;
}
__device__ __forceinline__ void bt_hi(float d1, float d2, float d3, float d4, float d5, float& o3, float& o4, float& o5)
{
    const float c = d4 - d2, s = d3 - d1;
    o3 = fmaf(2.0f, s, c); o4 = fmaf(-2.0f, s, c);
    o5 = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
}
__device__ __forceinline__ void bt6(const float (&d)[6], float (&o)[6])
{
    bt_lo(d[0], d[1], d[2], d[3], d[4], o[0], o[1], o[2]);
    bt_hi(d[1], d[2], d[3], d[4], d[5], o[3], o[4], o[5]);
}

template <bool DO_T, bool DO_M, bool DO_S>
__global__ __launch_bounds__(CT, 2) void phases(const u32x4* __restrict__ U, const float* __restrict__ in, float* __restrict__ out, int nitems, int in_units)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Vb = smem;
    float* slab = reinterpret_cast<float*>(smem + V_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LDS_BYTES / 16; i += CT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.001f * (i & 63), 0.5f, 0.25f, 1.0f);
    __syncthreads();

    // ---- MFMA role: wave = (column tile ct of the workgroup's 64 columns, plane group pg: planes 9 pg .. 9 pg + 8)
    const int ct = wave & 1, pg = wave >> 1;
    const int ctg = (int)blockIdx.y * 2 + ct;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(U), 0, NCHUNK * NPL * NT32 * 3 * 1024, 0x00020000);
    const int ulane = lane * 16;
    auto uload = [&](int c, int p, int part) {      // fragment [chunk][plane][column tile][part][lane]
        return __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, (((c * NPL + pg * 9 + p) * NT32 + ctg) * 3 + part) * 1024, 0);
    };
    const char* vsrc = Vb + (pg * 9) * 3072 + (lane & 31) * 32 + (lane >> 5) * 16;
    // ---- transform role: wave = (xi half wave & 1 -- wave-uniform: no divergent column pass --, eight tile rows), lane = (tile row, channel pair)
    const int cp = lane & 7, xh = wave & 1, row = 8 * (wave >> 1) + (lane >> 3);
    const int tile = row % 10, tr = tile / 5, tc = tile % 5, slot = row / 10;
    const float* wsrc = slab + ((slot * 7 + 3 * tr) * RP3 + 4 * tc * ROWF + 2 * cp);
    char* vdst = Vb + (xh * 18) * 3072 + row * 32 + cp * 4;
    // ---- slab role: five 16-byte pieces per thread and chunk
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);

    f32x16 acc[9];
    float keep = 0.f;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
#pragma unroll
        for (int p = 0; p < 9; ++p) acc[p] = 0.0f;
        const int u0 = (item * 32) / 10 % in_units;
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            f32x4 st[5];
            if (DO_S) {
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int f = tid + q * CT;     // 4 units x 140 positions x 4 parts = 2 240 pieces
                    st[q] = f < 2240 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, f * 16, ((u0 * NCHUNK + c) * 560) * 16, 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            if (DO_T) {
                // B^T d B for the xi rows 3 xh .. 3 xh + 2 on the channel pair; window rows 0..4 (xh = 0) or 1..5 (xh = 1)
                float t[2][3][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x2 d[5];
#pragma unroll
                    for (int r = 0; r < 5; ++r) d[r] = *reinterpret_cast<const f32x2*>(wsrc + (r + xh) * RP3 + j * ROWF);
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        if (xh == 0) bt_lo(d[0][ch], d[1][ch], d[2][ch], d[3][ch], d[4][ch], t[ch][0][j], t[ch][1][j], t[ch][2][j]);
                        else bt_hi(d[0][ch], d[1][ch], d[2][ch], d[3][ch], d[4][ch], t[ch][0][j], t[ch][1][j], t[ch][2][j]);
                    }
                }
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    float o[2][6];
                    bt6(t[0][x], o[0]);
                    bt6(t[1][x], o[1]);
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) {
                        const float v0 = o[0][nu], v1 = o[1][nu];
                        const unsigned H = cvt_pk(v0, v1);
                        const float r0 = v0 - __uint_as_float(H << 16), r1 = v1 - __uint_as_float(H & 0xffff0000u);
                        const unsigned M = cvt_pk(r0, r1);
                        const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
                        const unsigned L = cvt_pk(s0, s1);
                        unsigned* d = reinterpret_cast<unsigned*>(vdst + (x * 6 + nu) * 3072);
                        d[0] = H; d[256] = M; d[512] = L;
                    }
                }
            }
            __syncthreads();             // V complete
            if (DO_M) {
                u32x4 ur[2][3];
#pragma unroll
                for (int part = 0; part < 3; ++part) ur[0][part] = uload(c, 0, part);
#pragma unroll
                for (int p = 0; p < 9; ++p) {
                    if (p + 1 < 9) {
#pragma unroll
                        for (int part = 0; part < 3; ++part) ur[(p + 1) & 1][part] = uload(c, p + 1, part);
                    }
                    bf16x8 v[3], u[3];
#pragma unroll
                    for (int part = 0; part < 3; ++part) {
                        v[part] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(vsrc + p * 3072 + part * 1024));
                        u[part] = __builtin_bit_cast(bf16x8, ur[p & 1][part]);
                    }
                    // small terms first
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[1], v[1], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], v[2], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[2], v[0], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], v[1], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[1], v[0], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], v[0], acc[p], 0, 0, 0);
                }
            }
            if (DO_S) {
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int f = tid + q * CT;
                    if (f < 2240) *reinterpret_cast<f32x4*>(slab + ((f >> 2) / 20 % SLAB_ROWS) * RP3 + ((f >> 2) % 20 + 1) * ROWF + (f & 3) * 4) = st[q];
                }
            }
            __syncthreads();             // the slab of the next chunk is complete, V is free
        }
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[p][r];
    }
    if (keep == 12345.678f) out[blockIdx.x * CT + tid] = keep + reinterpret_cast<float*>(Vb)[tid];
    if (!DO_M && tid == 0) out[blockIdx.x] = reinterpret_cast<float*>(Vb)[(blockIdx.x * 7) & 1023];
}

template <bool DO_T, bool DO_M, bool DO_S>
static double run(const u32x4* U, const float* in, float* out, int nitems, int units, const char* name, double out_cycles)
{
    auto k = phases<DO_T, DO_M, DO_S>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(128, 2), dim3(CT), LDS_BYTES, 0, U, in, out, nitems, units);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float t; (void)hipEventElapsedTime(&t, a, b);
        if (rep > 0 && t < best) best = t;
    }
    const int rounds = (nitems + 127) / 128;
    const double us = best * 1e3, per_chunk_us = us / (rounds * NCHUNK);
    printf("%-28s %8.1f us per launch (%d items, %d rounds of 128 x 2 workgroups) = %6.3f us per (item, chunk) = %6.0f cycles at 2.2 GHz\n", name, us, nitems, rounds,
           per_chunk_us, per_chunk_us * 2200.0);
    fflush(stdout);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return us;
}

int main()
{
    const int units = 5000, nitems = (units * 10 + 31) / 32;
    u32x4* U; float *in, *out;
    const size_t ub = (size_t)NCHUNK * NPL * NT32 * 3 * 1024, ib = (size_t)units * NCHUNK * 140 * 64;
    (void)hipMalloc(&U, ub); (void)hipMalloc(&in, ib + 65536); (void)hipMalloc(&out, 1 << 20);
    std::vector<unsigned short> h(ub / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22 & 0x1ff));      // bf16 values around 0.01
    (void)hipMemcpy(U, h.data(), ub, hipMemcpyHostToDevice);
    std::vector<float> hi(ib / 4 + 16384);
    for (size_t i = 0; i < hi.size(); ++i) hi[i] = (float)((i * 37) % 101) * 0.01f;
    (void)hipMemcpy(in, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    printf("# Winograd F(4x4) bf16 x 3 split form, phase skeletons, Cylindrical_Net layer 3 geometry (8 chunks, 128 columns, %d items); LDS %d B; U %.2f MB\n",
           nitems, LDS_BYTES, ub / 1e6);
    run<false, true, false>(U, in, out, nitems, units, "M  (MFMA phase)", 0);
    run<false, true, true>(U, in, out, nitems, units, "M+S (MFMA + slab traffic)", 0);
    run<true, false, false>(U, in, out, nitems, units, "T  (transform + split)", 0);
    const double all = run<true, true, true>(U, in, out, nitems, units, "A  (T | barrier | M+S)", 0);
    // today's kernel: output phase 13 326 cycles per item (profiles/r05_wino43_variants.txt, item32 L3 stamps), 13 rounds
    const double out_us = 13326.0 / 2200.0 * ((nitems + 127) / 128);
    printf("# projected layer 3 = A + today's output phase (13 326 cycles per item, 13 rounds = %.0f us) = %.0f us   (shipped f32 form: 709-712 us; gate <= 480 us)\n",
           out_us, all + out_us);
    return 0;
}
