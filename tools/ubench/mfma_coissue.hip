// micro-benchmark 4 (round 5 rewrite): WHICH instruction classes run under a stream of f32 MFMAs on a gfx950 SIMD, and where?
//
// Round 4 measured one mix (LDS reads + fmaf chains + LDS writes in the SIMD's other wave: both / sum = 1.00) and concluded "nothing
// co-issues".  The review pointed out that hipcc had SLP-packed that "fmaf chain" into v_pk_fma_f32 (the class MI355X_MICROARCH.md
// lists as an anti-lever beside MFMAs), so the conclusion was not established per class.  This version pins every instruction with
// inline asm (the ISA is what the source says: check with `llvm-objdump -d`), one row per class:
//
//   plain f32 VALU      v_fma_f32, v_add_f32
//   packed f32 VALU     v_pk_fma_f32, v_pk_add_f32
//   integer / move VALU v_add_u32, v_xor_b32, v_lshl_add_u32, v_mov_b32
//   LDS                 ds_read_b32, ds_read_b128, ds_write_b32, ds_write_b128
//   VMEM                global_load_dword (L1 / L2 resident), global_load_dwordx4
//   SALU                s_add_u32, s_nop 0
//
// in two placements:
//   SAME  the filler instructions sit in the MFMA wave itself, NF of them behind every v_mfma_f32_16x16x4_f32 (issue interval 32
//         cycles per SIMD: the guide hides <= 5 single-issue fillers per 32-cycle gap of a bf16 MFMA stream);
//   OTHER the fillers run in the SIMD's second wave (waves 4..7 of the workgroup; waves 0..3 stream the MFMAs), NF per MFMA-time.
// One workgroup of 8 waves per CU, 256 workgroups.  For every row three launches: MFMAs only, fillers only, both; loop time =
// time(2 x iterations) - time(iterations).  Reported per MFMA slot (144 per iteration): cycles from s_memtime of workgroup 0 and
// the wall-clock ratio both / max(M, F), both / (M + F), and the EXTRA cycles one filler adds to the MFMA stream.
// Round 6: the same table under a stream of BF16 MFMAs (MK = 1: v_mfma_f32_16x16x32_bf16, 16-cycle issue interval; MK = 2:
// v_mfma_f32_32x32x16_bf16, 32-cycle interval) -- the question behind a Winograd-domain 3 x bf16 split form of the conv stacks: do the
// transform / split VALU instructions and the fragment loads hide beside bf16 MFMAs, where they serialise 1 : 1 with f32 MFMAs?
// Extra classes for that form: v_cvt_pk_bf16_f32, v_and_b32, v_lshlrev_b32, v_sub_f32, v_perm_b32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue [f32|bf16|all]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { C_FMA, C_ADD, C_PKFMA, C_PKADD, C_IADD, C_XOR, C_LSHLADD, C_MOV, C_DSR32, C_DSR128, C_DSW32, C_DSW128, C_GLD, C_GLD4, C_SALU, C_SNOP, C_CVTPK, C_AND, C_LSHL, C_SUB, C_PERM, NCLS };
static const char* CLS_NAME[NCLS] = {"v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_add_u32", "v_xor_b32", "v_lshl_add_u32", "v_mov_b32",
                                     "ds_read_b32", "ds_read_b128", "ds_write_b32", "ds_write_b128", "global_load_dword", "global_load_dwordx4", "s_add_u32", "s_nop 0",
                                     "v_cvt_pk_bf16_f32", "v_and_b32", "v_lshlrev_b32", "v_sub_f32", "v_perm_b32"};

struct Regs {
    float f[8];
    f32x2 p[4];
    f32x4 q[4];
    int i[8];
    float c1, c2;
    f32x2 pc1, pc2;
    unsigned lds;          // byte address of this lane's LDS word
    unsigned lds16;        // byte address of this lane's 16-byte LDS piece
    const float* gp;       // this lane's global word (L1 / L2 resident)
    int s;
};

template <int CLS>
__device__ __forceinline__ void filler(Regs& r, int k)
{
    if constexpr (CLS == C_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.f[k & 7]) : "v"(r.c1), "v"(r.c2));
    else if constexpr (CLS == C_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.f[k & 7]) : "v"(r.c2));
    else if constexpr (CLS == C_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r.p[k & 3]) : "v"(r.pc1), "v"(r.pc2));
    else if constexpr (CLS == C_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r.p[k & 3]) : "v"(r.pc2));
    else if constexpr (CLS == C_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.i[k & 7]) : "v"(r.i[(k + 3) & 7]));
    else if constexpr (CLS == C_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r.i[k & 7]) : "v"(r.i[(k + 3) & 7]));
    else if constexpr (CLS == C_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r.i[k & 7]) : "v"(r.i[(k + 3) & 7]));
    else if constexpr (CLS == C_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(r.i[k & 7]) : "v"(r.i[(k + 3) & 7]));
    else if constexpr (CLS == C_DSR32) asm volatile("ds_read_b32 %0, %1" : "+v"(r.f[k & 7]) : "v"(r.lds) : "memory");
    else if constexpr (CLS == C_DSR128) asm volatile("ds_read_b128 %0, %1" : "+v"(r.q[k & 3]) : "v"(r.lds16) : "memory");
    else if constexpr (CLS == C_DSW32) asm volatile("ds_write_b32 %0, %1" : : "v"(r.lds), "v"(r.c1) : "memory");
    else if constexpr (CLS == C_DSW128) asm volatile("ds_write_b128 %0, %1" : : "v"(r.lds16), "v"(r.q[0]) : "memory");
    else if constexpr (CLS == C_GLD) asm volatile("global_load_dword %0, %1, off" : "+v"(r.f[k & 7]) : "v"(r.gp) : "memory");
    else if constexpr (CLS == C_GLD4) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r.q[k & 3]) : "v"(r.gp) : "memory");
    else if constexpr (CLS == C_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(r.s) : : "scc");
    else if constexpr (CLS == C_SNOP) asm volatile("s_nop 0");
    else if constexpr (CLS == C_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.i[k & 7]) : "v"(r.f[k & 7]), "v"(r.f[(k + 3) & 7]));
    else if constexpr (CLS == C_AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r.i[k & 7]) : "v"(r.i[(k + 3) & 7]));
    else if constexpr (CLS == C_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r.i[k & 7]) : "v"(r.i[(k + 3) & 7]));
    else if constexpr (CLS == C_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r.f[k & 7]) : "v"(r.c2));
    else if constexpr (CLS == C_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r.i[k & 7]) : "v"(r.i[(k + 3) & 7]), "v"(r.i[(k + 5) & 7]), "v"(r.i[(k + 6) & 7]));
}

// the MFMA of the stream: MK 0 = v_mfma_f32_16x16x4_f32 (32 cycles), 1 = v_mfma_f32_16x16x32_bf16 (16), 2 = v_mfma_f32_32x32x16_bf16 (32)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MK> struct Acc { f32x4 a[4]; };
template <> struct Acc<2> { f32x16 a[4]; };
template <int MK>
__device__ __forceinline__ void mfma(Acc<MK>& A, int u, float x, float y, const bf16x8& bx, const bf16x8& by)
{
    if constexpr (MK == 0) A.a[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, A.a[u & 3], 0, 0, 0);
    else if constexpr (MK == 1) A.a[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, A.a[u & 3], 0, 0, 0);
    else A.a[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, A.a[u & 3], 0, 0, 0);
}

constexpr int NM = 144;    // MFMA slots per iteration (one chunk of the F(4x4) kernel per wave)

template <int MK, int CLS, int NF, bool SAME, bool DO_M, bool DO_F>
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[8 * 64 * 4 * 2];
    for (int i = threadIdx.x; i < 8 * 64 * 4 * 2; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Regs r;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r.f[i] = in[lane + i]; r.i[i] = lane * 7 + i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) r.p[i] = (f32x2){in[lane + i], in[lane + 2 * i]};
    r.q[0] = (f32x4){in[lane], in[lane + 1], in[lane + 2], in[lane + 3]};
    r.q[1] = r.q[0]; r.q[2] = r.q[0]; r.q[3] = r.q[0];
    r.c1 = 1.0001f; r.c2 = in[lane + 9] * 1e-3f;
    r.pc1 = (f32x2){1.0001f, 0.9999f}; r.pc2 = (f32x2){r.c2, r.c2};
    r.lds = (unsigned)(threadIdx.x * 4);          // the array is the kernel's only LDS object: it starts at LDS address 0
    r.lds16 = (unsigned)(threadIdx.x * 16);
    r.gp = in + (threadIdx.x & 1023);
    r.s = __builtin_amdgcn_readfirstlane(wave);
    Acc<MK> acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc.a[i] = 0.0f;
    const float x = in[lane], y = in[lane + 64];
    bf16x8 bx, by;
#pragma unroll
    for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)in[lane + i]; by[i] = (__bf16)in[lane + 64 + i]; }
    const bool mwave = wave < 4;
    long long t0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) t0 = __builtin_readcyclecounter();
    if (SAME) {
        if (mwave)      // waves 4..7 idle: one wave per SIMD
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < NM; ++u) {
                    if (DO_M) mfma<MK>(acc, u, x, y, bx, by);
                    __builtin_amdgcn_sched_barrier(0);
                    if (DO_F) {
#pragma unroll
                        for (int kf = 0; kf < NF; ++kf) filler<CLS>(r, u * NF + kf);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
    } else {
        if (mwave) {
            if (DO_M)
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int u = 0; u < NM; ++u) {
                        mfma<MK>(acc, u, x, y, bx, by);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        } else if (DO_F) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < NM * NF; ++u) filler<CLS>(r, u);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = (long long)__builtin_readcyclecounter() - t0;
    float res = acc.a[0][0] + acc.a[1][1] + acc.a[2][2] + acc.a[3][3] + r.q[0].x + r.q[1].y + r.q[2].z + r.q[3].w + (float)r.s;
#pragma unroll
    for (int i = 0; i < 8; ++i) res += r.f[i] + (float)r.i[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) res += r.p[i].x + r.p[i].y;
    if (res == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

static float *g_in, *g_out;
static long long* g_cyc;

template <int MK, int CLS, int NF, bool SAME, bool DO_M, bool DO_F>
static void measure(double& ms, double& cycles)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double best[2] = {1e9, 1e9}; long long bc[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep)
        for (int h = 0; h < 2; ++h) {
            hipEventRecord(a);
            hipLaunchKernelGGL((k<MK, CLS, NF, SAME, DO_M, DO_F>), dim3(256), dim3(512), 0, 0, g_in, g_out, g_cyc, 100 * (h + 1));
            hipEventRecord(b); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b);
            long long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
            if (rep > 0 && t < best[h]) { best[h] = t; bc[h] = c; }
        }
    ms = best[1] - best[0];
    cycles = (double)(bc[1] - bc[0]) / (100.0 * NM);        // s_memtime ticks per MFMA slot
    hipEventDestroy(a); hipEventDestroy(b);
}

static double g_m_ms[3][2], g_m_cyc[3][2];     // MFMA-only baselines [MK][SAME]
static const char* MK_NAME[3] = {"f32_16x16x4", "bf16_16x16x32", "bf16_32x32x16"};

template <int MK, int CLS, int NF, bool SAME>
static void row()
{
    double fm, fc, bm, bcy;
    measure<MK, CLS, NF, SAME, false, true>(fm, fc);
    measure<MK, CLS, NF, SAME, true, true>(bm, bcy);
    const double mm = g_m_ms[MK][SAME], mc = g_m_cyc[MK][SAME];
    printf("%-13s %-20s %-5s NF=%d | MFMA only %6.1f cyc/slot | fillers only %6.1f cyc/slot (%5.2f per filler) | both %6.1f cyc/slot | both/max %.2f both/sum %.2f | extra per filler %+6.2f cyc | wall ms M %.3f F %.3f both %.3f\n",
           MK_NAME[MK], CLS_NAME[CLS], SAME ? "SAME" : "OTHER", NF, mc, fc, fc / NF, bcy, bm / (mm > fm ? mm : fm), bm / (mm + fm), (bcy - mc) / NF, mm, fm, bm);
    fflush(stdout);
}

template <int MK, int CLS>
static void cls_rows()
{
    if constexpr (MK == 1) {          // 16-cycle gaps
        row<MK, CLS, 1, true>(); row<MK, CLS, 2, true>(); row<MK, CLS, 3, true>(); row<MK, CLS, 4, true>();
        row<MK, CLS, 1, false>(); row<MK, CLS, 2, false>(); row<MK, CLS, 4, false>();
    } else {
        row<MK, CLS, 1, true>(); row<MK, CLS, 2, true>(); row<MK, CLS, 4, true>(); row<MK, CLS, 6, true>();
        row<MK, CLS, 2, false>(); row<MK, CLS, 4, false>(); row<MK, CLS, 8, false>();
    }
}

template <int MK>
static void base()
{
    measure<MK, C_FMA, 1, true, true, false>(g_m_ms[MK][1], g_m_cyc[MK][1]);
    measure<MK, C_FMA, 1, false, true, false>(g_m_ms[MK][0], g_m_cyc[MK][0]);
    printf("# %s MFMA only: one wave per SIMD (SAME placement) %.2f cyc/MFMA, %.3f ms per 100 x 144; with an idle sibling wave resident (OTHER) %.2f cyc/MFMA, %.3f ms\n",
           MK_NAME[MK], g_m_cyc[MK][1], g_m_ms[MK][1], g_m_cyc[MK][0], g_m_ms[MK][0]);
}

template <int MK>
static void bf16_rows()
{
    base<MK>();
    cls_rows<MK, C_FMA>(); cls_rows<MK, C_SUB>(); cls_rows<MK, C_PKFMA>(); cls_rows<MK, C_CVTPK>(); cls_rows<MK, C_AND>(); cls_rows<MK, C_LSHL>();
    cls_rows<MK, C_PERM>(); cls_rows<MK, C_DSR32>(); cls_rows<MK, C_DSR128>(); cls_rows<MK, C_DSW32>(); cls_rows<MK, C_DSW128>();
    cls_rows<MK, C_GLD4>(); cls_rows<MK, C_SALU>();
}

int main(int argc, char** argv)
{
    const char* what = argc > 1 ? argv[1] : "all";
    const bool f32 = what[0] == 'f' || what[0] == 'a', bf = what[0] == 'b' || what[0] == 'a';
    hipMalloc(&g_in, 4096 * 4); hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 64);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 37) % 101) * 0.01f + 0.5f;
    hipMemcpy(g_in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    if (f32) {
        base<0>();
        cls_rows<0, C_FMA>(); cls_rows<0, C_ADD>(); cls_rows<0, C_PKFMA>(); cls_rows<0, C_PKADD>();
        cls_rows<0, C_IADD>(); cls_rows<0, C_XOR>(); cls_rows<0, C_LSHLADD>(); cls_rows<0, C_MOV>();
        cls_rows<0, C_DSR32>(); cls_rows<0, C_DSR128>(); cls_rows<0, C_DSW32>(); cls_rows<0, C_DSW128>();
        cls_rows<0, C_GLD>(); cls_rows<0, C_GLD4>(); cls_rows<0, C_SALU>(); cls_rows<0, C_SNOP>();
    }
    if (bf) { bf16_rows<2>(); bf16_rows<1>(); }
    return 0;
}
