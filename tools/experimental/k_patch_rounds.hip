// tools/experimental/k_patch_rounds.hip -- NOT part of the library (round-4 experiment, kept for the record).  patch_features_kernel with
// the voxel query in ROUNDS of 64 list entries per row, two voxels per lane (packed-f32 tests against a broadcast candidate), six rows
// per wave and round, rows re-packed every round (PF_ROUNDS = 1).  Bit-exact (tests/test_gpu_stages.py, test_gpu_pipeline.py pass with
// it) and ~45 % fewer query instructions, but SLOWER: 375 / 356 / 289 us per launch at the three scales against 282 / 263 / 222 us for
// the three-rows-per-wave form with the same hit recording (K = 5000, P = 1024, tools/bench_stage.py patch): only four of the eight
// waves have work in a round and every round ends in a workgroup barrier, so the patch holds its LDS longer (LABBOOK.md section 2).
// To try it again: copy over buffer-x_amd/csrc/k_patch.hip and build with tools/build_variant.sh.
// k_patch.hip -- patch -> cylindrical voxel features, fused:
//   axis_align   (reference models/patch_embedder.py:122-148; utils/common.py:709-726 cal_Z_axis,
//                 :501-525 RodsRotatFormula, :111-114 l2_norm)
//   normalize    (models/patch_embedder.py:167-170)
//   SPT          (models/patch_embedder.py:150-165; utils/common.py:431-469 sphere_query, :472-498 var_to_invar)
//   pnt_layer + max over the voxel samples (models/patch_embedder.py:26-30, 73-77)
// The reference materialises [K,P,3] x4 temporaries, a [K,420,10,3] gather, the constant voxel grid and 20
// rotation matrices per call.  Two kernels here:
//   patch_axis_kernel      one WAVE per patch: 3x3 covariance (lane-strided fmaf partials + xor butterfly, the
//                          arithmetic contract's "wave order"), binary64 Jacobi eigenvector, Rodrigues -> R [K][9].
//                          The Jacobi is a ~40k-cycle serial chain; as its own kernel with 32 waves per CU it is
//                          hidden by parallelism instead of stalling the other waves of a patch's workgroup.
//   patch_features_kernel  one 512-thread workgroup per patch: the patch lives in LDS (16 B/point: x, y, z, and the
//                          cylindrical radius); candidate lists per (shell, elevation) row by ballot compaction; a
//                          wave owns 3 rows (60 voxels) and scans the rows' lists 8 candidates per step with LDS
//                          broadcast reads; mask, azimuth de-rotation, 3->16 conv + ReLU and the max in registers.
// Output: feat [K][rad][ele*azi][16] in chunk-slot order (bx_chunk_slot).
#include "bx_common.h"
#include <cstdlib>

// experiment switches (tools/build_variant.sh): all on in the shipped build
#ifndef PF_POS
#define PF_POS 1      // a hit is recorded as its POSITION in the row list (one add instead of an 8-way select); looked up when sampled
#endif
#ifndef PF_MAX3
#define PF_MAX3 1     // ReLU + running max as max(mx, acc, 0) (v_max3_f32)
#endif
#ifndef PF_ADDC
#define PF_ADDC 1     // hit bits shifted in by add-with-carry (candidate j -> bit j, candidates tested 7..0)
#endif
#ifndef PF_ROUNDS
#define PF_ROUNDS 1   // the query in rounds of PF_SEG list entries, two voxels per lane, six rows per wave
#endif

namespace {
constexpr int PF_THREADS = 512;
constexpr int PF_WAVES = PF_THREADS / 64;
constexpr int MAX_NS = 16;
constexpr int NROWS = BX_RAD * BX_ELE;                 // 21 (shell, elevation) rows of BX_AZI voxels
constexpr int RPW = (NROWS + PF_WAVES - 1) / PF_WAVES;   // candidate-list rows built per wave (3)
constexpr int PF_SEG = 64;                              // list entries of a row scanned per query round
constexpr int PF_LPR = BX_AZI / 2;                      // query lanes per row: a lane tests the voxels at azimuths a and a + 10
constexpr int PF_RPW = 64 / PF_LPR;                     // rows per wave and query round (6)
static_assert(NROWS == 21 && BX_AZI == 20, "PF_ORDER and the two-voxel query are written for the 3 x 7 x 20 voxel grid");
// the (shell, elevation) rows by expected scan length: the sparse rows around the equator of the outer shells (long lists, never full)
// first, the short sparse rows next, the dense inner-shell rows (full after a few steps) last
__device__ const unsigned char PF_ORDER[NROWS] = {17, 10, 9, 16, 11, 18, 8, 15, 12, 19, 7, 13, 14, 20, 0, 1, 2, 3, 4, 5, 6};

__global__ __launch_bounds__(256) void patch_axis_kernel(const float* __restrict__ patches, int K, int P, float* __restrict__ R_out,
                                                         const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= K) return;
    const float* pp = patches + (size_t)q * P * 3;
    const float cx = pp[(size_t)(P - 1) * 3], cy = pp[(size_t)(P - 1) * 3 + 1], cz = pp[(size_t)(P - 1) * 3 + 2];
    float c00 = 0.f, c01 = 0.f, c02 = 0.f, c11 = 0.f, c12 = 0.f, c22 = 0.f;
    for (int i = lane; i < P; i += 64) {
        const float dx = pp[(size_t)i * 3] - cx, dy = pp[(size_t)i * 3 + 1] - cy, dz = pp[(size_t)i * 3 + 2] - cz;
        c00 = fmaf(dx, dx, c00); c01 = fmaf(dx, dy, c01); c02 = fmaf(dx, dz, c02);
        c11 = fmaf(dy, dy, c11); c12 = fmaf(dy, dz, c12); c22 = fmaf(dz, dz, c22);
    }
    c00 = bx_wave_sum(c00); c01 = bx_wave_sum(c01); c02 = bx_wave_sum(c02);
    c11 = bx_wave_sum(c11); c12 = bx_wave_sum(c12); c22 = bx_wave_sum(c22);
    double A[9] = {(double)c00, (double)c01, (double)c02, (double)c01, (double)c11, (double)c12,
                   (double)c02, (double)c12, (double)c22};
    double V[9], w[3];
    bxd_jacobi3(A, V, w);
    int mi = 0;
    double mv = fabs(w[0]);
    if (fabs(w[1]) < mv) { mv = fabs(w[1]); mi = 1; }
    if (fabs(w[2]) < mv) { mv = fabs(w[2]); mi = 2; }
    float z0 = (float)(mi == 0 ? V[0] : (mi == 1 ? V[1] : V[2]));
    float z1 = (float)(mi == 0 ? V[3] : (mi == 1 ? V[4] : V[5]));
    float z2 = (float)(mi == 0 ? V[6] : (mi == 1 ? V[7] : V[8]));
    float sdot = ((-z0) * cx + (-z1) * cy) + (-z2) * cz;
    if (sdot < 0.0f) { z0 = -z0; z1 = -z1; z2 = -z2; }
    float nz = sqrtf((z0 * z0 + z1 * z1) + z2 * z2);
    z0 = z0 / nz; z1 = z1 / nz; z2 = z2 / nz;
    float c0 = z1, c1 = -z0, c2 = 0.0f;
    float na = sqrtf((z0 * z0 + z1 * z1) + z2 * z2);
    float nae = na > 1e-8f ? na : 1e-8f;
    float cosv = z2 / nae;
    float theta = (float)bxd_acos((double)cosv);
    double sd, cd;
    bxd_sincos((double)theta, &sd, &cd);
    float sn = (float)sd, cs = (float)cd;
    float nc = sqrtf((c0 * c0 + c1 * c1) + c2 * c2);
    float nce = nc > 1e-12f ? nc : 1e-12f;
    c0 = c0 / nce; c1 = c1 / nce; c2 = c2 / nce;
    float Rx[9] = {0.0f, -c2, c1, c2, 0.0f, -c0, -c1, c0, 0.0f};
    float Rx2[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Rx2[i * 3 + j] = fmaf(Rx[i * 3 + 2], Rx[2 * 3 + j], fmaf(Rx[i * 3 + 1], Rx[1 * 3 + j], Rx[i * 3 + 0] * Rx[0 * 3 + j]));
    float omc = 1.0f - cs;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float I = (i == j) ? 1.0f : 0.0f;
                float rr = (I + sn * Rx[i * 3 + j]) + omc * Rx2[i * 3 + j];
                R_out[(size_t)q * 9 + j * 3 + i] = rr;  // transpose(-1,-2)
            }
    }
}

__global__ __launch_bounds__(PF_THREADS) void patch_features_kernel(
    const float* __restrict__ patches, int K, int P, const double* __restrict__ radius, int aligned,
    const float* __restrict__ centres, const float* __restrict__ rowc, const float* __restrict__ rot, int nsample, float voxel_r,
    const float* __restrict__ pnt_w, const float* __restrict__ pnt_b, float* __restrict__ R_out, float* __restrict__ feat,
    const int32_t* __restrict__ skip, int cap, long long* __restrict__ dbg)
{
    if (skip && *skip) return;
    // optional cycle stamps (BX_BALL_DEBUG): {t0, normalised, row lists, query of wave 0, conv+store of wave 0}
    long long t0 = 0;
    const bool tr = dbg != nullptr && (blockIdx.x % 79) == 0 && blockIdx.x / 79 < 60 && threadIdx.x == 0;
    long long* td = dbg + (blockIdx.x / 79) * 8;
    if (tr) { t0 = __builtin_readcyclecounter(); td[0] = t0; }
#define PF_TR(k) do { if (tr) td[k] = __builtin_readcyclecounter() - t0; } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sp = reinterpret_cast<float4*>(smem);                             // [P + 1] x, y, z, sqrt(x^2 + y^2); entry P = a point far away
    unsigned short* shit = reinterpret_cast<unsigned short*>(sp + P + 1);    // [nsample][BX_VOX] (+ 4 when nsample is odd: 16-byte lists)
    int* rlen = reinterpret_cast<int*>(shit + (((size_t)nsample * BX_VOX + 7) & ~(size_t)7));   // [32] candidates per (shell, elevation) row
    unsigned short* rlist = reinterpret_cast<unsigned short*>(rlen + 32);    // [NROWS][cap] candidate point indices, ascending
    unsigned short* far8 = rlist + (size_t)NROWS * cap + 8;                   // eight copies of index P (behind the 16-byte read slack)
    unsigned char* vcnt = reinterpret_cast<unsigned char*>(far8 + 8);         // [BX_VOX (-> 448)] hits recorded per voxel
    int* rowfull = reinterpret_cast<int*>(vcnt + 448);                        // [32] every voxel of the row has its nsample hits

    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pp = patches + (size_t)q * P * 3;
    const float des_r = (float)(*radius);
    const float cx = pp[(size_t)(P - 1) * 3], cy = pp[(size_t)(P - 1) * 3 + 1], cz = pp[(size_t)(P - 1) * 3 + 2];
    float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (!aligned) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = R_out[(size_t)q * 9 + i];
    } else if (tid < 9) {
        R_out[(size_t)q * 9 + tid] = (tid % 4 == 0) ? 1.0f : 0.0f;
    }

    // ---- centre on the keypoint, rotate (delta @ R), normalise by the scale radius
    for (int i = tid; i < P; i += PF_THREADS) {
        const float x = pp[(size_t)i * 3] - cx, y = pp[(size_t)i * 3 + 1] - cy, z = pp[(size_t)i * 3 + 2] - cz;
        float nx = x, ny = y, nzc = z;
        if (!aligned) {
            nx = fmaf(z, R[6], fmaf(y, R[3], x * R[0]));
            ny = fmaf(z, R[7], fmaf(y, R[4], x * R[1]));
            nzc = fmaf(z, R[8], fmaf(y, R[5], x * R[2]));
        }
        const float px = nx / des_r, py = ny / des_r;
        sp[i] = make_float4(px, py, nzc / des_r, sqrtf(px * px + py * py));
    }
    // the far point: a list entry P never passes the distance test, so list tails and idle lanes need no per-candidate bounds checks
    if (tid == 0) sp[P] = make_float4(1.0e30f, 1.0e30f, 1.0e30f, 1.0e30f);
    if (tid < 8) far8[tid] = (unsigned short)P;
    if (tid < 448) vcnt[tid] = 0;
    if (tid < 32) rowfull[tid] = 0;
    __syncthreads();
    PF_TR(1);

    const float vr2 = voxel_r * voxel_r;

    // ---- candidate lists per (shell, elevation) row.  The 20 voxel centres of a row lie on the circle {radius R_c,
    //      height z_c}; a point can only be within voxel_r of one of them if its distance to that CIRCLE is, i.e.
    //      (R_p - R_c)^2 + (p_z - z_c)^2 < voxel_r^2 (exact inequality; a 1e-4 margin covers fp32 rounding and the
    //      fp32-rounded centres).  A wave builds the lists of its (up to 3) rows in one sweep over the patch with
    //      ballot compaction, so a list is in ascending point order and "the first voxel_sample hits in patch order"
    //      is simply a scan of the row's list.
    {
        const float cthr = vr2 * 1.0001f + 1.0e-6f;
        float Rc[RPW], zc[RPW];
        int base[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave + r * PF_WAVES;
            Rc[r] = row < NROWS ? rowc[row * 2] : 1.0e30f;     // rows beyond the table never pass
            zc[r] = row < NROWS ? rowc[row * 2 + 1] : 0.f;
            base[r] = 0;
        }
        for (int k0 = 0; k0 < P; k0 += 64) {
            const int k = k0 + lane;
            float4 d = make_float4(0.f, 0.f, 1.0e30f, 0.f);
            if (k < P) d = sp[k];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave + r * PF_WAVES;
                if (row >= NROWS) continue;                  // wave-uniform: waves 5..7 own two rows
                const float t1 = d.w - Rc[r], t2 = d.z - zc[r];
                const bool pass = (t1 * t1 + t2 * t2) < cthr;
                const unsigned long long m = __ballot(pass);
                // base + the number of passing lanes below this one (v_mbcnt_lo / _hi carry the base in)
                const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base[r]));
                if (pass && pos < cap) rlist[(size_t)row * cap + pos] = (unsigned short)k;
                base[r] += __popcll(m);
            }
        }
        // a row with more candidates than its list holds (rare: e.g. the padded points at the keypoint for the inner
        // shell) is scanned over the whole patch instead: len = -1
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave + r * PF_WAVES;
                if (row < NROWS) rlen[row] = base[r] <= cap ? base[r] : -1;
            }
        }
        // pad every list to a multiple of 8 entries with the far point (cap is a multiple of 8)
        if (lane < 8) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave + r * PF_WAVES;
                if (row < NROWS && base[r] <= cap && base[r] + lane < ((base[r] + 7) & ~7))
                    rlist[(size_t)row * cap + base[r] + lane] = (unsigned short)P;
            }
        }
    }
    __syncthreads();
    PF_TR(2);

#if PF_ROUNDS
    // ---- voxel query in ROUNDS of PF_SEG list entries per row.  A lane tests TWO voxels of one row (azimuths a and a + 10) against the
    //      row's candidates -- the packed-f32 VALU ops take the candidate's coordinate as a broadcast operand, so a (candidate, voxel)
    //      test costs half the instructions, LDS reads and index unpacking of the one-voxel form -- and a wave holds up to 6 rows (10
    //      lanes each).  A round hands the wave 6 of the rows that still have unscanned candidates AND an unfilled voxel, in a static
    //      order that puts rows of similar scan length together (sparse outer rows first), so lanes whose row is done do not ride along
    //      with the longest row of a fixed group: the rounds after the first hold only the long rows, packed.  The hit count of a
    //      voxel lives in LDS between rounds; hits are appended in list order because a row's segments are scanned in round order.
    {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int grp = lane / PF_LPR, sub = lane - grp * PF_LPR;       // grp 0..5 = a row of this round, 6 = lanes 60..63 (idle)
        const int my_row = lane < NROWS ? (int)PF_ORDER[lane] : 0;
        for (int round = 0; ; ++round) {
            const int start = round * PF_SEG;
            bool actl = false;
            if (lane < NROWS) {
                const int l_ = rlen[my_row];
                actl = start < (l_ < 0 ? P : l_) && rowfull[my_row] == 0;
            }
            unsigned m = (unsigned)__ballot(actl);                      // the same value in every wave
            if (m == 0u) break;
            if (wave * PF_RPW < __popc(m)) {
                for (int i = 0; i < wave * PF_RPW; ++i) m &= m - 1u;    // scalar: skip the rows of the waves before this one
                int slot = -1;
#pragma unroll
                for (int g = 0; g < PF_RPW; ++g) {
                    const int b_ = m != 0u ? __ffs((int)m) - 1 : -1;
                    m &= m - 1u;
                    slot = grp == g ? b_ : slot;
                }
                const bool act = slot >= 0;
                const int row = __shfl(my_row, act ? slot : 0);
                const int l_ = rlen[row];
                const bool direct = l_ < 0;                              // the list overflowed: the row is scanned over the patch itself
                const int L = direct ? P : ((l_ + 7) & ~7);              // lists are padded to a multiple of 8 with the far point
                const int end = act ? min(L, start + PF_SEG) : 0;
                const int v0 = row * BX_AZI + sub, v1 = v0 + PF_LPR;
                f2 qx = {0.f, 0.f}, qy = {0.f, 0.f}, qz = {0.f, 0.f};
                int c0 = nsample, c1 = nsample;
                if (act) {
                    qx = f2{centres[v0 * 3], centres[v1 * 3]};
                    qy = f2{centres[v0 * 3 + 1], centres[v1 * 3 + 1]};
                    qz = f2{centres[v0 * 3 + 2], centres[v1 * 3 + 2]};
                    c0 = vcnt[v0]; c1 = vcnt[v1];
                }
                const unsigned short* rl = rlist + (size_t)row * cap;
                // the hits of a step in list order, as positions in the list (point indices for a directly scanned row)
#define PF_RECORD(hm, cnt, v)                                                  \
                while (hm != 0u && cnt < nsample) {                             \
                    const int j_ = __ffs((int)hm) - 1;                          \
                    shit[cnt * BX_VOX + v] = (unsigned short)(i0 + j_);         \
                    ++cnt;                                                      \
                    hm &= hm - 1u;                                              \
                }
                const bool anyd = __any(direct && act);
                for (int i0 = start; ; i0 += 8) {
                    const bool live = i0 < end && (c0 < nsample || c1 < nsample);
                    if (!__any(live)) break;
                    // a lane that is done reads eight far points: no per-candidate bounds checks
                    const uint4 kq = *reinterpret_cast<const uint4*>(live && !direct ? rl + i0 : far8);
                    int ks[8];
                    ks[0] = kq.x & 0xffff; ks[1] = kq.x >> 16; ks[2] = kq.y & 0xffff; ks[3] = kq.y >> 16;
                    ks[4] = kq.z & 0xffff; ks[5] = kq.z >> 16; ks[6] = kq.w & 0xffff; ks[7] = kq.w >> 16;
                    if (anyd) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) ks[j] = (direct && live) ? min(i0 + j, P) : ks[j];
                    }
                    unsigned hm0 = 0, hm1 = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 d = sp[ks[j]];
                        const f2 dx = qx - d.x, dy = qy - d.y, dz = qz - d.z;
                        const f2 dd = (dx * dx + dy * dy) + dz * dz;
                        if (dd.x < vr2) hm0 |= 1u << j;
                        if (dd.y < vr2) hm1 |= 1u << j;
                    }
                    PF_RECORD(hm0, c0, v0)
                    PF_RECORD(hm1, c1, v1)
                }
#undef PF_RECORD
                if (act) { vcnt[v0] = (unsigned char)c0; vcnt[v1] = (unsigned char)c1; }
                // a row whose 20 voxels are full is out of the later rounds
                const unsigned long long fb = __ballot(c0 >= nsample && c1 >= nsample);
                if (act && sub == 0 && ((fb >> (grp * PF_LPR)) & ((1ULL << PF_LPR) - 1ULL)) == ((1ULL << PF_LPR) - 1ULL)) rowfull[row] = 1;
            }
            __syncthreads();
        }
    }
    PF_TR(3);
#endif
#if !PF_ROUNDS
    // ---- voxel query: a wave owns 3 whole rows (60 voxels, lanes 60..63 idle); lanes of one row read the same list
    //      entries and the same points (LDS broadcast), 8 candidates per step so that the dependent LDS reads of a step
    //      overlap; each lane tests its own centre and keeps the first `nsample` hits (ascending point order)
#endif
    for (int task = wave; task < NROWS / 3; task += PF_WAVES) {
        const int rsub = lane / BX_AZI;                   // 0..2, 3 for the idle lanes
        const bool act = rsub < 3;
        const int row = task * 3 + (act ? rsub : 0);
        const int v = row * BX_AZI + (lane - rsub * BX_AZI);
#if !PF_ROUNDS
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (act) { qx = centres[v * 3]; qy = centres[v * 3 + 1]; qz = centres[v * 3 + 2]; }
        const int rl_len = rlen[row];
        const bool full = rl_len < 0;
        const int len = act ? (full ? P : rl_len) : 0;
        const unsigned short* rl = rlist + (size_t)row * cap;
        int cnt = 0;
        // the hits of a step, in list order, up to nsample (a macro: a lambda taking ks[] by reference sends the array to scratch)
#if PF_POS
#define PF_RECORD(hm, ks)                                                       \
        while (hm != 0u && cnt < nsample) {                                     \
            const int j_ = __ffs((int)hm) - 1;                                  \
            shit[cnt * BX_VOX + v] = (unsigned short)(i0 + j_);                 \
            ++cnt;                                                              \
            hm &= hm - 1u;                                                      \
        }
#else
#define PF_RECORD(hm, ks)                                                       \
        while (hm != 0u && cnt < nsample) {                                     \
            const int j_ = __ffs((int)hm) - 1;                                  \
            int k_ = ks[0];                                                     \
            _Pragma("unroll") for (int u = 1; u < 8; ++u) k_ = (j_ == u) ? ks[u] : k_; \
            shit[cnt * BX_VOX + v] = (unsigned short)k_;                        \
            ++cnt;                                                              \
            hm &= hm - 1u;                                                      \
        }
#endif
        if (!__any(full)) {
            // the common case: every row of the wave has a list.  Lists are padded with the far point and a lane whose list has
            // ended reads eight far points, so a candidate costs its index unpack, one LDS read, the distance and one compare
            for (int i0 = 0; ; i0 += 8) {
                if (__all(cnt >= nsample || i0 >= len)) break;
                const uint4 kq = *reinterpret_cast<const uint4*>(i0 < len ? rl + i0 : far8);
                int ks[8];
                ks[0] = kq.x & 0xffff; ks[1] = kq.x >> 16; ks[2] = kq.y & 0xffff; ks[3] = kq.y >> 16;
                ks[4] = kq.z & 0xffff; ks[5] = kq.z >> 16; ks[6] = kq.w & 0xffff; ks[7] = kq.w >> 16;
                unsigned hm = 0;
#if PF_ADDC
                float dd[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 d = sp[ks[j]];
                    const float dx = qx - d.x, dy = qy - d.y, dz = qz - d.z;
                    dd[j] = (dx * dx + dy * dy) + dz * dz;
                }
#pragma unroll
                for (int j = 7; j >= 0; --j)      // hm = 2 hm + (dd < vr2): candidate j ends at bit j
                    asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hm) : "v"(dd[j]), "v"(vr2) : "vcc");
#else
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 d = sp[ks[j]];
                    const float dx = qx - d.x, dy = qy - d.y, dz = qz - d.z;
                    const float dd = (dx * dx + dy * dy) + dz * dz;
                    if (dd < vr2) hm |= 1u << j;
                }
#endif
                PF_RECORD(hm, ks)
            }
        } else {
            for (int i0 = 0; ; i0 += 8) {
                if (__all(cnt >= nsample || i0 >= len)) break;
                const uint4 kq = *reinterpret_cast<const uint4*>(rl + i0);
                int ks[8];
                ks[0] = kq.x & 0xffff; ks[1] = kq.x >> 16; ks[2] = kq.y & 0xffff; ks[3] = kq.y >> 16;
                ks[4] = kq.z & 0xffff; ks[5] = kq.z >> 16; ks[6] = kq.w & 0xffff; ks[7] = kq.w >> 16;
                unsigned hm = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int k = full ? i0 + j : ks[j];
                    k = k < P ? k : P - 1;
                    ks[j] = k;
                    const float4 d = sp[k];
                    const float dx = qx - d.x, dy = qy - d.y, dz = qz - d.z;
                    const float dd = (dx * dx + dy * dy) + dz * dz;
                    if (dd < vr2 && i0 + j < len) hm |= 1u << j;
                }
                PF_RECORD(hm, ks)
            }
        }
#undef PF_RECORD
        PF_TR(3);
#else
        const int rl_len = rlen[row];
        const bool full = rl_len < 0;
        const unsigned short* rl = rlist + (size_t)row * cap;
        const int cnt = act ? (int)vcnt[v] : 0;
#endif
        if (!act) continue;
        const int a = v % BX_AZI;
        const float r00 = rot[a * 4], r01 = rot[a * 4 + 1], r10 = rot[a * 4 + 2], r11 = rot[a * 4 + 3];
#if PF_POS
        // a recorded hit is a position in the row's list (the point index itself for a row scanned over the whole patch)
#define PF_HIT(j) (full ? (int)shit[(j) * BX_VOX + v] : (int)rl[shit[(j) * BX_VOX + v]])
#else
#define PF_HIT(j) ((int)shit[(j) * BX_VOX + v])
#endif
        const int first = cnt > 0 ? PF_HIT(0) : 0;
        float mx[16];
#if PF_MAX3
#pragma unroll
        for (int c = 0; c < 16; ++c) mx[c] = 0.0f;         // ReLU outputs are >= +0: a running max that starts at +0 is the same max
#endif
        for (int j = 0; j < nsample; ++j) {
            int id = j < cnt ? PF_HIT(j) : first;
            float mask = (j > 0 && id == first) ? 1.0f : 0.0f;
            if (j == 0 && first == 0) mask = 1.0f;
            float om = 1.0f - mask;
            float4 d = sp[id];
            float x = d.x * om, y = d.y * om, z = d.z * om;
            float nx = fmaf(y, r01, x * r00);
            float ny = fmaf(y, r11, x * r10);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float acc = pnt_b[c];
                acc = fmaf(pnt_w[c * 3 + 0], nx, acc);
                acc = fmaf(pnt_w[c * 3 + 1], ny, acc);
                acc = fmaf(pnt_w[c * 3 + 2], z, acc);
#if PF_MAX3
                mx[c] = fmaxf(fmaxf(mx[c], acc), 0.0f);
#else
                acc = acc > 0.0f ? acc : 0.0f;
                mx[c] = (j == 0 || acc > mx[c]) ? acc : mx[c];
#endif
            }
        }
        const int s = v / BX_EA, pos = v % BX_EA;
        float4* fo = reinterpret_cast<float4*>(feat + (((size_t)q * BX_RAD + s) * BX_EA + pos) * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) fo[u] = make_float4(mx[u], mx[4 + u], mx[8 + u], mx[12 + u]);
        PF_TR(4);
#undef PF_HIT
    }
}
}  // namespace

int bxk_patch_features(bx_ctx* c, hipStream_t s, const float* patches, int K, int P, const double* radius, int aligned,
                       float* R_out, float* feat_out)
{
    if (K <= 0) return BX_OK;
    const int ns = c->p.voxel_sample;
    if (ns < 1 || ns > MAX_NS || P < 2 || P > 8192) { bx_set_error("bxk_patch_features: voxel_sample=%d P=%d unsupported", ns, P); return BX_ERR_ARG; }
    int cap = ((P / 2 + 7) / 8) * 8 + 8;                      // row-list capacity (multiple of 8, one 16-byte read of slack)
    size_t lds = (size_t)(P + 1) * 16 + (((size_t)ns * BX_VOX + 7) & ~(size_t)7) * 2 + 128 + (size_t)NROWS * cap * 2 + 16 + 16   // + far point, + far8
                 + 448 + 128;                                                                                                    // + vcnt, rowfull
    if (lds > 160 * 1024) { bx_set_error("bxk_patch_features: P=%d needs %zu B of LDS", P, lds); return BX_ERR_ARG; }
    if (lds > 64 * 1024 && !c->patch_attr_set) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(patch_features_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        c->patch_attr_set = 1;
    }
    const float voxel_r = (float)(c->p.delta / (double)c->p.rad_n);
    if (!aligned) hipLaunchKernelGGL(patch_axis_kernel, dim3((K + 3) / 4), dim3(256), 0, s, patches, K, P, R_out, c->skip);
    hipLaunchKernelGGL(patch_features_kernel, dim3(K), dim3(PF_THREADS), lds, s, patches, K, P, radius, aligned, c->d_centres,
                       c->d_rowc, c->d_rot, ns, voxel_r, c->d_pnt_w, c->d_pnt_b, R_out, feat_out, c->skip, cap,
                       getenv("BX_BALL_DEBUG") ? c->ball_dbg : nullptr);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
