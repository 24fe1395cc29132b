// k_fps.hip -- furthest point sampling + keypoint gather for gfx950.
//
// Replaces pointnet2_ops.furthest_point_sample / gather_operation (reference call sites
// models/BUFFERX.py:286-290, 338-346; semantics SURVEY.md A.1).  The reference runs ONE thread block per
// cloud and re-runs FPS (1+S) times per cloud; here FPS runs once per cloud (FPS(2000) is a prefix of
// FPS(num_fps)) and both clouds of a pair run in one launch.
//
// Design (latency-bound: m strictly dependent arg-max steps):
//   * each workgroup (1024 threads = 16 waves) keeps PPT points per thread -- xyz AND the running
//     min-distance -- in VGPRs for the whole kernel: no memory traffic inside the iteration loop.
//     16 K points fill half of a CU's vector register file, so a cloud of N points is split over
//     G = ceil(N / 16384) workgroups (CUs).
//   * per iteration: thread-local scan (strict '>' == upstream per-thread rule), 64-bit key
//     {fp32 bits of d, ~tie-break} max-reduced over the wave with lane shuffles, one LDS record per
//     wave, one s_barrier.  The tie-break reproduces the upstream block-tree rule for T = 512
//     (lower k mod T, then lower k) -- see oracle/bx_oracle.c bxo_fps.
//   * G > 1: workgroups exchange their winner {key, x, y, z} through 8-byte {epoch, value} granules
//     written with agent-scope relaxed (sc1, write-through) stores and polled with agent-scope relaxed
//     loads -- the "R2 granule" hand-off of the CDNA4 guide: no fence, placement independent, spins
//     bounded.  Slots are double-buffered by iteration parity and zeroed by a memset node before launch.  All G workgroups of a
//     cloud must be resident at once (they wait for each other): G <= 64 per cloud, 2 clouds, 256 CUs.
#include "bx_common.h"

namespace {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / 64;
constexpr int FPS_MAX_G = 64;            // 64 workgroups x 16 384 points = 1 048 576 points per cloud (the neighbour bitmap's limit too)
constexpr int FPS_NR = (5 * FPS_MAX_G + 63) / 64;   // polling rounds: one granule per lane and round
constexpr int FPS_MAX_CLOUDS = 2;
constexpr unsigned FPS_SPIN_LIMIT = 1u << 24;

struct FpsArgs {
    const float* xyz[FPS_MAX_CLOUDS];
    int32_t* idx_out[FPS_MAX_CLOUDS];
    float* kpts_out[FPS_MAX_CLOUDS];
    int n[FPS_MAX_CLOUDS];
    int G[FPS_MAX_CLOUDS];
    int wg_start[FPS_MAX_CLOUDS + 1];
    int nclouds;
    int j0, m;                  // iterations [j0, m) of this launch; j0 > 0 resumes from td_state + kpts_out[j0 - 1]
    int save;                   // store the running min-distances into td_state at the end (another launch follows)
    float* td_state[FPS_MAX_CLOUDS];   // [n] running min-distance of every point between the launches of a tiled run
    unsigned long long* slots;  // [cloud][parity 2][FPS_MAX_G][5] granules
    int32_t* err_flag;
};

struct Rec {
    long long key;
    float x, y, z;
};

// max over the 64 lanes on the DPP network (result valid in lane 63, returned broadcast through v_readlane)
__device__ __forceinline__ int wave_max_i32(int v)
{
    const int id = (int)0x80000000;
    int o;
    o = __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;   // row_shr:1
    o = __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;   // row_shr:2
    o = __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;   // row_shr:4
    o = __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;   // row_shr:8
    o = __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;   // row_bcast:15
    o = __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit key max as two 32-bit DPP reductions: high word (signed: fp32 bits of the distance, -1.0f = "no candidate"),
// then the low word (unsigned) among the lanes that hold the maximal high word
__device__ __forceinline__ long long wave_max_key(long long key)
{
    const int hi = (int)(key >> 32);
    const unsigned lo = (unsigned)((unsigned long long)key & 0xffffffffu);
    const int mh = wave_max_i32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    return (long long)(((unsigned long long)(unsigned)mh << 32) | ml);
}

template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(FpsArgs a)
{
    __shared__ long long s_key[2][FPS_WAVES];
    __shared__ float s_xyz[2][FPS_WAVES][3];
    __shared__ long long s_fkey[2];
    __shared__ float s_fxyz[2][3];

    int cloud = 0;
    if (a.nclouds > 1 && (int)blockIdx.x >= a.wg_start[1]) cloud = 1;
    const int g = blockIdx.x - a.wg_start[cloud];
    const int G = a.G[cloud];
    const int n = a.n[cloud];
    const float* __restrict__ xyz = a.xyz[cloud];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int base = g * (FPS_THREADS * PPT);

    int T = 1, lt = 0;
    while (T * 2 <= n && T * 2 <= 512) { T *= 2; ++lt; }
    const unsigned tmask = (unsigned)(T - 1);

    float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int k = base + i * FPS_THREADS + t;
        if (k < n) {
            px[i] = xyz[(size_t)k * 3 + 0];
            py[i] = xyz[(size_t)k * 3 + 1];
            pz[i] = xyz[(size_t)k * 3 + 2];
            float mag = (px[i] * px[i] + py[i] * py[i]) + pz[i] * pz[i];
            td[i] = (mag <= 1e-3f) ? -1.0f : 1e10f;  // -1 marks "never a candidate" (upstream `continue`)
            if (a.j0 > 0) td[i] = a.td_state[cloud][k];
        } else {
            px[i] = 0.f; py[i] = 0.f; pz[i] = 0.f; td[i] = -1.0f;
        }
    }
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    if (a.j0 > 0) {   // the previous launch of this stream wrote the last keypoint (kernel boundary: visible)
        const float* lk = a.kpts_out[cloud] + (size_t)(a.j0 - 1) * 3;
        cx = lk[0]; cy = lk[1]; cz = lk[2];
    } else if (g == 0 && t == 0) {
        a.idx_out[cloud][0] = 0;
        if (a.kpts_out[cloud]) { a.kpts_out[cloud][0] = cx; a.kpts_out[cloud][1] = cy; a.kpts_out[cloud][2] = cz; }
    }
    unsigned long long* slots = a.slots + (size_t)cloud * 2 * FPS_MAX_G * 5;

    for (int j = a.j0 > 0 ? a.j0 : 1; j < a.m; ++j) {
        const int par = j & 1;
        float bd = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            float dx = px[i] - cx, dy = py[i] - cy, dz = pz[i] - cz;
            float d = (dx * dx + dy * dy) + dz * dz;
            float d2 = fminf(d, td[i]);
            td[i] = d2;
            bool better = d2 > bd;
            bd = better ? d2 : bd;
            bi = better ? i : bi;
            bx = better ? px[i] : bx;
            by = better ? py[i] : by;
            bz = better ? pz[i] : bz;
        }
        unsigned k = (unsigned)(base + bi * FPS_THREADS + t);
        unsigned tb = ((k & tmask) << 23) | (k >> lt);
        long long key = ((long long)__float_as_int(bd) << 32) | (long long)(unsigned)(~tb);
        // wave max (signed 64-bit) on the DPP network
        const long long wk = wave_max_key(key);
        if (key == wk) {  // keys are unique per thread (they embed the point index)
            s_key[par][wave] = wk;
            s_xyz[par][wave][0] = bx; s_xyz[par][wave][1] = by; s_xyz[par][wave][2] = bz;
        }
        __syncthreads();
        long long fk;
        float fx, fy, fz;
        if (G == 1) {
            fk = s_key[par][0]; fx = s_xyz[par][0][0]; fy = s_xyz[par][0][1]; fz = s_xyz[par][0][2];
#pragma unroll
            for (int w = 1; w < FPS_WAVES; ++w) {
                long long kk = s_key[par][w];
                bool b = kk > fk;
                fk = b ? kk : fk;
                fx = b ? s_xyz[par][w][0] : fx;
                fy = b ? s_xyz[par][w][1] : fy;
                fz = b ? s_xyz[par][w][2] : fz;
            }
        } else {
            if (wave == 0) {
                // combine the 16 wave records
                long long k0 = lane < FPS_WAVES ? s_key[par][lane] : (long long)0x8000000000000000LL;
                const long long mk = wave_max_key(k0);
                // the (unique, or lowest) lane holding the max publishes this workgroup's record
                unsigned long long bal = __ballot(lane < FPS_WAVES && k0 == mk);
                int src = __ffsll((long long)bal) - 1;
                unsigned long long* my = slots + ((size_t)par * FPS_MAX_G + g) * 5;
                if (lane == src) {
                    unsigned ep = (unsigned)j;
                    unsigned v[5] = {(unsigned)((unsigned long long)mk >> 32), (unsigned)((unsigned long long)mk & 0xffffffffu),
                                     __float_as_uint(s_xyz[par][lane][0]), __float_as_uint(s_xyz[par][lane][1]),
                                     __float_as_uint(s_xyz[par][lane][2])};
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        __hip_atomic_store(my + q, ((unsigned long long)ep << 32) | v[q], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
                // poll all G records (one granule per lane and round; rounds beyond 5G granules are skipped: G is uniform)
                unsigned val[FPS_NR];
                bool fail = false;
#pragma unroll
                for (int r = 0; r < FPS_NR; ++r) {
                    val[r] = 0;
                    if (r * 64 >= 5 * G) continue;
                    int q = lane + r * 64;
                    bool act = q < 5 * G;
                    unsigned long long* gp = slots + (size_t)par * FPS_MAX_G * 5 + q;
                    unsigned spins = 0;
                    while (true) {
                        bool ok = true;
                        if (act) {
                            unsigned long long x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            val[r] = (unsigned)x;
                            ok = (unsigned)(x >> 32) == (unsigned)j;
                        }
                        if (__all(ok)) break;
                        if (++spins > FPS_SPIN_LIMIT) { fail = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (fail && lane == 0) atomicOr(a.err_flag, 1);
                // gather record w: granules 5w..5w+4 -> lanes; reduce
                long long bestk = (long long)0x8000000000000000LL;
                float ox = 0.f, oy = 0.f, oz = 0.f;
                auto granule = [&](int qq) -> unsigned {    // wave-uniform index: v_readlane, no LDS crossbar
                    if (5 * G <= 64) return (unsigned)__builtin_amdgcn_readlane((int)val[0], qq);   // G <= 12: one polling round
                    unsigned v = 0;
#pragma unroll
                    for (int r = 0; r < FPS_NR; ++r)
                        if ((qq >> 6) == r) v = (unsigned)__builtin_amdgcn_readlane((int)val[r], qq & 63);   // uniform branch
                    return v;
                };
                for (int w = 0; w < G; ++w) {
                    const int q0 = 5 * w;
                    const unsigned hi = granule(q0), lo = granule(q0 + 1);
                    const unsigned ux = granule(q0 + 2), uy = granule(q0 + 3), uz = granule(q0 + 4);
                    long long kk = (long long)(((unsigned long long)hi << 32) | lo);
                    if (kk > bestk) { bestk = kk; ox = __uint_as_float(ux); oy = __uint_as_float(uy); oz = __uint_as_float(uz); }
                }
                if (lane == 0) {
                    s_fkey[par] = bestk;
                    s_fxyz[par][0] = ox; s_fxyz[par][1] = oy; s_fxyz[par][2] = oz;
                }
            }
            __syncthreads();
            fk = s_fkey[par]; fx = s_fxyz[par][0]; fy = s_fxyz[par][1]; fz = s_fxyz[par][2];
        }
        int old;
        if (fk < 0) {  // no candidate anywhere (all points within 1e-3 of the origin): upstream yields index 0
            old = 0; fx = xyz[0]; fy = xyz[1]; fz = xyz[2];
        } else {
            unsigned tbw = ~(unsigned)((unsigned long long)fk & 0xffffffffu);
            old = (int)(((tbw & 0x7fffffu) << lt) | (tbw >> 23));
        }
        cx = fx; cy = fy; cz = fz;
        if (g == 0 && t == 0) {
            a.idx_out[cloud][j] = old;
            if (a.kpts_out[cloud]) {
                a.kpts_out[cloud][(size_t)j * 3 + 0] = cx;
                a.kpts_out[cloud][(size_t)j * 3 + 1] = cy;
                a.kpts_out[cloud][(size_t)j * 3 + 2] = cz;
            }
        }
    }
    if (a.save) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            int k = base + i * FPS_THREADS + t;
            if (k < n) a.td_state[cloud][k] = td[i];
        }
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ pts, const int32_t* __restrict__ idx, int n, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        size_t s = (size_t)idx[i] * 3;
        out[(size_t)i * 3 + 0] = pts[s + 0];
        out[(size_t)i * 3 + 1] = pts[s + 1];
        out[(size_t)i * 3 + 2] = pts[s + 2];
    }
}

}  // namespace

// Iterations [j0, j1) of an m-point run.  A run may be cut into consecutive launches on ONE stream (latency mode of
// bx_register_pair: the descriptors of the first keypoints are computed while the later ones are still being sampled): launches
// with j1 < m leave the running min-distances in c->fps_td, launches with j0 > 0 pick them up.  The result is the single-launch
// one bit for bit (same per-thread state, same epochs in the exchange slots).
int bxk_fps_range(bx_ctx* c, hipStream_t s, const float* const* xyz, const int* n, int nclouds, int j0, int j1, int m,
                  int32_t* const* idx_out, float* const* kpts_out)
{
    if (nclouds < 1 || nclouds > FPS_MAX_CLOUDS) { bx_set_error("bxk_fps: nclouds"); return BX_ERR_ARG; }
    if (j0 < 0 || j1 <= j0 || j1 > m) { bx_set_error("bxk_fps: range [%d, %d) of %d", j0, j1, m); return BX_ERR_ARG; }
    if ((j0 > 0 || j1 < m) && (!kpts_out || !c->fps_td[0])) { bx_set_error("bxk_fps: a tiled run needs kpts_out and the td state"); return BX_ERR_ARG; }
    FpsArgs a{};
    a.nclouds = nclouds;
    a.j0 = j0;
    a.m = j1;
    a.save = j1 < m ? 1 : 0;
    a.slots = c->fps_slots;
    a.err_flag = c->err_flag;
    int ppt = 4;
    int total = 0;
    int nmax = 0;
    for (int i = 0; i < nclouds; ++i) nmax = n[i] > nmax ? n[i] : nmax;
    // fewest workgroups first (cross-CU exchange is the expensive part), then the smallest PPT that fits
    int Gneed = (nmax + FPS_THREADS * 16 - 1) / (FPS_THREADS * 16);
    if (Gneed < 1) Gneed = 1;
    if (Gneed > FPS_MAX_G) { bx_set_error("bxk_fps: cloud of %d points exceeds %d", nmax, FPS_MAX_G * FPS_THREADS * 16); return BX_ERR_ARG; }
    int per_wg = (nmax + Gneed - 1) / Gneed;
    ppt = per_wg <= FPS_THREADS * 4 ? 4 : (per_wg <= FPS_THREADS * 8 ? 8 : 16);
    for (int i = 0; i < nclouds; ++i) {
        if (n[i] < 1) { bx_set_error("bxk_fps: empty cloud"); return BX_ERR_ARG; }
        a.xyz[i] = xyz[i];
        a.n[i] = n[i];
        a.idx_out[i] = idx_out[i];
        a.kpts_out[i] = kpts_out ? kpts_out[i] : nullptr;
        a.td_state[i] = c->fps_td[i];
        a.G[i] = (n[i] + FPS_THREADS * ppt - 1) / (FPS_THREADS * ppt);
        a.wg_start[i] = total;
        total += a.G[i];
    }
    a.wg_start[nclouds] = total;
    // epochs continue across the launches of a tiled run: the slots are cleared once, in front of the first one
    if (j0 == 0) BX_HIP(hipMemsetAsync(c->fps_slots, 0, sizeof(unsigned long long) * FPS_MAX_CLOUDS * 2 * FPS_MAX_G * 5, s));
    static int hog = -1;
    if (hog < 0) { const char* e = getenv("BX_FPS_HOG"); hog = e ? atoi(e) : 0; }
    size_t lds = 0;
    if (hog > 0 && (j0 > 0 || j1 < m)) {
        lds = (size_t)hog * 1024;
        static bool attr = false;
        if (!attr) {
            BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
            BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
            BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
            attr = true;
        }
    }
    if (ppt == 4) hipLaunchKernelGGL(fps_kernel<4>, dim3(total), dim3(FPS_THREADS), lds, s, a);
    else if (ppt == 8) hipLaunchKernelGGL(fps_kernel<8>, dim3(total), dim3(FPS_THREADS), lds, s, a);
    else hipLaunchKernelGGL(fps_kernel<16>, dim3(total), dim3(FPS_THREADS), lds, s, a);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_fps(bx_ctx* c, hipStream_t s, const float* const* xyz, const int* n, int nclouds, int m, int32_t* const* idx_out,
            float* const* kpts_out)
{
    return bxk_fps_range(c, s, xyz, n, nclouds, 0, m, m, idx_out, kpts_out);
}

int bxk_gather_rows(hipStream_t s, const float* pts, const int32_t* idx, int n, float* out)
{
    if (n <= 0) return BX_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, idx, n, out);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
