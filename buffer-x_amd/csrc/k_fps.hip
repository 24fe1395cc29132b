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
//   * XCD co-location (clouds of up to 32 workgroups): the launch is 8 x G blocks and only the blocks with blockIdx % 8 == x do
//     work, so that the G workgroups of a cloud land on ONE XCD (observed dispatch order: block b -> XCD b % 8; nothing relies on
//     it).  Every workgroup reads its XCC id and the workgroups of a cloud exchange the ids through the agent-scope granules; only
//     if all G ids agree do they switch to the L2 protocol -- plain write-through stores, polls = buffer_inv sc1 + plain load --
//     which never leaves the XCD's L2 (measured all-to-all round, G = 3: 0.57 us against 1.12 us across XCDs).  Any other
//     placement keeps the agent-scope protocol: correctness never depends on where the blocks run.
//   * bucket pruning (round 5; exact).  The per-iteration scan is bound by VALU issue (4 waves per SIMD run ~12 instructions per point),
//     not by one wave's latency, so work that is skipped is time saved.  A pre-pass orders the cloud along a Morton curve of 16^3 cells
//     (count / scan / scatter: bxk_fps_order) and a WAVE owns 64 x PPT points that are consecutive in that order -- a spatially compact
//     BUCKET whose bounding box it holds in SGPRs.  A new sample c can lower the running min-distance of a bucket's point only if
//     d2(c, box) < the bucket's largest min-distance: the box distance is built from the same rounded operations as the point
//     distances, ((dx*dx + dy*dy) + dz*dz) with |dx| of the box <= |dx| of every point inside, and fp32 rounding is monotone, so
//     d2(c, box) <= the COMPUTED d2(c, p) of every point of the bucket -- no margin is needed and the result is the un-pruned one bit
//     for bit (all FPS parity tests run through this path; BX_FPS_PRUNE=0 scans every bucket every iteration).  A skipped bucket
//     re-publishes its cached winner.  Buckets are dealt round-robin to the workgroups and waves of a cloud (bucket b -> workgroup
//     b % G, wave b / G), so that the few buckets near a new sample sit on different SIMDs.
#include "bx_common.h"
// The bucket pruning is exact only because the box distance and the point distances are the SAME unfused operation sequences (monotone
// rounding): no implicit FMA contraction in this file, whatever flags a variant build passes (csrc/Makefile sets -ffp-contract=off too).
#pragma clang fp contract(off)

namespace {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / 64;
constexpr int FPS_MAX_G = 64;            // 64 workgroups x 16 384 points = 1 048 576 points per cloud (the neighbour bitmap's limit too)
constexpr int FPS_REC = 64;              // 8-byte words per exchange record: the bound's {epoch, value} granule, then K x six {key hi, key lo, x, y, z, second key hi}
constexpr int FPS_K = 8;                 // most candidates a workgroup publishes per round (1 + 6 K <= FPS_REC; fewer when G x K would exceed the resolving wave's 64 lanes)
constexpr int FPS_TMAX = 16;             // samples one round may resolve
constexpr int FPS_NE = 2 * FPS_WAVES;    // entries of a workgroup: the two largest keys of each of its 16 buckets
constexpr long long FPS_IDK = (long long)0x8000000000000000LL;    // identity of the key maximum
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
    unsigned long long* slots;  // [cloud][parity 2][FPS_MAX_G][FPS_REC] granules
    int colocate;               // 1: grid = 8 x max G, the workgroups of cloud c are the blocks with blockIdx % 8 == xcd[c]
    int xcd[FPS_MAX_CLOUDS];
    unsigned long long* hello;  // [cloud][FPS_MAX_G] placement handshake granules, zeroed in front of every launch
    long long* dbg;             // BX_FPS_TRACE: cycle stamps of rounds 100..107 of workgroup 0 ([8][8])
    int32_t* err_flag;
    const int32_t* ord[FPS_MAX_CLOUDS];   // [n] point index at every position of the Morton-cell order (bxk_fps_order)
    int prune;                  // 1: skip the buckets a new sample cannot change
    int kmax;                   // candidates a workgroup publishes per round (<= FPS_K)
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
// two independent maxima in one pass: the DPP steps of one chain fill the dependent-issue bubbles of the other
__device__ __forceinline__ void wave_max2_i32(int a, int b, int& ma, int& mb)
{
    const int id = (int)0x80000000;
    int o, p;
#define BX_STEP2(ctl, rm) \
    o = __builtin_amdgcn_update_dpp(id, a, ctl, rm, 0xf, false); p = __builtin_amdgcn_update_dpp(id, b, ctl, rm, 0xf, false); \
    a = o > a ? o : a; b = p > b ? p : b;
    BX_STEP2(0x111, 0xf) BX_STEP2(0x112, 0xf) BX_STEP2(0x114, 0xf) BX_STEP2(0x118, 0xf) BX_STEP2(0x142, 0xa) BX_STEP2(0x143, 0xc)
#undef BX_STEP2
    ma = __builtin_amdgcn_readlane(a, 63);
    mb = __builtin_amdgcn_readlane(b, 63);
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
    // one lane holds the largest distance almost always: its low word is the answer; ties go through the second reduction
    const unsigned long long bal = __ballot(hi == mh);
    unsigned ml;
    if ((bal & (bal - 1ULL)) == 0ULL) ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll((long long)bal) - 1);
    else ml = wave_max_u32(hi == mh ? lo : 0u);
    return (long long)(((unsigned long long)(unsigned)mh << 32) | ml);
}

// the same over lanes 0..15 only (the 16 wave records of a workgroup, the records of <= 16 workgroups): four row_shr steps, result in
// lane 15 -- the two row_bcast steps of the full reduction are two dependent DPP operations on the one wave that is on the critical path
__device__ __forceinline__ int row_max_i32(int v)
{
    const int id = (int)0x80000000;
    int o;
    o = __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
    return __builtin_amdgcn_readlane(v, 15);
}
__device__ __forceinline__ unsigned row_max_u32(unsigned v)
{
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
// lanes >= 16 must hold the identity (0x8000...0)
__device__ __forceinline__ long long row_max_key(long long key)
{
    const int hi = (int)(key >> 32);
    const unsigned lo = (unsigned)((unsigned long long)key & 0xffffffffu);
    const int mh = row_max_i32(hi);
    const unsigned long long bal = __ballot(hi == mh) & 0xffffULL;
    unsigned ml;
    if ((bal & (bal - 1ULL)) == 0ULL) ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll((long long)bal) - 1);
    else ml = row_max_u32(hi == mh ? lo : 0u);
    return (long long)(((unsigned long long)(unsigned)mh << 32) | ml);
}

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}
// granule store / load of the two protocols (fast = all workgroups of the cloud share one XCD's L2)
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v, bool fast)
{
    if (fast) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(unsigned long long* p, bool fast)
{
    unsigned long long x;
    if (fast) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    else x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return x;
}

template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(FpsArgs a)
{
    __shared__ long long s_ekey[2][FPS_NE];        // [parity][{0: largest, 1: second largest key of the bucket}][wave]
    __shared__ int s_eslot[2][FPS_WAVES];          // local slot (i * 1024 + t) of every bucket's largest: its coordinates sit in s_pts (PPT <= 8)
    __shared__ float s_exyz[2][FPS_WAVES][3];      // or its coordinates (PPT 16)
    __shared__ float4 s_pick[2][FPS_TMAX];         // the samples a round resolved (x, y, z), applied by every wave in the next round
    __shared__ long long s_pkey[2][FPS_TMAX];      // and their keys (the point index is the low word)
    __shared__ int s_npick[2];

    int cloud = 0, g;
    if (a.colocate) {
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        cloud = -1;
        for (int cl = 0; cl < a.nclouds; ++cl)
            if (x == a.xcd[cl] && slot < a.G[cl]) cloud = cl;
        if (cloud < 0) return;
        g = slot;
    } else {
        if (a.nclouds > 1 && (int)blockIdx.x >= a.wg_start[1]) cloud = 1;
        g = blockIdx.x - a.wg_start[cloud];
    }
    const int G = a.G[cloud];
    const int n = a.n[cloud];
    const float* __restrict__ xyz = a.xyz[cloud];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int32_t* __restrict__ ord = a.ord[cloud];
    // bucket of this wave: 64 x PPT consecutive positions of the spatial order; buckets dealt round-robin over (workgroup, wave)
    const int sp0 = (wave * G + g) * (64 * PPT) + lane;

    int T = 1, lt = 0;
    while (T * 2 <= n && T * 2 <= 512) { T *= 2; ++lt; }
    const unsigned tmask = (unsigned)(T - 1);

    // PPT <= 8: the workgroup's coordinates also sit in LDS ([3][PPT * 1024] floats), so that the scan only tracks {distance, slot}
    // and the coordinates of a winner are ONE LDS read by the lane that needs them (3 selects per point less in the VALU-bound scan)
    constexpr bool LDSXYZ = PPT <= 8;
    constexpr int NP = PPT * FPS_THREADS;
    extern __shared__ float s_pts[];
    // running min-distances: registers (PPT <= 8) or, for PPT 16 -- whose coordinates alone take 48 of the 128 VGPRs a 1024-thread
    // workgroup leaves a lane --, the dynamic LDS ([16][1024] floats, conflict-free: consecutive lanes, consecutive words)
    constexpr bool LDSTD = !LDSXYZ;
    float px[PPT], py[PPT], pz[PPT], td_r[LDSTD ? 1 : PPT];
#define TD(i) (*(LDSTD ? &s_pts[(i) * FPS_THREADS + t] : &td_r[LDSTD ? 0 : (i)]))
    // low word of the point's key = ~tie-break(k): the tie rule, and k itself (invertible).  PPT 16 has no register for it (the kernel
    // sits at its 128-VGPR ceiling): the word is re-derived from the spatial order where a bucket's entries are computed / its state saved
    constexpr int NPT = LDSXYZ ? PPT : 1;
    unsigned pt[NPT];
    auto tiebreak = [&](int i) -> unsigned {
        const int sp = sp0 + i * 64;
        if (sp >= n) return 0u;
        const int k = ord ? ord[sp] : sp;
        return ~((((unsigned)k & tmask) << 23) | ((unsigned)k >> lt));
    };
    float lox = 3.0e38f, loy = 3.0e38f, loz = 3.0e38f, hix = -3.0e38f, hiy = -3.0e38f, hiz = -3.0e38f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int sp = sp0 + i * 64;
        if (sp < n) {
            const int k = ord ? ord[sp] : sp;
            if (LDSXYZ) pt[i < NPT ? i : 0] = ~((((unsigned)k & tmask) << 23) | ((unsigned)k >> lt));
            px[i] = xyz[(size_t)k * 3 + 0];
            py[i] = xyz[(size_t)k * 3 + 1];
            pz[i] = xyz[(size_t)k * 3 + 2];
            float mag = (px[i] * px[i] + py[i] * py[i]) + pz[i] * pz[i];
            TD(i) = (mag <= 1e-3f) ? -1.0f : 1e10f;  // -1 marks "never a candidate" (upstream `continue`)
            if (a.j0 > 0) TD(i) = a.td_state[cloud][k];
            lox = fminf(lox, px[i]); loy = fminf(loy, py[i]); loz = fminf(loz, pz[i]);
            hix = fmaxf(hix, px[i]); hiy = fmaxf(hiy, py[i]); hiz = fmaxf(hiz, pz[i]);
        } else {
            if (LDSXYZ) pt[i < NPT ? i : 0] = 0u;
            px[i] = 0.f; py[i] = 0.f; pz[i] = 0.f; TD(i) = -1.0f;
        }
        if (LDSXYZ) {
            s_pts[i * FPS_THREADS + t] = px[i];
            s_pts[NP + i * FPS_THREADS + t] = py[i];
            s_pts[2 * NP + i * FPS_THREADS + t] = pz[i];
        }
    }
    // the bucket's bounding box (wave-uniform: SGPRs); an empty bucket keeps lo > hi and is never scanned (all its td are -1)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lox = fminf(lox, __shfl_xor(lox, o)); loy = fminf(loy, __shfl_xor(loy, o)); loz = fminf(loz, __shfl_xor(loz, o));
        hix = fmaxf(hix, __shfl_xor(hix, o)); hiy = fmaxf(hiy, __shfl_xor(hiy, o)); hiz = fmaxf(hiz, __shfl_xor(hiz, o));
    }
    lox = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lox))); loy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(loy)));
    loz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(loz))); hix = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hix)));
    hiy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hiy))); hiz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hiz)));
    float wmax = 0.0f;                             // the bucket's largest running min-distance, known from the first scan on
    bool scanned = false;                          // (an explicit flag, not a 3e38 sentinel: a box distance that overflows to inf / NaN on
                                                   //  far-out coordinates must not keep a bucket from ever being scanned)
    // cached entries of the bucket: its two largest keys {fp32 bits of the running min-distance, tie-break word} + where their points are
    long long e1k = FPS_IDK, e2k = FPS_IDK;
    int e1s = 0;
    float e1x = 0.f, e1y = 0.f, e1z = 0.f, e2x = 0.f, e2y = 0.f, e2z = 0.f;
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    if (a.j0 > 0) {   // the previous launch of this stream wrote the last keypoint (kernel boundary: visible)
        const float* lk = a.kpts_out[cloud] + (size_t)(a.j0 - 1) * 3;
        cx = lk[0]; cy = lk[1]; cz = lk[2];
    } else if (g == 0 && t == 0) {
        a.idx_out[cloud][0] = 0;
        if (a.kpts_out[cloud]) { a.kpts_out[cloud][0] = cx; a.kpts_out[cloud][1] = cy; a.kpts_out[cloud][2] = cz; }
    }
    unsigned long long* slots = a.slots + (size_t)cloud * 2 * FPS_MAX_G * FPS_REC;
    // placement handshake: do all G workgroups of this cloud sit on one XCD?
    bool fast = false;
    if (a.colocate && G > 1) {
        __shared__ int s_fast;
        if (wave == 0) {
            unsigned long long* hello = a.hello + (size_t)cloud * FPS_MAX_G;
            const unsigned me = xcc_id();
            if (lane == 0) __hip_atomic_store(hello + g, 0x100000000ULL | me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool same = true, fail = false;
            unsigned spins = 0;
            while (true) {
                bool ok = true;
                if (lane < G) {
                    const unsigned long long x = __hip_atomic_load(hello + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (x >> 32) != 0;
                    same = (unsigned)x == me;
                }
                if (__all(ok)) break;
                if (++spins > FPS_SPIN_LIMIT) { fail = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (fail && lane == 0) atomicOr(a.err_flag, 1);
            const bool all_same = __all(same) && !fail;
            if (lane == 0) s_fast = all_same ? 1 : 0;
        }
        __syncthreads();
        fast = s_fast != 0;
    }
    if (a.dbg && blockIdx.x == 0 && t == 0) a.dbg[63] = (fast ? 1 : 0) + 2 * a.colocate + 100 * G;

    // ---------------------------------------------------------------- the sampling loop, in ROUNDS (round 6)
    // Rounds 1-5 resolved ONE sample per cross-workgroup exchange: K strictly dependent iterations of (scan -> workgroup reduce ->
    // publish -> L2 round trip -> reduce over G -> decode) = 1.95 us each, none of it shortened by skipped work (profiles/r05_fps.txt).
    // A round now resolves SEVERAL samples from one exchange, exactly:
    //   * every bucket (wave) keeps its TWO largest keys; a workgroup publishes its K largest bucket MAXIMA (the candidates) with their
    //     coordinates and, per candidate, the bucket's second key -- every other point of that bucket is below it --, plus a BOUND for the
    //     buckets it does not list: its (K + 1)-th largest bucket maximum;
    //   * every workgroup then runs the same deterministic resolution on the same G x K candidates (one per lane of wave 0): take the
    //     largest key -> that is the next sample (the global arg-max: keys are unique, every workgroup's maximum is listed); apply it to
    //     the CANDIDATES (td = min(td, d), the very operations of the scan); raise the bound B to the second key of the sampled
    //     candidate's bucket (its other points are no longer covered by a listed maximum); the next largest candidate key is the next
    //     sample as long as it is >= B -- running min-distances only fall, and the hidden points of a bucket whose maximum is still an
    //     untouched candidate are below that candidate; a candidate whose key a sample LOWERED hands its bucket's second key to B as
    //     well --; stop at the first candidate below B or after FPS_TMAX samples, and exchange again;
    //   * the next round starts by applying the resolved samples to the buckets they can change (box test per sample, one lane each).
    // The sample sequence is the sequential one bit for bit (every FPS test, tilings and tie lattices included); what changes is the
    // number of exchanges: ~1 per 4-6 samples.
    const int K = a.kmax < 64 / G ? a.kmax : 64 / G;                            // G K <= 64 lanes of the resolving wave
    const int cw = lane / K, ce = lane - cw * K;                                 // candidate (workgroup, entry) of this lane in the resolving wave
    if (t == 0) { s_pick[1][0] = make_float4(cx, cy, cz, 0.f); s_npick[1] = 1; }   // "previous round": the first sample
    __syncthreads();
    int j = a.j0 > 0 ? a.j0 : 1;                     // next output index
    int lastpar = 1;
    int npv = 1;                                     // samples of the previous round (the "round" in front of the first: the first sample)
    int outn = 0, outj = 0;                          // samples of the previous round that still have to be written, their first output index
    // the samples of a round are written to global memory by wave 1 of workgroup 0 DURING the next round's resolution (the wave would be parked at
    // the barrier): ~400 cycles per round off the path every workgroup waits on.  s_pkey / s_pick of a parity are stable until the round after next.
    auto emit = [&](int pp, int cnt, int jb) {
        if (g == 0 && wave == 1 && lane < cnt) {
            const long long fk = s_pkey[pp][lane];
            int old = 0;
            if (fk >= 0) {
                const unsigned tbw = ~(unsigned)((unsigned long long)fk & 0xffffffffu);
                old = (int)(((tbw & 0x7fffffu) << lt) | (tbw >> 23));
            }
            const float4 q = s_pick[pp][lane];
            a.idx_out[cloud][jb + lane] = old;
            if (a.kpts_out[cloud]) {
                a.kpts_out[cloud][(size_t)(jb + lane) * 3 + 0] = q.x;
                a.kpts_out[cloud][(size_t)(jb + lane) * 3 + 1] = q.y;
                a.kpts_out[cloud][(size_t)(jb + lane) * 3 + 2] = q.z;
            }
        }
    };
    for (unsigned r = 0; j < a.m; ++r) {
        const int par = (int)(r & 1u), prv = par ^ 1;
        lastpar = par;
        const unsigned ep = r + 1u;                  // epoch of this round's granules (the slots are zeroed in front of every launch)
        const bool tr = a.dbg != nullptr && blockIdx.x == 0 && t == 0 && r >= 100u && r < 108u;
        long long* tdp = a.dbg + (r - 100u) * 8;
#define FPS_TR(q) do { if (tr) tdp[q] = __builtin_readcyclecounter(); } while (0)
        FPS_TR(0);
        // ---- A: the samples of the previous round against this bucket.  Box test of every sample at once (lane q = sample q):
        //      d2(sample, box) from the same rounded operations as the point distances (monotone: <= the computed distance of every
        //      point inside the box), so a sample with d2 >= the bucket's largest running min-distance cannot change the bucket
        const int np = npv;
        unsigned long long hit;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);      // lane q: sample q of the previous round (the update below takes it from here)
        {
            bool h = false;
            if (lane < np) {
                h = true;
                qv = s_pick[prv][lane];
                if (a.prune && scanned) {
                    const float4 q = qv;
                    const float ex = fmaxf(fmaxf(lox - q.x, q.x - hix), 0.0f), ey = fmaxf(fmaxf(loy - q.y, q.y - hiy), 0.0f), ez = fmaxf(fmaxf(loz - q.z, q.z - hiz), 0.0f);
                    const float db = (ex * ex + ey * ey) + ez * ez;
                    h = db < wmax;
                }
            }
            hit = __ballot(h);
        }
        // a sample changes the bucket's two cached entries only if it lowers THEIR running min-distances (d < td, the comparison the
        // update itself makes): every other point can only fall, so untouched entries stay the two largest and the selection is skipped
        bool upd = false, sel = !scanned;
        while (hit != 0ULL) {
            const int q = __ffsll((long long)hit) - 1;
            hit &= hit - 1ULL;
            float4 c4;                                    // (a lane broadcast instead of a second LDS read with its latency in the chain)
            c4.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv.x), q));
            c4.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv.y), q));
            c4.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv.z), q));
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                float dx = px[i] - c4.x, dy = py[i] - c4.y, dz = pz[i] - c4.z;
                float d = (dx * dx + dy * dy) + dz * dz;
                TD(i) = fminf(d, TD(i));
            }
            {
                float dx = e1x - c4.x, dy = e1y - c4.y, dz = e1z - c4.z;
                const float d1 = (dx * dx + dy * dy) + dz * dz;
                dx = e2x - c4.x; dy = e2y - c4.y; dz = e2z - c4.z;
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                sel = sel || d1 < __int_as_float((int)(e1k >> 32)) || d2 < __int_as_float((int)(e2k >> 32));
            }
            upd = true;
        }
        if (upd && sel) {
            // the bucket's two largest keys.  key = {fp32 bits of the running min-distance, ~tie-break(k)}: unique per point, and its
            // order IS the upstream rule (larger distance, then lower k mod T, then lower k)
            long long k1 = FPS_IDK, k2 = FPS_IDK;
            int i1 = 0, i2 = 0;
            auto offer = [&](int i, unsigned pti, float tdi) {
                const long long ki = (long long)(((unsigned long long)(unsigned)__float_as_int(tdi) << 32) | (unsigned long long)pti);
                const bool b1 = ki > k1, b2 = ki > k2;
                k2 = b1 ? k1 : (b2 ? ki : k2);
                i2 = b1 ? i1 : (b2 ? i : i2);
                k1 = b1 ? ki : k1;
                i1 = b1 ? i : i1;
            };
            if constexpr (LDSTD) {
                // PPT 16: distances from LDS, tie-break words from the spatial order -- no register array involved, so the loop stays
                // rolled (unrolled, its sixteen loads in flight cost more registers than the kernel has)
#pragma unroll 1
                for (int i = 0; i < PPT; ++i) offer(i, tiebreak(i), TD(i));
            } else {
#pragma unroll
                for (int i = 0; i < PPT; ++i) offer(i, pt[i < NPT ? i : 0], TD(i));
            }
            e1k = wave_max_key(k1);
            const int wl1 = __ffsll((long long)__ballot(k1 == e1k)) - 1;
            const long long kk = lane == wl1 ? k2 : k1;          // the winner's lane offers its second point
            const int ii = lane == wl1 ? i2 : i1;
            e2k = wave_max_key(kk);
            const int wl2 = __ffsll((long long)__ballot(kk == e2k)) - 1;
            if (LDSXYZ) {
                e1s = __builtin_amdgcn_readlane(i1 * FPS_THREADS + t, wl1);
                const int e2s = __builtin_amdgcn_readlane(ii * FPS_THREADS + t, wl2);
                e1x = s_pts[e1s]; e1y = s_pts[NP + e1s]; e1z = s_pts[2 * NP + e1s];
                e2x = s_pts[e2s]; e2y = s_pts[NP + e2s]; e2z = s_pts[2 * NP + e2s];
            } else {
                // PPT 16 keeps no LDS copy of the coordinates and no register to spare for select chains over px / py / pz: the two
                // points are re-read from the cloud by their index (the key's low word; wave-uniform addresses, L2-resident) -- two
                // loads in flight on the scanning wave only, on clouds beyond 131 072 points only
                const unsigned tb1 = ~(unsigned)((unsigned long long)e1k & 0xffffffffu), tb2 = ~(unsigned)((unsigned long long)e2k & 0xffffffffu);
                const int p1 = (int)(((tb1 & 0x7fffffu) << lt) | (tb1 >> 23)), p2 = (int)(((tb2 & 0x7fffffu) << lt) | (tb2 >> 23));
                const bool v1 = p1 >= 0 && p1 < n, v2 = p2 >= 0 && p2 < n;       // (identity / padding keys decode to nothing: never picked)
                const float* q1 = xyz + (size_t)(v1 ? p1 : 0) * 3;
                const float* q2 = xyz + (size_t)(v2 ? p2 : 0) * 3;
                e1x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q1[0]))); e1y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q1[1])));
                e1z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q1[2]))); e2x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q2[0])));
                e2y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q2[1]))); e2z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q2[2])));
                (void)ii; (void)wl1; (void)wl2;
            }
            scanned = true;
            wmax = __int_as_float((int)(e1k >> 32));           // the bucket's largest min-distance (-1: no candidate left)
        }
        if (lane == 0) {
            s_ekey[par][wave] = e1k; s_ekey[par][FPS_WAVES + wave] = e2k;
            if (LDSXYZ) s_eslot[par][wave] = e1s;
            else { s_exyz[par][wave][0] = e1x; s_exyz[par][wave][1] = e1y; s_exyz[par][wave][2] = e1z; }
        }
        FPS_TR(1);
        __syncthreads();
        FPS_TR(2);
        if (G > 1) {
            // ---- publish, DISTRIBUTED over the sixteen waves (the first form ranked the bucket maxima on wave 0 alone: ~180 dependent
            //      instructions = 1 800 cycles per round on the one wave every other wave waits for).  A wave knows its own bucket maximum;
            //      its rank among the sixteen = the maxima that beat it (equal keys -- identities, padding -- ordered by wave index): one LDS
            //      read, one compare, one ballot.  Rank < K: the wave stores its candidate's six granules itself; rank K: its distance is the
            //      bound of everything the workgroup does not list.
            const long long other = lane < FPS_WAVES ? s_ekey[par][lane] : FPS_IDK;
            const bool beats = lane < FPS_WAVES && (other > e1k || (other == e1k && lane < wave));
            const int rank = __popcll(__ballot(beats));
            unsigned long long* my = slots + ((size_t)par * FPS_MAX_G + g) * FPS_REC;
            const unsigned long long eh = (unsigned long long)ep << 32;
            if (rank < K) {
                // (bounds travel as HIGH WORDS only -- the fp32 distances: the resolution continues only while a candidate's distance is
                //  strictly ABOVE the bound's, a tie in the distance ends the round -- conservative, hence exact, and half the words)
                if (lane < 6) {
                    const unsigned v = lane == 0 ? (unsigned)((unsigned long long)e1k >> 32)
                                     : lane == 1 ? (unsigned)((unsigned long long)e1k & 0xffffffffu)
                                     : lane == 2 ? __float_as_uint(e1x) : lane == 3 ? __float_as_uint(e1y) : lane == 4 ? __float_as_uint(e1z)
                                     : (unsigned)((unsigned long long)e2k >> 32);
                    granule_store(my + 1 + rank * 6 + lane, eh | v, fast);
                }
            } else if (rank == K) {
                if (lane == 0) granule_store(my, eh | (unsigned)((unsigned long long)e1k >> 32), fast);
            }
        }
        emit(prv, outn, outj);
        if (wave == 0) {
            long long ck = FPS_IDK;                        // candidate key (resolution input)
            int bh = (int)0x80000000, ce2h = (int)0x80000000;   // high words: bound of the candidate's workgroup / second key of its bucket
            float cxv = 0.f, cyv = 0.f, czv = 0.f;
            bool act = lane < FPS_WAVES;
            bool fail = false;
            if (G == 1) {
                // one workgroup: all sixteen bucket maxima are candidates (straight from LDS), nothing is unlisted
                if (act) {
                    ck = s_ekey[par][lane];
                    ce2h = (int)(s_ekey[par][FPS_WAVES + lane] >> 32);
                    if (LDSXYZ) { const int sl = s_eslot[par][lane]; cxv = s_pts[sl]; cyv = s_pts[NP + sl]; czv = s_pts[2 * NP + sl]; }
                    else { cxv = s_exyz[par][lane][0]; cyv = s_exyz[par][lane][1]; czv = s_exyz[par][lane][2]; }
                }
            } else {
                FPS_TR(3);
                // ---- poll: lane (workgroup cw, entry ce) reads its candidate's six granules + the one of the workgroup's bound
                act = lane < G * K;
                unsigned long long r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, b0 = 0;
                {
                    unsigned long long* rp = slots + ((size_t)par * FPS_MAX_G + (act ? cw : 0)) * FPS_REC + 1 + (act ? ce : 0) * 6;
                    unsigned long long* bp = slots + ((size_t)par * FPS_MAX_G + (act ? cw : 0)) * FPS_REC;
                    unsigned spins = 0;
                    while (true) {
                        bool ok = true;
                        if (act) {
                            if (fast) {
                                asm volatile("buffer_inv sc1\n\t"
                                             "global_load_dwordx2 %0, %7, off\n\t"
                                             "global_load_dwordx2 %1, %7, off offset:8\n\t"
                                             "global_load_dwordx2 %2, %7, off offset:16\n\t"
                                             "global_load_dwordx2 %3, %7, off offset:24\n\t"
                                             "global_load_dwordx2 %4, %7, off offset:32\n\t"
                                             "global_load_dwordx2 %5, %7, off offset:40\n\t"
                                             "global_load_dwordx2 %6, %8, off\n\t"
                                             "s_waitcnt vmcnt(0)"
                                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(b0) : "v"(rp), "v"(bp) : "memory");
                            } else {
                                r0 = __hip_atomic_load(rp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                r1 = __hip_atomic_load(rp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                r2 = __hip_atomic_load(rp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                r3 = __hip_atomic_load(rp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                r4 = __hip_atomic_load(rp + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                r5 = __hip_atomic_load(rp + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                b0 = __hip_atomic_load(bp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            ok = (unsigned)(r0 >> 32) == ep && (unsigned)(r1 >> 32) == ep && (unsigned)(r2 >> 32) == ep &&
                                 (unsigned)(r3 >> 32) == ep && (unsigned)(r4 >> 32) == ep && (unsigned)(r5 >> 32) == ep && (unsigned)(b0 >> 32) == ep;
                        }
                        if (__all(ok)) break;
                        if (++spins > FPS_SPIN_LIMIT) { fail = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (fail && lane == 0) atomicOr(a.err_flag, 1);
                ck = act ? (long long)(((unsigned long long)(unsigned)r0 << 32) | (unsigned)r1) : FPS_IDK;
                bh = act ? (int)(unsigned)b0 : (int)0x80000000;
                ce2h = act ? (int)(unsigned)r5 : (int)0x80000000;
                cxv = __uint_as_float((unsigned)r2); cyv = __uint_as_float((unsigned)r3); czv = __uint_as_float((unsigned)r4);
            }
            FPS_TR(4);
            // ---- resolution: the same inputs and the same operations in every workgroup.  This loop is the serial instruction stream of ONE
            //      wave with fifteen parked at the barrier, ~10 cycles per dependent instruction: it is written for instruction count.  The
            //      keys stay split in their words; the second keys of the candidates a sample lowered wait in a per-lane maximum (`pend`)
            //      whose reduction rides in the bubbles of the NEXT iteration's key reduction (two independent DPP chains) and reaches the
            //      bound before that iteration's test -- exactly when the one-reduction-more of the first form applied it; a sample is kept
            //      in lane t of five registers (one compare, five selects) and stored once after the loop instead of behind an exec-mask branch per sample.
            int Bh = (int)0x80000000;      // bound (fp32 distance bits) of everything that is not a candidate: the workgroups' bounds reach it
                                           // through `pend` in the first iteration, whose sample (the global maximum) needs no bound
            int tlim = a.m - j;
            tlim = tlim < FPS_TMAX ? tlim : FPS_TMAX;
            int tc = 0;
            int khi = (int)(ck >> 32);
            const unsigned klo = (unsigned)((unsigned long long)ck & 0xffffffffu);
            int pend = bh;
            int pkh = 0, pkl = 0, pkx = 0, pky = 0, pkz = 0;           // lane t: the round's sample t (key words, coordinates)
            // (tlim >= 1: the first iteration can only leave through "no candidate", every later one through the bound or the count)
            for (;;) {
                int mh, pb;
                wave_max2_i32(khi, pend, mh, pb);
                Bh = __builtin_amdgcn_readfirstlane(pb > Bh ? pb : Bh);
                // no candidate anywhere (every point within 1e-3 of the origin, or sampled: upstream yields index 0 -- once per round), or a
                // point outside the lists may have a larger (or the same) distance: exchange again
                if (mh < 0 || (tc > 0 && mh <= Bh)) break;
                // one lane holds the largest distance almost always: its low word is the answer; ties go through a second reduction
                const unsigned long long bal = __ballot(khi == mh);
                int wl = __builtin_ctzll(bal);
                unsigned ml = (unsigned)__builtin_amdgcn_readlane((int)klo, wl);
                if (__builtin_expect(__popcll(bal) != 1, 0)) {
                    ml = wave_max_u32(khi == mh ? klo : 0u);
                    wl = __builtin_ctzll(__ballot(khi == mh && klo == ml));
                }
                const int qxi = __builtin_amdgcn_readlane(__float_as_int(cxv), wl);
                const int qyi = __builtin_amdgcn_readlane(__float_as_int(cyv), wl);
                const int qzi = __builtin_amdgcn_readlane(__float_as_int(czv), wl);
                const int e2w = __builtin_amdgcn_readlane(ce2h, wl);
                // lane tc of the five sample registers, in place (v_writelane ignores the exec mask; the lane select travels in m0, which
                // does not count against the one-SGPR limit of the encoding)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 is reserved but unused by this kernel (tests/test_isa_lint.py checks)
                asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                             "v_writelane_b32 %0, %6, m0\n\tv_writelane_b32 %1, %7, m0\n\tv_writelane_b32 %2, %8, m0\n\t"
                             "v_writelane_b32 %3, %9, m0\n\tv_writelane_b32 %4, %10, m0"
                             : "+v"(pkh), "+v"(pkl), "+v"(pkx), "+v"(pky), "+v"(pkz)
                             : "s"(tc), "s"(mh), "s"((int)ml), "s"(qxi), "s"(qyi), "s"(qzi) : "m0");
#pragma clang diagnostic pop
                ++tc;
                // the sampled candidate's bucket: its other points are bounded by the bucket's second key from now on
                Bh = __builtin_amdgcn_readfirstlane(e2w > Bh ? e2w : Bh);
                // the sample against the candidates: the operations of the bucket update above
                {
                    const float qx = __int_as_float(qxi), qy = __int_as_float(qyi), qz = __int_as_float(qzi);
                    float dx = cxv - qx, dy = cyv - qy, dz = czv - qz;
                    float d = (dx * dx + dy * dy) + dz * dz;
                    // (a lane without a candidate holds the identity key: its high word is -0.0f, which min() keeps)
                    const float nt = fminf(d, __int_as_float(khi));
                    // another candidate lost distance: the hidden points of ITS bucket are no longer covered by a listed maximum either -- the
                    // bucket's second key joins the bound (through `pend`, before the next test)
                    const bool lowered = act && lane != wl && __float_as_int(nt) != khi;
                    khi = __float_as_int(nt);
                    pend = lowered && ce2h > pend ? ce2h : pend;
                }
                if (tc >= tlim) break;
            }
            const bool none = tc == 0;
            if (none) {
                if (lane == 0) { s_pkey[par][0] = -1LL; s_pick[par][0] = make_float4(xyz[0], xyz[1], xyz[2], 0.f); }
                tc = 1;
            } else if (lane < tc) {
                s_pkey[par][lane] = (long long)(((unsigned long long)(unsigned)pkh << 32) | (unsigned)pkl);
                s_pick[par][lane] = make_float4(__int_as_float(pkx), __int_as_float(pky), __int_as_float(pkz), 0.f);
            }
            if (lane == 0) s_npick[par] = tc;
            FPS_TR(5);
        }
        __syncthreads();
        FPS_TR(6);
        const int npk = s_npick[par];
        if (tr) tdp[7] = npk;
        npv = npk; outn = npk; outj = j;
        j += npk;
    }
    emit(lastpar, outn, outj);
#undef FPS_TR
    // the samples of the last round have not been applied to the buckets yet: a following launch of a tiled run picks the running
    // min-distances up from td_state and applies only the LAST keypoint itself (min is idempotent: applying that one twice is harmless)
    if (a.save) {
        const int np = s_npick[lastpar];
        for (int q = 0; q < np; ++q) {
            const float4 c4 = s_pick[lastpar][q];
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                float dx = px[i] - c4.x, dy = py[i] - c4.y, dz = pz[i] - c4.z;
                float d = (dx * dx + dy * dy) + dz * dz;
                TD(i) = fminf(d, TD(i));
            }
        }
    }
    if (a.save) {
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (sp0 + i * 64 < n) {
                const unsigned tbw = ~(LDSXYZ ? pt[i < NPT ? i : 0] : tiebreak(i));
                a.td_state[cloud][(int)(((tbw & 0x7fffffu) << lt) | (tbw >> 23))] = TD(i);
            }
    }
}

#undef TD

// ---- spatial order of a cloud (bucket pruning): Morton curve over 16 x 16 x 16 cells of the bounding box, counting sort.  The order
//      inside a cell is whatever the atomics produce -- it decides which bucket a point sits in, never the sampling result.
struct OrderArgs {
    const float* xyz[FPS_MAX_CLOUDS];
    int n[FPS_MAX_CLOUDS];
    unsigned* bbmin[FPS_MAX_CLOUDS];    // [3] encoded min x, y, z (memset 0xff)
    unsigned* bbmax[FPS_MAX_CLOUDS];    // [3] encoded max x, y, z (memset 0)
    int* cnt[FPS_MAX_CLOUDS];           // [4096] points per cell -> exclusive start -> fill cursor
    unsigned short* cell;               // [2][max_points] Morton cell of every point
    int32_t* ord[FPS_MAX_CLOUDS];       // [n]
    int stride;                         // max_points
};
constexpr int FPS_CELLS = 4096;
__device__ __forceinline__ unsigned fenc(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }   // monotone
__device__ __forceinline__ float fdec(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ __forceinline__ unsigned spread4(unsigned v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }   // bit i -> bit 3 i

__global__ __launch_bounds__(256) void fps_bbox_kernel(OrderArgs a)
{
    const int c = blockIdx.y, n = a.n[c];
    const float* __restrict__ p = a.xyz[c];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256)
#pragma unroll
        for (int d = 0; d < 3; ++d) { const float v = p[(size_t)k * 3 + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(a.bbmin[c] + d, fenc(lo[d])); atomicMax(a.bbmax[c] + d, fenc(hi[d])); }
    }
}
__global__ __launch_bounds__(256) void fps_cell_kernel(OrderArgs a)
{
    const int c = blockIdx.y, n = a.n[c];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    unsigned q[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float lo = fdec(a.bbmin[c][d]), hi = fdec(a.bbmax[c][d]);
        const float w = hi - lo;
        const float s = w > 0.f ? 16.0f / w : 0.f;
        int v = (int)((a.xyz[c][(size_t)k * 3 + d] - lo) * s);
        q[d] = (unsigned)(v < 0 ? 0 : (v > 15 ? 15 : v));
    }
    const unsigned cell = spread4(q[0]) | (spread4(q[1]) << 1) | (spread4(q[2]) << 2);
    a.cell[(size_t)c * a.stride + k] = (unsigned short)cell;
    atomicAdd(a.cnt[c] + cell, 1);
}
__global__ __launch_bounds__(1024) void fps_cell_scan_kernel(OrderArgs a)
{
    __shared__ int s_w[16];
    const int c = blockIdx.x, t = threadIdx.x;
    int* cnt = a.cnt[c];
    int v[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = cnt[t * 4 + i]; sum += v[i]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
    if ((t & 63) == 63) s_w[t >> 6] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w = 0; w < (t >> 6); ++w) base += s_w[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) { cnt[t * 4 + i] = base; base += v[i]; }
}
__global__ __launch_bounds__(256) void fps_scatter_kernel(OrderArgs a)
{
    const int c = blockIdx.y, n = a.n[c];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int pos = atomicAdd(a.cnt[c] + a.cell[(size_t)c * a.stride + k], 1);
    a.ord[c][pos] = k;
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
    a.dbg = getenv("BX_FPS_TRACE") ? c->ball_dbg : nullptr;
    int ppt = 4;
    int total = 0;
    int nmax = 0;
    for (int i = 0; i < nclouds; ++i) nmax = n[i] > nmax ? n[i] : nmax;
    // points per thread: the per-iteration scan is VALU-bound (4 waves per SIMD), the all-to-all exchange grows by ~0.1 us per
    // workgroup.  Measured (K = 5000): PPT 4 wins up to 8 workgroups (32k points: 1.8 us per iteration), PPT 8 with its LDS copy of
    // the coordinates up to 16 workgroups (2.0 / 2.1 / 2.55 us at 38k / 55k / 100k points; PPT 16 at 100k: 3.2 us), PPT 16 beyond
    ppt = nmax <= 8 * FPS_THREADS * 4 ? 4 : (nmax <= 16 * FPS_THREADS * 8 ? 8 : 16);
    {
        const char* e = getenv("BX_FPS_PPT");       // test hook: every PPT instantiation on any cloud size
        const int force = e ? atoi(e) : 0;
        if ((force == 4 || force == 8 || force == 16) && (nmax + FPS_THREADS * force - 1) / (FPS_THREADS * force) <= FPS_MAX_G) ppt = force;
    }
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
    // XCD co-location: every cloud's workgroups on one XCD (32 CUs each), the two clouds of a pair on the two XCDs of the context's
    // pair; contexts take the four pairs in creation order.  The workgroups of a cloud wait for each other, so everything aimed at
    // one XCD must be able to be resident there at once whatever the other contexts launch: co-locate only while
    // G x (contexts sharing the XCD) <= 32 workgroups -- otherwise dispatch-order placement over the whole chip, as before.
    const char* ec = getenv("BX_FPS_COLOCATE");     // test hook: 0 = dispatch-order placement (agent-scope protocol)
    const int coloc = ec ? atoi(ec) : 1;
    int gmax = 0;
    for (int i = 0; i < nclouds; ++i) gmax = a.G[i] > gmax ? a.G[i] : gmax;
    if (c->fps_xcd_pair < 0) c->fps_xcd_pair = bx_xcd_slot_take(c->device);
    const int sharing = bx_xcd_pair_sharing(c->device, c->fps_xcd_pair);     // live contexts aimed at the same XCD pair
    a.colocate = (coloc && gmax > 1 && gmax * (sharing < 1 ? 1 : sharing) <= 32) ? 1 : 0;
    a.hello = c->fps_hello;
    if (a.colocate) {
        for (int i = 0; i < nclouds; ++i) a.xcd[i] = (2 * c->fps_xcd_pair + i) & 7;
        BX_HIP(hipMemsetAsync(c->fps_hello, 0, sizeof(unsigned long long) * FPS_MAX_CLOUDS * FPS_MAX_G, s));
        total = 8 * gmax;
    }
    // the epochs of a launch start at 1: the slots are cleared in front of every launch (tiled runs included)
    BX_HIP(hipMemsetAsync(c->fps_slots, 0, sizeof(unsigned long long) * FPS_MAX_CLOUDS * 2 * FPS_MAX_G * FPS_REC, s));
    // the spatial order of the clouds (bucket pruning), once per run
    {
        const char* ep = getenv("BX_FPS_PRUNE");       // test / measurement hook: 0 = scan every bucket in every iteration
        a.prune = ep ? atoi(ep) : 1;
        const char* ek = getenv("BX_FPS_K");          // measurement hook: candidates per workgroup and round (results do not depend on it)
        a.kmax = ek ? atoi(ek) : 8;
        if (a.kmax < 1) a.kmax = 1;
        if (a.kmax > FPS_K) a.kmax = FPS_K;
        // a cloud beyond the context's max_points (stage entry point only: bx_register_pair checks its clouds) has no room for its order:
        // it is sampled in input order -- buckets that are not compact are rarely skipped, the result is the same
        bool ordered = true;
        for (int i = 0; i < nclouds; ++i) ordered = ordered && n[i] <= c->p.max_points;
        for (int i = 0; i < nclouds; ++i) a.ord[i] = ordered ? c->fps_ord + (size_t)i * c->p.max_points : nullptr;
        if (j0 == 0 && ordered) {
            OrderArgs o{};
            for (int i = 0; i < nclouds; ++i) {
                o.xyz[i] = xyz[i]; o.n[i] = n[i];
                o.bbmin[i] = c->fps_bbmin + 4 * i; o.bbmax[i] = c->fps_bbmax + 4 * i; o.cnt[i] = c->fps_cnt + FPS_CELLS * i;
                o.ord[i] = c->fps_ord + (size_t)i * c->p.max_points;
            }
            o.cell = c->fps_cell; o.stride = c->p.max_points;
            BX_HIP(hipMemsetAsync(c->fps_cnt, 0, sizeof(int) * (FPS_CELLS * FPS_MAX_CLOUDS + 8), s));    // cell counters + fps_bbmax (carved behind them)
            BX_HIP(hipMemsetAsync(c->fps_bbmin, 0xff, sizeof(unsigned) * 8, s));
            hipLaunchKernelGGL(fps_bbox_kernel, dim3(32, nclouds), dim3(256), 0, s, o);
            hipLaunchKernelGGL(fps_cell_kernel, dim3((nmax + 255) / 256, nclouds), dim3(256), 0, s, o);
            hipLaunchKernelGGL(fps_cell_scan_kernel, dim3(nclouds), dim3(1024), 0, s, o);
            hipLaunchKernelGGL(fps_scatter_kernel, dim3((nmax + 255) / 256, nclouds), dim3(256), 0, s, o);
        }
    }
    const size_t lds = ppt <= 8 ? (size_t)3 * ppt * FPS_THREADS * sizeof(float) : (size_t)16 * FPS_THREADS * sizeof(float);     // PPT 8: 96 KiB of coordinates; PPT 16: 64 KiB of running min-distances
    if (lds > 48 * 1024 && !c->fps_attr_set) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        c->fps_attr_set = 1;
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
