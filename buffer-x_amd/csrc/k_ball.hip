// k_ball.hip -- neighbour gather: MiniSpinNet.select_patches (reference models/patch_embedder.py:92-120) =
// pointnet2_ops.ball_query + grouping_operation + the pad/mask arithmetic, fused; plus the cloud permutation.
//
// Semantics that must survive any acceleration: per keypoint the FIRST P points, in the
// order of the (permuted) cloud, with fp32 d2 < r2 (strict, un-fused ((dx*dx+dy*dy)+dz*dz)); unfilled slots
// repeat the first hit; slots equal to the first hit (except slot 0) and slot P-1 become the keypoint.
//
// Round-2 design (exact, grid-accelerated; replaces the O(K*N) streaming scan):
//   1. a uniform grid with cell edge h >= r(1+pad) is built over the permuted cloud per launch (the radius only
//      exists on the device): bbox -> counting sort by cell (atomic rank, 2-kernel scan, scatter).  A sorted
//      entry is {x, y, z, bits(i)} with i = position in the permuted cloud, one 16-byte load per candidate.
//   2. query: one 4-wave workgroup per keypoint.  The (y,z) cell rows around the keypoint are laid end to end into one
//      flat candidate sequence (a row's x-cells are ONE contiguous range of the sorted array), so candidates stream
//      in with coalesced 16-B loads, 8 in flight per lane.  Every hit sets bit i of an n-bit bitmap in LDS
//      (ds_or_b32).  The bitmap restores the reference's order for free: set bits in increasing i ARE the
//      ball_query output order, whatever order the candidates were visited in.
//   3. ordered expansion: every thread owns a run of bitmap words; popcount + prefix over the workgroup gives it the
//      output rank of its first hit; it peels its bits (ctz) into an LDS index list, stopping at P.
//   4. output: lane j reads list[j], gathers {x,y,z} with one 16-B load from the float4 copy of the cloud
//      (L2 resident), applies the mask arithmetic and stores 12 contiguous bytes (global_store_dwordx3) -- a wave
//      writes 768 contiguous bytes per instruction.
//   Cell membership is monotone in the coordinate (fp32 subtract, multiply by a positive, floor, clamp), so a
//   point with fp32 d2 < r2 always lies inside the visited cell range (pad covers the rounding of d2 and of q-r).
// Algorithmic HBM bytes per launch: 12N + 12K + 4KP + 12KP; the tests done drop from K*N to
// roughly K * (points in 27 cells).
#include "bx_common.h"
#include <cstdlib>
#include <cstring>


#ifndef BX_BALL_NOPRE_LOGC
#define BX_BALL_NOPRE_LOGC 5       // clouds of more than 65 536 points
#endif

namespace {

__global__ void permute_kernel(const float* __restrict__ pts, const int32_t* __restrict__ perm, int n, float* __restrict__ out,
                               const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        size_t s = (size_t)perm[i] * 3;
        out[(size_t)i * 3 + 0] = pts[s + 0];
        out[(size_t)i * 3 + 1] = pts[s + 1];
        out[(size_t)i * 3 + 2] = pts[s + 2];
    }
}

// order-preserving float <-> int map for atomicMin / atomicMax
__device__ __forceinline__ int f2ord(float f)
{
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// ---------------------------------------------------------------------------------------------- batched grid build
// One pair needs 2 clouds x S scales grids (the cell edge follows the radius of the scale).  All of them are built by ONE
// sequence of six launches whose blockIdx.y selects the grid ("set" j = cloud * S + scale): the 9 small launches per stage call
// of round 1 (x 6 calls per pair) had become longer than the query kernel itself.  The permutation of the cloud
// (models/patch_embedder.py:96-97) is applied on the fly: point i of set j is pts[perm_j[i]].
struct BallBatch {
    const float* pts[2];        // cloud of sets [0, S) and [S, 2S)
    const int32_t* perm[2];     // [S][n] permutations of the cloud, nullptr: identity (the stage entry point gets a permuted cloud)
    const float* kpts[2];
    int n[2];
    int S, nsets, K;
    int i0, ni;                 // the scales [i0, i0 + ni) of every cloud are in this batch: launch row y -> cloud y / ni, scale i0 + y % ni
    const int32_t* skip;        // device flag, nullptr or *skip != 0: the batch does nothing (the pair left by the early exit)
    const double* radius;       // device: radius of scale i at radius[i]
    // per-set arrays: element j at base + j * stride
    float* bbox_part;           // [2][64][6]
    BallGrid* grid;
    int32_t *cnt, *start, *bsum;
    int2* cellrank;
    float4 *pts4, *sorted;
    int2* ptab;                 // [nsets][K][NPMAX] candidate pieces {first slot in the sorted array, count <= 1 << logpw}
    int32_t* pnum;              // [nsets][K] number of pieces, -1: degenerate geometry (the query walks the cells itself)
    size_t st_cnt, st_bsum, st_pts, st_tab, st_num;
    int logpw[BX_MAX_SCALES];   // piece width of scale i: 64 / 16 / 8 candidates by the expected length of a cell row
};

// launch row -> (set, cloud, scale) of a batch; false when the batch is switched off
__device__ __forceinline__ bool ball_set(const BallBatch& B, int y, int& j, int& cl, int& sc)
{
    cl = y / B.ni;
    sc = B.i0 + (y - cl * B.ni);
    j = cl * B.S + sc;
    return !(B.skip != nullptr && *B.skip != 0);
}

// per-block partial bounds of a cloud (64 blocks x 2 clouds, no atomics, no initialisation launch)
__global__ __launch_bounds__(1024) void bbox_kernel(BallBatch B)
{
    __shared__ float red[16][6];
    if (B.skip != nullptr && *B.skip != 0) return;
    const int cl = blockIdx.y;
    const float* __restrict__ pts = B.pts[cl];
    const int n = B.n[cl];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (pts)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = pts[(size_t)i * 3 + c];
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], s, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], s, 64));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[wave][c] = lo[c]; red[wave][3 + c] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float v = red[0][c];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = c < 3 ? fminf(v, red[w][c]) : fmaxf(v, red[w][c]);
        B.bbox_part[((size_t)cl * 64 + blockIdx.x) * 6 + c] = v;
    }
}

// one wave per set: grid geometry from the cloud's bounds and the device-side radius of the set's scale
__global__ __launch_bounds__(64) void grid_setup_kernel(BallBatch B, int div)
{
    int j, cl, sc;
    if (!ball_set(B, blockIdx.x, j, cl, sc)) return;
    const int lane = threadIdx.x;
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float x = B.bbox_part[((size_t)cl * 64 + lane) * 6 + c];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const float o = __shfl_xor(x, s, 64);
            x = c < 3 ? fminf(x, o) : fmaxf(x, o);
        }
        v[c] = x;
    }
    if (lane != 0) return;
    float lo[3], hi[3], ext[3];
    float maxabs = 1.0f;
    for (int c = 0; c < 3; ++c) {
        lo[c] = v[c];
        hi[c] = v[3 + c];
        if (!(hi[c] >= lo[c])) { lo[c] = 0.f; hi[c] = 0.f; }   // empty / NaN cloud
        ext[c] = hi[c] - lo[c];
        maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[c]), fabsf(hi[c])));
    }
    float r = (float)(B.radius[sc]);
    if (!(r > 0.f)) r = 0.f;
    // pad: covers the rounding of d2 (a hit can have |dx| up to r(1+5 eps)) and of fl(q -+ rpad)
    float rpad = r + (r * 1.0e-3f + 4.0e-7f * maxabs);
    float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    float h = fmaxf(rpad / (float)div, fmaxf(emax * (1.0f / 1024.0f), 1.0e-20f));
    int d[3];
    for (int it = 0; it < 200; ++it) {
        float inv = 1.0f / h;
        long long tot = 1;
        for (int c = 0; c < 3; ++c) {
            float f = floorf(ext[c] * inv);
            d[c] = (int)fminf(f, 1023.0f) + 1;
            tot *= d[c];
        }
        if (tot <= BX_BALL_NCELL) break;
        h = h * 1.2599211f;
    }
    BallGrid* g = B.grid + j;
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2];
    g->inv_h = 1.0f / h;
    g->rpad = rpad;
    g->dx = d[0]; g->dy = d[1]; g->dz = d[2];
    g->ncells = d[0] * d[1] * d[2];
}

__device__ __forceinline__ int cell_coord(float x, float o, float inv_h, int d)
{
    float f = floorf((x - o) * inv_h);
    f = fminf(fmaxf(f, 0.0f), (float)(d - 1));   // NaN -> 0
    return (int)f;
}

// Rank of every point inside its cell (the order inside a cell is arbitrary; the query result does not depend on
// it).  Same-address device atomics serialise (~2 ns each measured with 100 hot cells), so when the grid fits the
// 64 KiB LDS histogram each workgroup ranks its points with LDS atomics and reserves one range per (workgroup,
// cell) with a single device atomic; finer grids (little contention) use the device atomic per point.
// cnt[] is all zeros on entry: scan_apply_kernel clears what it has consumed (and bx_create zeroes it once).
constexpr int COUNT_LDS_CELLS = 16384;
constexpr int COUNT_PPT = 4;   // points per thread (register-resident between the two phases)

__global__ __launch_bounds__(1024) void cell_count_kernel(BallBatch B)
{
    __shared__ int hist[COUNT_LDS_CELLS];
    int j, cl, sc;
    if (!ball_set(B, blockIdx.y, j, cl, sc)) return;
    const int n = B.n[cl];
    const int base = blockIdx.x * (1024 * COUNT_PPT);
    if (base >= n) return;
    const float* __restrict__ pts = B.pts[cl];
    const int32_t* __restrict__ perm = B.perm[cl] ? B.perm[cl] + (size_t)sc * n : nullptr;
    const BallGrid* __restrict__ g = B.grid + j;
    int32_t* __restrict__ cnt = B.cnt + (size_t)j * B.st_cnt;
    int2* __restrict__ cellrank = B.cellrank + (size_t)j * B.st_pts;
    float4* __restrict__ pts4 = B.pts4 + (size_t)j * B.st_pts;
    const float ox = g->ox, oy = g->oy, oz = g->oz, ih = g->inv_h;
    const int dx = g->dx, dy = g->dy, dz = g->dz, ncells = g->ncells;
    const bool use_lds = ncells <= COUNT_LDS_CELLS;
    const int tid = threadIdx.x;
    if (use_lds) {
        for (int c = tid; c < ncells; c += 1024) hist[c] = 0;
        __syncthreads();
    }
    int cell[COUNT_PPT], rank[COUNT_PPT];
#pragma unroll
    for (int u = 0; u < COUNT_PPT; ++u) {
        const int i = base + u * 1024 + tid;
        cell[u] = -1; rank[u] = 0;
        if (i < n) {
            const size_t src = perm ? (size_t)perm[i] : (size_t)i;
            float x = pts[src * 3], y = pts[src * 3 + 1], z = pts[src * 3 + 2];
            pts4[i] = make_float4(x, y, z, 0.f);
            cell[u] = (cell_coord(z, oz, ih, dz) * dy + cell_coord(y, oy, ih, dy)) * dx + cell_coord(x, ox, ih, dx);
            rank[u] = use_lds ? atomicAdd(&hist[cell[u]], 1) : atomicAdd(&cnt[cell[u]], 1);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int c = tid; c < ncells; c += 1024) {
            const int h = hist[c];
            if (h > 0) hist[c] = atomicAdd(&cnt[c], h);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < COUNT_PPT; ++u) {
        const int i = base + u * 1024 + tid;
        if (i < n) cellrank[i] = make_int2(cell[u], rank[u] + (use_lds ? hist[cell[u]] : 0));
    }
}

constexpr int SCAN_TILE = 2048;   // cells per 256-thread workgroup

__global__ __launch_bounds__(256) void scan_sums_kernel(BallBatch B)
{
    __shared__ int ws[4];
    int j, cl_, sc_;
    if (!ball_set(B, blockIdx.y, j, cl_, sc_)) return;
    const int32_t* __restrict__ cnt = B.cnt + (size_t)j * B.st_cnt;
    const int base = blockIdx.x * SCAN_TILE;
    int s = 0;
    if (base <= (B.grid + j)->ncells) {
        const int4* p = reinterpret_cast<const int4*>(cnt + base) + threadIdx.x * 2;
        int4 a = p[0], b = p[1];
        s = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
    }
    s = bx_wave_sum_i(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) B.bsum[(size_t)j * B.st_bsum + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// exclusive scan of cnt[0..ncells] -> start[0..ncells] (cnt[ncells] == 0, so start[ncells] == n); cnt is cleared behind the scan
__global__ __launch_bounds__(256) void scan_apply_kernel(BallBatch B)
{
    __shared__ int ws[4];
    __shared__ int boff;
    int j, cl_, sc_;
    if (!ball_set(B, blockIdx.y, j, cl_, sc_)) return;
    int32_t* __restrict__ cnt = B.cnt + (size_t)j * B.st_cnt;
    int32_t* __restrict__ start = B.start + (size_t)j * B.st_cnt;
    const int32_t* __restrict__ bsum = B.bsum + (size_t)j * B.st_bsum;
    const int base = blockIdx.x * SCAN_TILE;
    if (base > (B.grid + j)->ncells) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {   // offset of this tile = sum of the preceding tiles (<= BX_BALL_NCELL / SCAN_TILE + 1 values)
        int v = 0;
        for (int b = lane; b < (int)blockIdx.x; b += 64) v += bsum[b];
        v = bx_wave_sum_i(v);
        if (lane == 0) boff = v;
    }
    int4* p = reinterpret_cast<int4*>(cnt + base) + tid * 2;
    int4 a = p[0], b = p[1];
    p[0] = make_int4(0, 0, 0, 0); p[1] = make_int4(0, 0, 0, 0);
    int tsum = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
    int inc = bx_wave_incl_scan_dpp(tsum);
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    int off = boff + (inc - tsum);
    for (int w = 0; w < wave; ++w) off += ws[w];
    int4 oa, ob;
    oa.x = off; oa.y = oa.x + a.x; oa.z = oa.y + a.y; oa.w = oa.z + a.z;
    ob.x = oa.w + a.w; ob.y = ob.x + b.x; ob.z = ob.y + b.y; ob.w = ob.z + b.z;
    int4* o = reinterpret_cast<int4*>(start + base) + tid * 2;
    o[0] = oa; o[1] = ob;
}

__global__ __launch_bounds__(256) void cell_scatter_kernel(BallBatch B)
{
    int j, cl, sc_;
    if (!ball_set(B, blockIdx.y, j, cl, sc_)) return;
    const int n = B.n[cl];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 cr = (B.cellrank + (size_t)j * B.st_pts)[i];
    float4 p = (B.pts4 + (size_t)j * B.st_pts)[i];
    p.w = __int_as_float(i);
    (B.sorted + (size_t)j * B.st_pts)[(B.start + (size_t)j * B.st_cnt)[cr.x] + cr.y] = p;
}

struct F3 { float x, y, z; };   // 4-byte aligned triple: stores compile to global_store_dwordx3

constexpr int MAXROWS = 63;      // (y,z) cell rows kept in the per-wave row table (one lane each)

constexpr int NPMAX = 256;       // piece-table capacity per keypoint

// Candidate pieces of all keypoints, one wave per keypoint.  The candidates of a keypoint are the points of the (y,z) cell rows
// around it, every row trimmed to the chord of the ball (a row's x-cells are ONE contiguous range of the sorted array).  Rows are
// cut into PIECES of at most PW = 1 << logpw consecutive slots; the query kernel then needs no position -> row arithmetic at all:
// lane group g of an iteration takes piece (iteration, wave, g) = {first slot, count} and lane li of the group tests slot
// first + li.  (Round 1 laid the rows end to end into one flat sequence and mapped 64-candidate chunks back to rows with bit-mask
// tables: ~40 instructions of addressing per chunk, more than the distance test itself.)  PW follows the expected row length:
// 64 at the 5 % scale (rows of ~50 candidates), 16 at 2 %, 8 at 0.5 % (rows of ~4).
__global__ __launch_bounds__(256) void ball_rows_kernel(BallBatch B, int trim)
{
    int j_, cl_, sc_;
    if (!ball_set(B, blockIdx.y, j_, cl_, sc_)) return;
    const int32_t* __restrict__ start = B.start + (size_t)j_ * B.st_cnt;
    const BallGrid* __restrict__ g = B.grid + j_;
    const float* __restrict__ kpts = B.kpts[cl_];
    const int K = B.K;
    int2* __restrict__ ptab = B.ptab + (size_t)j_ * B.st_tab;
    int32_t* __restrict__ pnum = B.pnum + (size_t)j_ * B.st_num;
    const int logpw = B.logpw[sc_], PW = 1 << logpw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= K) return;
    const float qx = kpts[(size_t)q * 3], qy = kpts[(size_t)q * 3 + 1], qz = kpts[(size_t)q * 3 + 2];
    const float ox = g->ox, oy = g->oy, oz = g->oz, ih = g->inv_h, rp = g->rpad;
    const int dx = g->dx, dy = g->dy, dz = g->dz;
    const int xlo = cell_coord(qx - rp, ox, ih, dx), xhi = cell_coord(qx + rp, ox, ih, dx);
    const int ylo = cell_coord(qy - rp, oy, ih, dy), yhi = cell_coord(qy + rp, oy, ih, dy);
    const int zlo = cell_coord(qz - rp, oz, ih, dz), zhi = cell_coord(qz + rp, oz, ih, dz);
    const int ny = yhi - ylo + 1, nz = zhi - zlo + 1;
    const int R0 = ny * nz;
    if (R0 > MAXROWS) {
        if (lane == 0) pnum[q] = -1;
        return;
    }
    int st = 0, len = 0;
    if (lane < R0) {
        const int cz = zlo + lane / ny, cy = ylo + lane % ny;
        const int rowc = (cz * dy + cy) * dx;
        // chord trimming in cell units: every point of this row has u_y in [cy, cy+1], u_z in [cz, cz+1] (u = the value
        // whose floor binned it), so a hit's |u_x - uq_x| is bounded by the chord of the (padded) ball at the row's
        // smallest possible (y,z) distance.  DU covers the rounding of u (|u| <= 1024, two roundings) on both sides.
        constexpr float DU = 1.0e-3f;
        const float uqx = (qx - ox) * ih, uqy = (qy - oy) * ih, uqz = (qz - oz) * ih;
        const float gy = fmaxf(fmaxf((float)cy - uqy, uqy - (float)(cy + 1)) - DU, 0.0f);
        const float gz = fmaxf(fmaxf((float)cz - uqz, uqz - (float)(cz + 1)) - DU, 0.0f);
        const float m = rp * ih + DU;
        const float w2 = m * m - (gy * gy + gz * gz);
        int xl = xlo, xh = xhi;
        if (trim) {
            if (w2 < 0.0f) { xl = 1; xh = 0; }
            else {
                const float w = sqrtf(w2) * 1.0001f + DU;
                const float fl = fminf(fmaxf(floorf(uqx - w), 0.0f), (float)(dx - 1));
                const float fh = fminf(fmaxf(floorf(uqx + w), 0.0f), (float)(dx - 1));
                xl = max(xlo, (int)fl);
                xh = min(xhi, (int)fh);
            }
        }
        if (xl <= xh) {
            st = start[rowc + xl];
            len = start[rowc + xh + 1] - st;
        }
    }
    // pieces of every row, laid out row after row (the order is immaterial: the hit bitmap restores the index order)
    const int np = (len + PW - 1) >> logpw;
    const int inc = bx_wave_incl_scan_dpp(np);
    const int NP = __builtin_amdgcn_readlane(inc, 63);
    if (NP > NPMAX) {                                        // very long candidate sequences: the query walks the cells itself
        if (lane == 0) pnum[q] = -1;
        return;
    }
    int2* __restrict__ pq = ptab + (size_t)q * NPMAX + (inc - np);
    for (int i = 0; i < np; ++i) pq[i] = make_int2(st + (i << logpw), min(PW, len - (i << logpw)));
    if (lane == 0) pnum[q] = NP;
}

// QW waves per workgroup, ONE keypoint per workgroup (template parameter: 4 / 2 / 1 by the expected neighbourhood size)

// Bit i of the hit bitmap = bit (i & 31) of 32-bit word i >> 5 (plain layout: 2 VALU + ds_or per hit).
__device__ __forceinline__ void set_hit(unsigned int* bm32, int i)
{
    atomicOr(&bm32[i >> 5], 1u << (i & 31));
}

// QW-wave workgroup per keypoint (one wave per keypoint left the kernel waiting for its slowest keypoints: the candidate count
// varies 3x between keypoints).  LOGC: log2 of the 64-bit bitmap words per 64 lanes (n <= 4096 << LOGC).
// The kernel is VALU-issue bound (rocprofv3 SQ_INSTS_VALU: ~800 instructions per wave at 8 waves per SIMD), not memory bound:
// the structure below is chosen for instruction count.
//   1. candidates = the pieces prepared by ball_rows_kernel, staged in LDS and dealt round-robin to the lane groups of the waves;
//      a lane's candidate is  piece.first + (lane & (PW - 1)):  no position -> row arithmetic.
//   2. hits set their bit in an LDS bitmap (bit = permuted point index): this restores the index order the reference's
//      ball_query scans in, whatever order the grid delivered the candidates in.
//   3. rank of a hit = set bits below it = prefix[word] + popcount(word & below): when all candidates of the keypoint were held in
//      registers (up to two blocks of QU chunks per wave: the common case) every hit computes its own rank and drops its index
//      into list[rank] -- no per-bit loops; otherwise the owners of the bitmap words expand their bits in order.
//   4. output: gather + mask arithmetic from the ordered list, 768 contiguous bytes per wave store.
// LDS: bitmap (n/8 B) | per-word prefix (n/16 B) | list (4 P B) | piece table (8 NPMAX B).
template <int LOGC, int QW>
__global__ __launch_bounds__(64 * QW, 8) void ball_query_kernel(const float4* __restrict__ sorted, const int32_t* __restrict__ start,
                                                        const BallGrid* __restrict__ g, const int2* __restrict__ ptab,
                                                        const int32_t* __restrict__ pnum, int logpw, const float4* __restrict__ pts4,
                                                        const float* __restrict__ kpts, int K,
                                                        const double* __restrict__ radius, int P, int32_t* __restrict__ idx_out,
                                                        float* __restrict__ patches, int32_t* __restrict__ cnt_out,
                                                        const int32_t* __restrict__ skip, long long* __restrict__ dbg)
{
    if (skip && *skip) return;
    constexpr int QT = 64 * QW;
    constexpr int NW64 = 64 << LOGC;                    // 64-bit bitmap words
    constexpr int CW = NW64 / QT;                       // consecutive bitmap words owned by a thread
    static_assert(CW >= 2 && CW % 2 == 0, "a thread owns an even number of bitmap words");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int wtot[QW];
    long long t0 = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: the chunk walk below is scalar code
    const int q = blockIdx.x;
    const bool tr = dbg != nullptr && (q % 79) == 0 && q / 79 < 60 && tid == 0;
    long long* td = dbg + (q / 79) * 8;
    if (tr) { t0 = __builtin_readcyclecounter(); td[0] = t0; }
#define BX_TR(k) do { if (tr) td[k] = __builtin_readcyclecounter() - t0; } while (0)
    unsigned long long* bm64 = reinterpret_cast<unsigned long long*>(smem);
    unsigned int* bm32 = reinterpret_cast<unsigned int*>(smem);
    // large clouds (LOGC >= BX_BALL_NOPRE_LOGC: bitmaps of 16 KB and more): no per-word prefix array -- the owners of the bitmap words expand
    // their bits themselves (the self-ranking of the hits needs the array) -- so that the workgroup's LDS shrinks by a third and more
    // keypoints are resident per CU
    constexpr bool NOPRE = LOGC >= BX_BALL_NOPRE_LOGC;
    int* pre = reinterpret_cast<int*>(smem + (size_t)NW64 * 8);            // [NW64] set bits in front of every 64-bit word (!NOPRE)
    int* list = reinterpret_cast<int*>(smem + (size_t)NW64 * (NOPRE ? 8 : 12));          // [P]

    {
        ulonglong2* z = reinterpret_cast<ulonglong2*>(bm64 + (size_t)tid * CW);
#pragma unroll
        for (int s = 0; s < CW / 2; ++s) z[s] = make_ulonglong2(0ULL, 0ULL);
    }

    int2* ptl = reinterpret_cast<int2*>(smem + (size_t)NW64 * (NOPRE ? 8 : 12) + (((size_t)P * 4 + 7) & ~(size_t)7));   // [NPMAX] piece table of this keypoint
    const int NP = __builtin_amdgcn_readfirstlane(pnum[q]);                          // -1: degenerate geometry
    for (int i = tid; i < NP; i += QT) ptl[i] = ptab[(size_t)q * NPMAX + i];
    const float r = (float)(*radius);
    const float r2 = r * r;
    const float qx = kpts[(size_t)q * 3], qy = kpts[(size_t)q * 3 + 1], qz = kpts[(size_t)q * 3 + 2];
    BX_TR(1);
    __syncthreads();                                        // bitmap zeroed, piece table staged
    if (tr) td[7] = NP;

    constexpr int MAXIT = 16;                               // iterations whose hits are kept in registers and rank themselves
    const int GP = 64 >> logpw;                             // pieces per wave and iteration
    const int nit = NP > 0 ? (NP + QW * GP - 1) / (QW * GP) : 0;   // uniform
    const bool one_block = !NOPRE && NP >= 0 && nit <= MAXIT;
    int hidx[MAXIT];                                        // one_block: index of the hit of iteration it, -1 otherwise
#pragma unroll
    for (int u = 0; u < MAXIT; ++u) hidx[u] = -1;
    if (NP >= 0) {
        const int grp = lane >> logpw, li = lane & ((1 << logpw) - 1);
        // a batch of 4 iterations: the four table reads, then the four candidate loads go out back to back (a load under a
        // divergent branch would be waited for right behind its issue), then the four tests
        auto scan4 = [&](int it0, int* hx) {
            float4 c[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pc = ((it0 + u) * QW + wave) * GP + grp;
                int2 e = make_int2(0, 0);
                if (pc < NP) e = ptl[pc];
                ok[u] = li < e.y;
                const unsigned addr = ok[u] ? (unsigned)(e.x + li) : 0u;     // lanes without a candidate re-read slot 0 (masked below)
                c[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sorted) + (addr << 4));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ax = qx - c[u].x, ay = qy - c[u].y, az = qz - c[u].z;
                const float d2 = (ax * ax + ay * ay) + az * az;
                if (d2 < r2 && ok[u]) {
                    const int i = __float_as_int(c[u].w);
                    set_hit(bm32, i);
                    hx[u] = i;
                }
            }
        };
        if (one_block) {
#pragma unroll
            for (int b = 0; b < MAXIT / 4; ++b)
                if (b * 4 < nit) scan4(b * 4, hidx + b * 4);        // uniform
        } else {
            int dump[4];
            for (int it0 = 0; it0 < nit; it0 += 4) scan4(it0, dump);
        }
    } else {
        // degenerate geometry (cell edge << radius because of the 1024-cells-per-axis floor): plain nested walk
        const float ox = g->ox, oy = g->oy, oz = g->oz, ih = g->inv_h, rp = g->rpad;
        const int dx = g->dx, dy = g->dy, dz = g->dz;
        const int xlo = cell_coord(qx - rp, ox, ih, dx), xhi = cell_coord(qx + rp, ox, ih, dx);
        const int ylo = cell_coord(qy - rp, oy, ih, dy), yhi = cell_coord(qy + rp, oy, ih, dy);
        const int zlo = cell_coord(qz - rp, oz, ih, dz), zhi = cell_coord(qz + rp, oz, ih, dz);
        int rowi = 0;
        for (int cz = zlo; cz <= zhi; ++cz)
            for (int cy = ylo; cy <= yhi; ++cy, ++rowi) {
                if ((rowi & (QW - 1)) != wave) continue;
                const int rowc = (cz * dy + cy) * dx;
                const int s = start[rowc + xlo], e = start[rowc + xhi + 1];
                for (int k = s + lane; k < e; k += 64) {
                    const float4 a = sorted[k];
                    const float ax = qx - a.x, ay = qy - a.y, az = qz - a.z;
                    const float d2 = (ax * ax + ay * ay) + az * az;
                    if (d2 < r2) set_hit(bm32, __float_as_int(a.w));
                }
            }
    }
    __syncthreads();
    BX_TR(3);

    // ---- set bits in front of every thread's CW words (thread t owns indices [t*64*CW, (t+1)*64*CW)); the words are re-read
    //      from LDS where they are needed again instead of being held in registers (64-VGPR budget: 8 waves per SIMD)
    const ulonglong2* wsrc = reinterpret_cast<const ulonglong2*>(bm64 + (size_t)tid * CW);
    int tot = 0;
#pragma unroll 4
    for (int s = 0; s < CW / 2; ++s) { const ulonglong2 t2 = wsrc[s]; tot += __popcll(t2.x) + __popcll(t2.y); }
    const int inc = bx_wave_incl_scan_dpp(tot);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();                                        // wave totals visible
    int pos = inc - tot, run = 0;
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const int wt = wtot[w];
        if (w < wave) pos += wt;
        run += wt;
    }
    const int nhit = run < P ? run : P;
    if (one_block) {
        // every hit ranks itself: set bits in front of its word + set bits below it inside the word
        {
            int2* pd = reinterpret_cast<int2*>(pre + (size_t)tid * CW);
            int pp = pos;
#pragma unroll 4
            for (int s = 0; s < CW / 2; ++s) {
                const ulonglong2 t2 = wsrc[s];
                const int p1 = pp + __popcll(t2.x);
                pd[s] = make_int2(pp, p1);
                pp = p1 + __popcll(t2.y);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXIT; ++u) {
            const int i = hidx[u];
            if (i >= 0) {
                const int w = i >> 6;
                const unsigned long long below = bm64[w] & ((1ULL << (i & 63)) - 1ULL);
                const int rank = pre[w] + __popcll(below);
                if (rank < P) list[rank] = i;
            }
        }
    } else if (tot > 0 && pos < P) {
        // owners expand their bits in order
#pragma unroll 2
        for (int s = 0; s < CW; ++s) {
            unsigned long long w = bm64[(size_t)tid * CW + s];
            const int base = (tid * CW + s) << 6;
            while (w != 0ULL && pos < P) {
                const int b = __ffsll((long long)w) - 1;
                list[pos++] = base + b;
                w &= w - 1ULL;
            }
        }
    }
    __syncthreads();
    BX_TR(4);

    // ---- output: gather + mask arithmetic (models/patch_embedder.py:105-111), 768 contiguous bytes per wave store
    const int first = nhit > 0 ? list[0] : 0;
    F3* out3 = reinterpret_cast<F3*>(patches) + (size_t)q * P;
    int32_t* outi = idx_out ? idx_out + (size_t)q * P : nullptr;
    // Hit-count hand-over (cnt_out != nullptr, the whole-pair path): slots [nreal, P) of a patch are copies of the keypoint -- the padded
    // slots (group_idx == group_idx[0]) and slot P - 1 (models/patch_embedder.py:105-111) -- so only the REAL slots are written, with
    // nreal = clamp(hits, 1, P - 1): slot 0 is always a cloud point (the first hit, or point 0 of the permuted cloud when the ball is
    // empty: the reference's quirk), slot P - 1 never is.  patch_axis_kernel / patch_features_kernel take the count and treat the rest
    // analytically (k_patch.hip); a 0.5 % patch is ~83 % copies of the keypoint that are neither gathered, written nor read back.
    const int nout = cnt_out ? (nhit < 1 ? 1 : (nhit < P - 1 ? nhit : P - 1)) : P;
    if (cnt_out && tid == 0) cnt_out[q] = nout;
    constexpr int OU = QW >= 4 ? 4 : 8;     // P = 1024: one pass, every gather of the keypoint in flight at once
    for (int j0 = 0; j0 < nout; j0 += QT * OU) {
        int idx[OU];
        float4 p[OU];
#pragma unroll
        for (int u = 0; u < OU; ++u) {
            const int j = j0 + u * QT + tid;
            idx[u] = j < nhit ? list[j] : first;
        }
#pragma unroll
        for (int u = 0; u < OU; ++u) p[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(pts4) + ((unsigned)idx[u] << 4));
#pragma unroll
        for (int u = 0; u < OU; ++u) {
            const int j = j0 + u * QT + tid;
            if (j < nout) {
                float mask = (idx[u] == first) ? 1.0f : 0.0f;
                if (j == 0) mask = 0.0f;
                if (j == P - 1) mask = 1.0f;
                const float om = 1.0f - mask;
                F3 o;
                o.x = p[u].x * om + qx * mask;
                o.y = p[u].y * om + qy * mask;
                o.z = p[u].z * om + qz * mask;
                out3[j] = o;
                if (outi) outi[j] = idx[u];
            }
        }
    }
    BX_TR(5);
    if (tr) { __builtin_amdgcn_s_waitcnt(0); td[6] = __builtin_readcyclecounter() - t0; }
}

template <int LOGC, int QW>
int launch_query_w(bx_ctx* c, hipStream_t s, int set, int k0, int K, const float* kpts, const double* radius, int P, int32_t* idx_out, float* patches_out, int32_t* cnt_out)
{
    const size_t lds = ((size_t)64 << LOGC) * (LOGC >= BX_BALL_NOPRE_LOGC ? 8 : 12) + (((size_t)P * 4 + 7) & ~(size_t)7) + (size_t)NPMAX * 8;   // bitmap | per-word prefix | ordered index list | pieces
    if (lds > 160 * 1024) { bx_set_error("bxk_ball_group: P=%d needs %zu B of LDS per keypoint (> 160 KiB)", P, lds); return BX_ERR_ARG; }
    if (lds > 64 * 1024 && !(c->ball_attr_set & (1LL << (LOGC * 3 + QW / 2)))) {
        BX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ball_query_kernel<LOGC, QW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        c->ball_attr_set |= 1 << (LOGC * 3 + QW / 2);
    }
    const size_t j = (size_t)set;
    hipLaunchKernelGGL((ball_query_kernel<LOGC, QW>), dim3(K), dim3(64 * QW), lds, s, c->ball_sorted + j * c->ball_st_pts,
                       c->ball_start + j * c->ball_st_cnt, c->ball_grid + j, c->ball_ptab + j * c->ball_st_tab + (size_t)k0 * NPMAX,
                       c->ball_pnum + j * c->ball_st_num + k0, c->ball_logpw[set], c->ball_pts4 + j * c->ball_st_pts, kpts, K,
                       radius, P, idx_out, patches_out, cnt_out, c->skip, getenv("BX_BALL_DEBUG") ? c->ball_dbg : nullptr);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

template <int LOGC>
int launch_query(bx_ctx* c, hipStream_t s, int set, int k0, int K, const float* kpts, const double* radius, int P, int32_t* idx_out, float* patches_out, int32_t* cnt_out)
{
    // waves per keypoint: 4 for the large neighbourhoods (their candidate scan dominates), 1 for the small ones (the
    // per-keypoint chain of dependent memory round trips dominates and more independent workgroups hide it better)
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("BX_BALL_WAVES"); forced = e ? atoi(e) : 0; }
    // measured (K = 5000, P = 1024): 4 waves win whenever the bitmap sweep is long (n > 32768: LOGC >= 4) or the neighbourhood is
    // large (hint from the radius threshold); 2 waves only for small neighbourhoods in small clouds
    const int w = forced ? forced : (LOGC >= 4 ? 4 : c->ball_waves_hint);
    if (w >= 4) return launch_query_w<LOGC, 4>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out);
    if (w == 1) return launch_query_w<LOGC, 1>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out);
    return launch_query_w<LOGC, 2>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out);
}
}  // namespace

static int bx_ball_trim()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("BX_BALL_TRIM"); v = e ? atoi(e) : 1; }
    return v;
}

int bx_ball_div()
{
    static int v = 0;
    if (!v) {
        const char* e = getenv("BX_BALL_DIV");
        int d = e ? atoi(e) : 3;
        v = d < 1 ? 1 : (d > 3 ? 3 : d);
    }
    return v;
}

int bx_permute_launch(hipStream_t s, const float* pts, const int32_t* perm, int n, float* out, const int32_t* skip)
{
    if (n <= 0) return BX_OK;
    hipLaunchKernelGGL(permute_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, perm, n, out, skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

namespace {
int ball_logc(int n)
{
    int logc = 3;
    while (((size_t)64 << logc) * 64 < (size_t)n) ++logc;   // 64 lanes x C words x 64 bits >= n
    return logc;
}
}  // namespace

namespace {
// the per-set array bases and strides of the context
void batch_from_ctx(const bx_ctx* c, BallBatch& B)
{
    memset(&B, 0, sizeof(B));
    B.bbox_part = c->ball_bbox_part; B.grid = c->ball_grid; B.cnt = c->ball_cnt; B.start = c->ball_start; B.bsum = c->ball_bsum;
    B.cellrank = c->ball_cellrank; B.pts4 = c->ball_pts4; B.sorted = c->ball_sorted; B.ptab = c->ball_ptab; B.pnum = c->ball_pnum;
    B.st_cnt = c->ball_st_cnt; B.st_bsum = c->ball_st_bsum; B.st_pts = c->ball_st_pts; B.st_tab = c->ball_st_tab; B.st_num = c->ball_st_num;
}
}  // namespace

// Cell grids of `nclouds` clouds x S scales in six launches (see BallBatch).  clouds / perms: per cloud; perm[cl] is [S][n] or
// nullptr (identity).  radius: device array of S doubles.  Independent of the keypoints.
int bxk_ball_grids(bx_ctx* c, hipStream_t s, const float* const* clouds, const int* ns, const int32_t* const* perms, int nclouds,
                   const double* radius, int S, const double* pw_hint, int i0, int ni)
{
    if (ni < 0) ni = S - i0;
    if (i0 < 0 || ni < 1 || i0 + ni > S) { bx_set_error("bxk_ball_grids: scales [%d, %d) of %d", i0, i0 + ni, S); return BX_ERR_ARG; }
    if (nclouds < 1 || nclouds > 2 || S < 1 || nclouds * S > c->ball_nsets) {
        bx_set_error("bxk_ball_prepare: %d clouds x %d scales exceed the context's %d grid sets", nclouds, S, c->ball_nsets);
        return BX_ERR_ARG;
    }
    BallBatch B;
    batch_from_ctx(c, B);
    int nmax = 0;
    for (int cl = 0; cl < nclouds; ++cl) {
        if (ns[cl] <= 0 || ns[cl] > c->p.max_points) { bx_set_error("bxk_ball_prepare: n=%d outside [1, max_points=%d]", ns[cl], c->p.max_points); return BX_ERR_ARG; }
        if (ball_logc(ns[cl]) > 8) { bx_set_error("bxk_ball_prepare: cloud of %d points exceeds the 1M-point bitmap", ns[cl]); return BX_ERR_ARG; }
        B.pts[cl] = clouds[cl]; B.perm[cl] = perms ? perms[cl] : nullptr; B.n[cl] = ns[cl];
        nmax = ns[cl] > nmax ? ns[cl] : nmax;
    }
    B.S = S; B.nsets = nclouds * ni; B.radius = radius; B.i0 = i0; B.ni = ni; B.skip = c->skip;
    for (int i = i0; i < i0 + ni; ++i) {
        // piece width by the share of the cloud a ball of this scale is expected to hold (its cell rows are ~1/40 of that): the
        // percentage thresholds of the pair path (cfg.patch.search_radius_thresholds), 64 when unknown (stage entry point)
        const double thr = pw_hint ? pw_hint[i] : 100.0;
        const int lp = thr >= 3.5 ? 6 : (thr >= 1.0 ? 4 : 3);
        for (int cl = 0; cl < nclouds; ++cl) c->ball_logpw[cl * S + i] = lp;
    }
    const int nb = (nmax + 255) / 256;
    const int nb4k = (nmax + 1024 * COUNT_PPT - 1) / (1024 * COUNT_PPT);
    const int ntile = BX_BALL_NCELL / SCAN_TILE + 1;
    if (i0 == 0) hipLaunchKernelGGL(bbox_kernel, dim3(64, nclouds), dim3(1024), 0, s, B);      // the bounds serve every scale
    hipLaunchKernelGGL(grid_setup_kernel, dim3(B.nsets), dim3(64), 0, s, B, bx_ball_div());
    hipLaunchKernelGGL(cell_count_kernel, dim3(nb4k, B.nsets), dim3(1024), 0, s, B);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(ntile, B.nsets), dim3(256), 0, s, B);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(ntile, B.nsets), dim3(256), 0, s, B);
    hipLaunchKernelGGL(cell_scatter_kernel, dim3(nb, B.nsets), dim3(256), 0, s, B);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

// Candidate piece tables of keypoints [k0, k0 + K) of every prepared set (one launch); kpts: per cloud, the whole keypoint array.
int bxk_ball_rows(bx_ctx* c, hipStream_t s, const float* const* kpts, int nclouds, int S, int k0, int K, int i0, int ni)
{
    if (K <= 0) return BX_OK;
    if (ni < 0) ni = S - i0;
    if (i0 < 0 || ni < 1 || i0 + ni > S) { bx_set_error("bxk_ball_rows: scales [%d, %d) of %d", i0, i0 + ni, S); return BX_ERR_ARG; }
    if (k0 < 0 || (size_t)(k0 + K) > c->ball_st_num) { bx_set_error("bxk_ball_rows: keypoints [%d, %d) exceed the table", k0, k0 + K); return BX_ERR_ARG; }
    BallBatch B;
    batch_from_ctx(c, B);
    B.S = S; B.nsets = nclouds * ni; B.K = K; B.i0 = i0; B.ni = ni; B.skip = c->skip;
    for (int cl = 0; cl < nclouds; ++cl) B.kpts[cl] = kpts[cl] + (size_t)k0 * 3;
    for (int i = 0; i < S; ++i) B.logpw[i] = c->ball_logpw[i];
    B.ptab += (size_t)k0 * NPMAX;       // the same shift inside every set
    B.pnum += k0;
    hipLaunchKernelGGL(ball_rows_kernel, dim3((K + 3) / 4, B.nsets), dim3(256), 0, s, B, bx_ball_trim());
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_ball_prepare(bx_ctx* c, hipStream_t s, const float* const* clouds, const int* ns, const int32_t* const* perms,
                     const float* const* kpts, int nclouds, int K, const double* radius, int S, const double* pw_hint)
{
    if (K <= 0) return BX_OK;
    int rc = bxk_ball_grids(c, s, clouds, ns, perms, nclouds, radius, S, pw_hint);
    if (rc != BX_OK) return rc;
    return bxk_ball_rows(c, s, kpts, nclouds, S, 0, K);
}

// The query of one prepared set: n = size of its cloud, radius = device pointer to the radius of its scale.
// Keypoints [k0, k0 + K) of the set (kpts = the whole keypoint array); rows of idx_out / patches_out are relative to k0.
// cnt_out (nullable, int32 [K] relative to k0): hit-count hand-over -- only the real slots of a patch are written + their number.
int bxk_ball_query(bx_ctx* c, hipStream_t s, int set, int n, const float* kpts, int k0, int K, const double* radius, int P,
                   int32_t* idx_out, float* patches_out, int32_t* cnt_out)
{
    if (K <= 0) return BX_OK;
    kpts += (size_t)k0 * 3;
    if (P < 2) { bx_set_error("bxk_ball_query: P=%d", P); return BX_ERR_ARG; }
    bx_prof_mark(c, s, 12, 1);
    int rc = BX_OK;
    switch (ball_logc(n)) {
    case 3: rc = launch_query<3>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out); break;
    case 4: rc = launch_query<4>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out); break;
    case 5: rc = launch_query<5>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out); break;
    case 6: rc = launch_query<6>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out); break;
    case 7: rc = launch_query<7>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out); break;
    default: rc = launch_query<8>(c, s, set, k0, K, kpts, radius, P, idx_out, patches_out, cnt_out); break;
    }
    bx_prof_mark(c, s, 12, 0);
    return rc;
}

// stage entry point (bx_ball_group): one already permuted cloud, one radius
int bxk_ball_group(bx_ctx* c, hipStream_t s, const float* pts_perm, int n, const float* kpts, int K, const double* radius,
                   int P, int32_t* idx_out, float* patches_out, int32_t* cnt_out)
{
    if (K <= 0) return BX_OK;
    if (n <= 0 || P < 2) { bx_set_error("bxk_ball_group: n=%d P=%d", n, P); return BX_ERR_ARG; }
    const float* cl[1] = {pts_perm};
    const float* kp[1] = {kpts};
    const int ns[1] = {n};
    int rc = bxk_ball_prepare(c, s, cl, ns, nullptr, kp, 1, K, radius, 1, nullptr);
    if (rc != BX_OK) return rc;
    return bxk_ball_query(c, s, 0, n, kpts, 0, K, radius, P, idx_out, patches_out, cnt_out);
}
