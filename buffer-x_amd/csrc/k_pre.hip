// k_pre.hip -- the step in front of the hot path, MI355X-native:
//   bx_pre_voxel_downsample   open3d.geometry.PointCloud.voxel_down_sample as the loaders call it
//                             (reference dataset/threedmatch.py:90-102, dataset/kitti.py, dataset/tiers.py; Open3D 0.18
//                             PointCloud::VoxelDownSample: origin = min_bound - voxel/2, index = floor((p - origin)/voxel) in
//                             binary64, voxel point = sum of the members in input order / count, binary64)
//   bx_pre_pca                compute_pca_alignment + the z-range of sphericity_based_voxel_analysis
//                             (reference utils/tools.py:132-198; sklearn PCA = covariance + symmetric eigen-decomposition)
// The reference does this on the host (Open3D hash map, scikit-learn) for every pair while the GPU waits; at > 20 pairs/s per GPU
// it becomes the throughput ceiling.
//
// voxel_down_sample: the voxel index triple of every point is inserted into an open-addressing hash table of >= 2n slots (64-bit
// atomicCAS, linear probing); the slot is the voxel's id.  Counting per slot, a multi-level exclusive scan and an atomic scatter
// build the member list of every voxel; the one thread that landed on a list's first entry orders the (few) indices of its voxel
// ascending and accumulates the members in that order in binary64 -- exactly the arithmetic of Open3D's sequential AddPoint loop,
// whatever order the atomics ran in, and with memory O(n) whatever the extent of the scene (a KITTI sweep at 2 cm voxels spans
// 10^11 grid cells).  Voxels are emitted in order of
// first appearance (scan over per-point "smallest index of its voxel" flags); Open3D's own order is unspecified and the loaders
// shuffle afterwards.
#include "bx_common.h"

namespace {
constexpr int ST = 2048;   // scan tile (256 threads x 8)

__device__ __forceinline__ int f2ord_i(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f_i(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__device__ __forceinline__ long long d2ord(double d) { long long i = __double_as_longlong(d); return i >= 0 ? i : i ^ 0x7fffffffffffffffLL; }
__device__ __forceinline__ double ord2d(long long i) { return __longlong_as_double(i >= 0 ? i : i ^ 0x7fffffffffffffffLL); }

struct PreGrid {
    double ox, oy, oz, vs;
    int32_t status;   // 1: a voxel index does not fit 21 bits per axis (voxel size too small for the extent)
    int32_t pad;
};
constexpr unsigned long long PRE_EMPTY = 0xffffffffffffffffULL;

__global__ void pre_init_kernel(int32_t* bbox, long long* zr)
{
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = (int)0x80000000;
    if (zr && threadIdx.x == 6) zr[0] = 0x7fffffffffffffffLL;
    if (zr && threadIdx.x == 7) zr[1] = (long long)0x8000000000000000LL;
}

__global__ __launch_bounds__(1024) void pre_bbox_kernel(const float* __restrict__ pts, int n, int32_t* bbox)
{
    __shared__ float red[16][6];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = pts[(size_t)i * 3 + c];
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], s, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], s, 64));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[wave][c] = lo[c]; red[wave][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float v = red[0][c];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = c < 3 ? fminf(v, red[w][c]) : fmaxf(v, red[w][c]);
        if (c < 3) atomicMin(&bbox[c], f2ord_i(v)); else atomicMax(&bbox[c], f2ord_i(v));
    }
}

__global__ void pre_setup_kernel(const int32_t* __restrict__ bbox, double vs, PreGrid* g)
{
    if (threadIdx.x != 0) return;
    double lo[3], hi[3];
    for (int c = 0; c < 3; ++c) { lo[c] = (double)ord2f_i(bbox[c]); hi[c] = (double)ord2f_i(bbox[3 + c]); }
    const double o[3] = {lo[0] - vs * 0.5, lo[1] - vs * 0.5, lo[2] - vs * 0.5};
    int status = 0;
    for (int c = 0; c < 3; ++c) {
        const double f = floor((hi[c] - o[c]) / vs);
        if (!(f >= 0.0) || f >= 2097152.0) status = 1;
    }
    g->ox = o[0]; g->oy = o[1]; g->oz = o[2]; g->vs = vs;
    g->status = status;
    g->pad = 0;
}

// voxel index triple -> hash slot (= voxel id) by linear probing; then one count per member
__global__ __launch_bounds__(256) void pre_count_kernel(const float* __restrict__ pts, int n, const PreGrid* __restrict__ gp,
                                                        unsigned long long* __restrict__ keys, unsigned tmask,
                                                        int32_t* __restrict__ cnt, int32_t* __restrict__ cellid)
{
    const PreGrid g = *gp;
    if (g.status) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long ix = (unsigned long long)(long long)floor(((double)pts[(size_t)i * 3] - g.ox) / g.vs);
    const unsigned long long iy = (unsigned long long)(long long)floor(((double)pts[(size_t)i * 3 + 1] - g.oy) / g.vs);
    const unsigned long long iz = (unsigned long long)(long long)floor(((double)pts[(size_t)i * 3 + 2] - g.oz) / g.vs);
    const unsigned long long key = (iz << 42) | (iy << 21) | ix;
    unsigned h = (unsigned)bx_mix64(key, 0x51edULL) & tmask;
    for (;;) {
        const unsigned long long old = atomicCAS(&keys[h], PRE_EMPTY, key);
        if (old == PRE_EMPTY || old == key) break;
        h = (h + 1) & tmask;
    }
    cellid[i] = (int)h;
    atomicAdd(&cnt[h], 1);
}

// ---- multi-level exclusive scan over int32 arrays of up to 2^27 elements (tiles of ST)
__global__ __launch_bounds__(256) void tile_sums_kernel(const int32_t* __restrict__ in, long long n, int32_t* __restrict__ sums)
{
    __shared__ int ws[4];
    const long long base = (long long)blockIdx.x * ST + threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (base + k < n) s += in[base + k];
    s = bx_wave_sum_i(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ void small_scan_kernel(int32_t* a, int n)   // exclusive, in place, n <= a few thousand
{
    if (threadIdx.x != 0) return;
    int run = 0;
    for (int i = 0; i < n; ++i) { const int v = a[i]; a[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void tile_scan_kernel(const int32_t* __restrict__ in, long long n, const int32_t* __restrict__ tile_off,
                                                        int32_t* __restrict__ out, int32_t* __restrict__ total)
{
    __shared__ int ws[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * ST + tid * 8;
    int v[8];
    int ts = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : 0; ts += v[k]; }
    const int inc = bx_wave_incl_scan_dpp(ts);
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    int off = tile_off[blockIdx.x] + (inc - ts);
    for (int w = 0; w < wave; ++w) off += ws[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
    }
    if (total && base <= n - 1 && n - 1 < base + 8) *total = off;   // the thread owning the last element: off == grand total
}

__global__ __launch_bounds__(256) void pre_scatter_kernel(int n, const PreGrid* __restrict__ gp, const int32_t* __restrict__ cellid,
                                                          const int32_t* __restrict__ start, int32_t* __restrict__ cnt,
                                                          int32_t* __restrict__ seg, int32_t* __restrict__ pos)
{
    if (gp->status) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cellid[i];
    const int r = atomicSub(&cnt[c], 1) - 1;
    const int slot = start[c] + r;
    seg[slot] = i;
    pos[i] = slot;
}

// the thread sitting on the first slot of a voxel's segment: order the member indices, accumulate in that order (binary64)
__global__ __launch_bounds__(256) void pre_reduce_kernel(const float* __restrict__ pts, int n, int nslots, const PreGrid* __restrict__ gp,
                                                         const int32_t* __restrict__ cellid, const int32_t* __restrict__ start,
                                                         const int32_t* __restrict__ pos, int32_t* __restrict__ seg,
                                                         float* __restrict__ cen, int32_t* __restrict__ flag)
{
    const PreGrid g = *gp;
    if (g.status) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cellid[i];
    const int s0 = start[c];
    if (pos[i] != s0) return;
    const int k = (c + 1 < nslots ? start[c + 1] : n) - s0;
    int* sg = seg + s0;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    int im;
    if (k <= 256) {
        for (int a = 1; a < k; ++a) {        // insertion sort: voxels hold a handful of points (<= 32 k moves at the cut-over)
            const int v = sg[a];
            int b = a - 1;
            while (b >= 0 && sg[b] > v) { sg[b + 1] = sg[b]; --b; }
            sg[b + 1] = v;
        }
        for (int a = 0; a < k; ++a) {
            const size_t j = (size_t)sg[a] * 3;
            sx += (double)pts[j]; sy += (double)pts[j + 1]; sz += (double)pts[j + 2];
        }
        im = sg[0];
    } else {
        // a crowded voxel (voxel size large against the sampling density): sorting its k members would be O(k^2) in one thread;
        // walking ALL points in index order and picking its members is O(n), bounded, and the same summation order
        im = -1;
        int left = k;
        for (int j = 0; j < n && left > 0; ++j) {
            if (cellid[j] != c) continue;
            if (im < 0) im = j;
            sx += (double)pts[(size_t)j * 3]; sy += (double)pts[(size_t)j * 3 + 1]; sz += (double)pts[(size_t)j * 3 + 2];
            --left;
        }
    }
    const double dk = (double)k;
    cen[(size_t)im * 3] = (float)(sx / dk);
    cen[(size_t)im * 3 + 1] = (float)(sy / dk);
    cen[(size_t)im * 3 + 2] = (float)(sz / dk);
    flag[im] = 1;
}

__global__ __launch_bounds__(256) void pre_compact_kernel(int n, const PreGrid* __restrict__ gp, const float* __restrict__ cen,
                                                          const int32_t* __restrict__ flag, const int32_t* __restrict__ slot,
                                                          const int32_t* __restrict__ total, float* __restrict__ out,
                                                          int32_t* __restrict__ count_out)
{
    const int st = gp->status;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { count_out[0] = st ? 0 : *total; count_out[1] = st; }
    if (st || i >= n || !flag[i]) return;
    const size_t o = (size_t)slot[i] * 3;
    out[o] = cen[(size_t)i * 3]; out[o + 1] = cen[(size_t)i * 3 + 1]; out[o + 2] = cen[(size_t)i * 3 + 2];
}

// ---------------------------------------------------------------------------------------------- random permutation
// A permutation of [0, n) computed element by element, no sort and no host RNG: a 4-round Feistel network over 2w bits
// (2^(2w) >= n, w = ceil(log2(n) / 2)) keyed by mix64(seed, round, half) is a bijection of [0, 2^(2w)); walking its cycle until the
// value falls below n ("cycle walking") restricts it to a bijection of [0, n).  2^(2w) < 4n, so a lane walks < 4 steps on average.
// It stands in for np.random.choice(N, N, replace=False) (models/patch_embedder.py:96) / np.random.shuffle
// (dataset/threedmatch.py:99,109) when the reference's exact legacy-NumPy stream is not required: any permutation gives "the
// first P neighbours in a random order"; the host draws cost ~20 ms per pair, this costs microseconds.
__host__ __device__ inline uint32_t pre_feistel(uint32_t x, int w, uint64_t seed)
{
    const uint32_t mask = (1u << w) - 1u;
    uint32_t L = x >> w, R = x & mask;
    for (int r = 0; r < 4; ++r) {
        const uint32_t f = (uint32_t)bx_mix64(seed, ((uint64_t)r << 32) | R) & mask;
        const uint32_t nl = R;
        R = L ^ f;
        L = nl;
    }
    return (L << w) | R;
}

__global__ __launch_bounds__(256) void random_perm_kernel(int n, int w, uint64_t seed, int32_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = pre_feistel((uint32_t)i, w, seed);
    while (x >= (uint32_t)n) x = pre_feistel(x, w, seed);
    out[i] = (int32_t)x;
}

// ---------------------------------------------------------------------------------------------- PCA
constexpr int PCA_BLOCKS = 64;

__global__ __launch_bounds__(256) void pca_partial_kernel(const float* __restrict__ pts, const int32_t* __restrict__ idx, int ns,
                                                          double* __restrict__ part /*[PCA_BLOCKS][9]*/)
{
    __shared__ double red[4][9];
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // sum x,y,z, xx,xy,xz,yy,yz,zz
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ns; j += gridDim.x * blockDim.x) {
        const size_t k = (size_t)idx[j] * 3;
        const double x = (double)pts[k], y = (double)pts[k + 1], z = (double)pts[k + 2];
        a[0] += x; a[1] += y; a[2] += z;
        a[3] += x * x; a[4] += x * y; a[5] += x * z; a[6] += y * y; a[7] += y * z; a[8] += z * z;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = bx_wave_sum(a[q]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 9; ++q) red[threadIdx.x >> 6][q] = a[q];
    __syncthreads();
    if (threadIdx.x < 9) part[blockIdx.x * 9 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one thread: covariance as scikit-learn forms it (X^T X - n mean mean^T) / (n - 1), Jacobi, descending order, v-based sign flip
__global__ void pca_final_kernel(const double* __restrict__ part, int ns, double* __restrict__ out /*[17]*/)
{
    if (threadIdx.x != 0) return;
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < PCA_BLOCKS; ++b)
        for (int q = 0; q < 9; ++q) a[q] += part[b * 9 + q];
    const double n = (double)ns;
    const double m[3] = {a[0] / n, a[1] / n, a[2] / n};
    double C[9];
    const double xx[6] = {a[3], a[4], a[5], a[6], a[7], a[8]};
    const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    for (int q = 0; q < 6; ++q) {
        const int i = ij[q][0], j = ij[q][1];
        const double v = (xx[q] - n * m[i] * m[j]) / (n - 1.0);
        C[i * 3 + j] = v; C[j * 3 + i] = v;
    }
    double V[9], w[3];
    bxd_jacobi3(C, V, w);
    int o[3] = {0, 1, 2};
    for (int p = 0; p < 2; ++p)
        for (int q = 0; q < 2 - p; ++q)
            if (w[o[q]] < w[o[q + 1]]) { const int t = o[q]; o[q] = o[q + 1]; o[q + 1] = t; }
    for (int r = 0; r < 3; ++r) {
        double wv = w[o[r]];
        out[r] = wv < 0.0 ? 0.0 : wv;
        double c0 = V[0 * 3 + o[r]], c1 = V[1 * 3 + o[r]], c2 = V[2 * 3 + o[r]];
        double mx = c0;
        if (fabs(c1) > fabs(mx)) mx = c1;
        if (fabs(c2) > fabs(mx)) mx = c2;
        const double sg = mx < 0.0 ? -1.0 : 1.0;
        out[3 + r * 3] = c0 * sg; out[3 + r * 3 + 1] = c1 * sg; out[3 + r * 3 + 2] = c2 * sg;
    }
    out[12] = m[0]; out[13] = m[1]; out[14] = m[2];
}

__global__ __launch_bounds__(256) void pca_zrange_kernel(const float* __restrict__ pts, int n, const double* __restrict__ pca, long long* zr)
{
    __shared__ double rlo[4], rhi[4];
    const double mx = pca[12], my = pca[13], mz = pca[14], cx = pca[9], cy = pca[10], cz = pca[11];
    double lo = 1.0e300, hi = -1.0e300;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double t = (((double)pts[(size_t)i * 3] - mx) * cx + ((double)pts[(size_t)i * 3 + 1] - my) * cy) + ((double)pts[(size_t)i * 3 + 2] - mz) * cz;
        lo = t < lo ? t : lo;
        hi = t > hi ? t : hi;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const double ol = __shfl_xor(lo, s, 64), oh = __shfl_xor(hi, s, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) { rlo[threadIdx.x >> 6] = lo; rhi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = rlo[w] < lo ? rlo[w] : lo; hi = rhi[w] > hi ? rhi[w] : hi; }
        lo = rlo[0] < lo ? rlo[0] : lo; hi = rhi[0] > hi ? rhi[0] : hi;
        atomicMin(&zr[0], d2ord(lo));
        atomicMax(&zr[1], d2ord(hi));
    }
}

__global__ void pca_zfinish_kernel(const long long* zr, double* out)
{
    if (threadIdx.x == 0) { out[15] = ord2d(zr[0]); out[16] = ord2d(zr[1]); }
}

// exclusive scan of data[0..n) -> out (may alias), optional grand total; t1/t2: scratch for the tile sums
int scan_excl(hipStream_t s, const int32_t* data, long long n, int32_t* out, int32_t* total, int32_t* t1, int32_t* t2)
{
    const long long nt1 = (n + ST - 1) / ST;
    const long long nt2 = (nt1 + ST - 1) / ST;
    if (nt2 > 4096) { bx_set_error("scan_excl: %lld elements exceed the scan capacity", n); return BX_ERR_ARG; }
    hipLaunchKernelGGL(tile_sums_kernel, dim3((unsigned)nt1), dim3(256), 0, s, data, n, t1);
    hipLaunchKernelGGL(tile_sums_kernel, dim3((unsigned)nt2), dim3(256), 0, s, t1, nt1, t2);
    hipLaunchKernelGGL(small_scan_kernel, dim3(1), dim3(64), 0, s, t2, (int)nt2);
    hipLaunchKernelGGL(tile_scan_kernel, dim3((unsigned)nt2), dim3(256), 0, s, t1, nt1, t2, t1, (int32_t*)nullptr);
    hipLaunchKernelGGL(tile_scan_kernel, dim3((unsigned)nt1), dim3(256), 0, s, data, n, t1, out, total);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
}  // namespace

struct bx_pre_ws {
    int64_t max_points, slots;
    char* mem;
    unsigned long long* keys;
    int32_t *bbox, *cnt, *start, *t1, *t2, *cellid, *seg, *pos, *flag, *slot, *total;
    float* cen;
    PreGrid* grid;
    double* part;
    long long* zr;
};

static int64_t pre_slots(int64_t n)   // hash slots for n points: a power of two, load factor <= 0.5
{
    int64_t t = 1024;
    while (t < 2 * n) t *= 2;
    return t;
}

int bxk_pre_reserve(bx_ctx* c, int64_t max_points)
{
    if (max_points < 1 || max_points > ((int64_t)1 << 26)) {
        bx_set_error("bx_pre_reserve: max_points must be in [1, 2^26]");
        return BX_ERR_ARG;
    }
    bx_pre_ws* w = static_cast<bx_pre_ws*>(c->pre);
    if (w && w->max_points >= max_points) return BX_OK;
    if (w) { (void)hipFree(w->mem); delete w; c->pre = nullptr; }
    w = new bx_pre_ws();
    w->max_points = max_points;
    w->slots = pre_slots(max_points);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t NP = (size_t)max_points, NS = (size_t)w->slots;
    const size_t nt1 = NS / ST + 2, nt2 = nt1 / ST + 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t o_bbox = take(64), o_grid = take(sizeof(PreGrid)), o_keys = take(NS * 8), o_cnt = take((NS + ST) * 4),
                 o_start = take((NS + ST) * 4), o_t1 = take(nt1 * 4), o_t2 = take(nt2 * 4), o_cell = take(NP * 4), o_seg = take(NP * 4),
                 o_pos = take(NP * 4), o_flag = take((NP + ST) * 4), o_slot = take((NP + ST) * 4), o_total = take(64),
                 o_cen = take(NP * 12), o_part = take(PCA_BLOCKS * 9 * 8), o_zr = take(64);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&w->mem), off);
    if (e != hipSuccess) { bx_set_error("bx_pre_reserve: hipMalloc(%zu) failed: %s", off, hipGetErrorString(e)); delete w; return BX_ERR_HIP; }
    char* m = w->mem;
    w->bbox = (int32_t*)(m + o_bbox); w->grid = (PreGrid*)(m + o_grid); w->keys = (unsigned long long*)(m + o_keys);
    w->cnt = (int32_t*)(m + o_cnt); w->start = (int32_t*)(m + o_start);
    w->t1 = (int32_t*)(m + o_t1); w->t2 = (int32_t*)(m + o_t2); w->cellid = (int32_t*)(m + o_cell); w->seg = (int32_t*)(m + o_seg);
    w->pos = (int32_t*)(m + o_pos); w->flag = (int32_t*)(m + o_flag); w->slot = (int32_t*)(m + o_slot); w->total = (int32_t*)(m + o_total);
    w->cen = (float*)(m + o_cen); w->part = (double*)(m + o_part); w->zr = (long long*)(m + o_zr);
    c->pre = w;
    return BX_OK;
}

void bxk_pre_release(bx_ctx* c)
{
    bx_pre_ws* w = static_cast<bx_pre_ws*>(c->pre);
    if (w) { (void)hipFree(w->mem); delete w; c->pre = nullptr; }
}

int bxk_pre_voxel_downsample(bx_ctx* c, hipStream_t s, const float* pts, int n, double voxel_size, float* out, int32_t* count_out)
{
    bx_pre_ws* w = static_cast<bx_pre_ws*>(c->pre);
    if (!w) { bx_set_error("bx_pre_voxel_downsample: call bx_pre_reserve first"); return BX_ERR_STATE; }
    if (n < 1 || n > w->max_points || !(voxel_size > 0.0)) { bx_set_error("bx_pre_voxel_downsample: n=%d (reserved %lld) voxel=%g", n, (long long)w->max_points, voxel_size); return BX_ERR_ARG; }
    const int nb = (n + 255) / 256;
    const int64_t ts = pre_slots(n);   // this call's table: memset and scan stay proportional to n
    int rc;
    hipLaunchKernelGGL(pre_init_kernel, dim3(1), dim3(64), 0, s, w->bbox, (long long*)nullptr);
    hipLaunchKernelGGL(pre_bbox_kernel, dim3(nb < 1024 ? (nb + 3) / 4 : 256), dim3(1024), 0, s, pts, n, w->bbox);
    hipLaunchKernelGGL(pre_setup_kernel, dim3(1), dim3(64), 0, s, w->bbox, voxel_size, w->grid);
    BX_HIP(hipMemsetAsync(w->keys, 0xff, (size_t)ts * 8, s));
    BX_HIP(hipMemsetAsync(w->cnt, 0, ((size_t)ts + ST) * 4, s));
    BX_HIP(hipMemsetAsync(w->flag, 0, ((size_t)n + 1) * 4, s));
    hipLaunchKernelGGL(pre_count_kernel, dim3(nb), dim3(256), 0, s, pts, n, w->grid, w->keys, (unsigned)(ts - 1), w->cnt, w->cellid);
    if ((rc = scan_excl(s, w->cnt, ts, w->start, nullptr, w->t1, w->t2)) != BX_OK) return rc;
    hipLaunchKernelGGL(pre_scatter_kernel, dim3(nb), dim3(256), 0, s, n, w->grid, w->cellid, w->start, w->cnt, w->seg, w->pos);
    hipLaunchKernelGGL(pre_reduce_kernel, dim3(nb), dim3(256), 0, s, pts, n, (int)ts, w->grid, w->cellid, w->start, w->pos, w->seg, w->cen, w->flag);
    if ((rc = scan_excl(s, w->flag, n, w->slot, w->total, w->t1, w->t2)) != BX_OK) return rc;
    hipLaunchKernelGGL(pre_compact_kernel, dim3(nb), dim3(256), 0, s, n, w->grid, w->cen, w->flag, w->slot, w->total, out, count_out);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_pre_pca(bx_ctx* c, hipStream_t s, const float* pts, int n, const int32_t* sample_idx, int ns, double* out17)
{
    bx_pre_ws* w = static_cast<bx_pre_ws*>(c->pre);
    if (!w) { bx_set_error("bx_pre_pca: call bx_pre_reserve first"); return BX_ERR_STATE; }
    if (n < 1 || ns < 2) { bx_set_error("bx_pre_pca: n=%d ns=%d", n, ns); return BX_ERR_ARG; }
    hipLaunchKernelGGL(pre_init_kernel, dim3(1), dim3(64), 0, s, w->bbox, w->zr);
    hipLaunchKernelGGL(pca_partial_kernel, dim3(PCA_BLOCKS), dim3(256), 0, s, pts, sample_idx, ns, w->part);
    hipLaunchKernelGGL(pca_final_kernel, dim3(1), dim3(64), 0, s, w->part, ns, out17);
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(pca_zrange_kernel, dim3(nb < 512 ? nb : 512), dim3(256), 0, s, pts, n, out17, w->zr);
    hipLaunchKernelGGL(pca_zfinish_kernel, dim3(1), dim3(64), 0, s, w->zr, out17);
    BX_LAUNCH_CHECK();
    return BX_OK;
}

int bxk_random_perm(hipStream_t s, int n, uint64_t seed, int32_t* out)
{
    if (n < 1) return BX_OK;
    int w = 1;
    while (((int64_t)1 << (2 * w)) < (int64_t)n) ++w;
    hipLaunchKernelGGL(random_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, w, seed, out);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
