// bx_common.h -- internal declarations shared by the HIP translation units of libbufferx_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, f32 MFMA, 160 KiB LDS.  Build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/bufferx.h"

#define BX_PI_F 3.14159274101257324219f
#define BX_WAVE 64

void bx_set_error(const char* fmt, ...);
#define BX_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            bx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return BX_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)
#define BX_LAUNCH_CHECK() BX_HIP(hipGetLastError())

// Every entry point that launches work or allocates makes its device current for the calling thread and restores the previous
// one on return: contexts on several devices may be driven from one thread (or one device from nn.DataParallel's replica
// threads, whose current device is not ours).
struct BxDevScope {
    int prev = -1; bool switched = false;
    explicit BxDevScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~BxDevScope() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    BxDevScope(const BxDevScope&) = delete;
    BxDevScope& operator=(const BxDevScope&) = delete;
};

// ------------------------------------------------------------------ geometry constants
constexpr int BX_RAD = 3, BX_ELE = 7, BX_AZI = 20;
constexpr int BX_EA = BX_ELE * BX_AZI;          // 140 positions of the cylindrical map
constexpr int BX_VOX = BX_RAD * BX_EA;          // 420 voxels
constexpr int BX_NDESC = 8, BX_NPOSE = 10;

struct ConvLayerDev {
    float* W;        // B fragments of the 16x16x4 kernels: [chunk*taps][column tile 16][lane][4]
    float* Wwino;    // B fragments of the Winograd kernels (k_wino.hip): U = G g G^T, [chunk*16 + plane][column tile 16][lane][4]; Desc layers only
    float* Wwino43;  // MFMA fragments of the F(4x4, 3x3) kernels (k_wino43.hip): [chunk*36 + plane][column tile 16][lane][4], columns in output-slot order; desc_conv_form winograd43
    float* b;        // [cout]
    int32_t* lrow;   // [p_in]  LDS row of an input position inside a unit's p_lds-row slab
    int32_t* lrow2;  // [p_in]  second copy (azimuth wrap halo) or -1
    int32_t* obase;  // [p_out] LDS row of the window origin of an output position
    int32_t* toff;   // [ntaps] row offset of a tap
    int nchunk, ntaps, p_in, p_lds, p_out, cout, relu;
};

// device-side per-pair state written/read by the pipeline kernels (no host round trips)
struct PairState {
    int32_t m_scale;        // mutual matches of the current scale
    int32_t M;              // accumulated matches
    int32_t C;              // |inlier_ind|
    int32_t best;           // best hypothesis
    int32_t done;           // early exit taken
    int32_t scales_used;
    int32_t num_inliers;
    int32_t ransac_iters;
    int32_t refine_iters;
    int32_t status;
    int32_t pad[2];
    double des_r[BX_MAX_SCALES];
    double T[16];           // RANSAC pose
    float Tf[16];           // refined pose
    // ransac scan state
    int32_t r_best_inl;
    int32_t r_est_k;
    double r_best_rmse;
    int32_t r_itr;
    int32_t r_pad;
};

// uniform grid of the neighbour gather (k_ball.hip), rebuilt per launch on the device
constexpr int BX_BALL_NCELL = 1 << 17;
struct BallGrid {
    float ox, oy, oz, inv_h, rpad;
    int32_t dx, dy, dz, ncells;
};
// tuning knob (env BX_BALL_DIV, read once): cell edge = padded radius / div  (1..3)
int bx_ball_div();
void bx_prof_mark(bx_ctx* c, hipStream_t s, int tag, int begin);   // hipEvent bracket (bx_profile_*), no-op unless enabled

// One lane per GPU: the convolution stacks of the pairs in flight on different contexts / streams run one after the other
// (in submission order) instead of interleaving workgroup by workgroup, while everything latency-bound (FPS, RANSAC, the small
// kernels) still overlaps with them.  Events are recorded round-robin from a ring; a wait captures the latest record.
struct bx_lane {
    static constexpr int NEV = 64;
    hipEvent_t ev[NEV];
    int next;      // ring position of the next record
    int last;      // ring position of the latest record, -1: none yet
    int mode;      // 1: whole main phase (everything after FPS), 2: conv stacks only
};

struct bx_ctx {
    int device;
    bx_params p;
    bool weights_loaded;
    // constants
    float* d_centres;   // [420][3]
    float* d_rot;       // [20][4]
    float* d_rowc;      // [21][2] circle {radius, height} of every (shell, elevation) row of voxel centres
    int patch_attr_set;
    float* d_rad_thr;   // [8193] radius-estimation thresholds (float)(r_m^2)
    float *d_pnt_w, *d_pnt_b, *d_pool_w1, *d_pool_b1, *d_pool_w2, *d_pool_b2;
    ConvLayerDev desc[BX_NDESC];
    ConvLayerDev pose[BX_NPOSE];
    // workspace arena
    char* arena;
    int64_t arena_bytes;
    // carved buffers (see bx_api.hip)
    float *act0, *act1;                 // conv ping-pong
    float* patches;                     // [K][P][3]
    int32_t *pcnt, *pcnt2;              // [K] real slots per patch (hit-count hand-over; pcnt2: the target chain's scratch in the latency form)
    int32_t* ball_idx;                  // [K][P] ball_query indices (the reference op's first output; kept for parity of the op)
    float* feat;                        // [K][3][140][16]
    float* pts_perm;                    // [max_points][3]
    int32_t* fps_idx[2];                // per cloud [max(K,nk)]
    float* kpts[2];                     // [K][3]
    float* kpts_r[2];                   // radius-estimation keypoints [nk][3]
    float* desc_out[2];                 // [K][32]
    float* equi[2];                     // [K][140][32]
    float* Rpatch[2];                   // [K][9]
    // per-scale views of the three arrays above: all scales share one buffer unless the keypoint tiles of the latency mode are on
    // (params.keypoint_tiles > 1: the descriptors of every scale are then produced tile by tile before any scale is matched)
    float *desc_sc[BX_MAX_SCALES][2], *equi_sc[BX_MAX_SCALES][2], *R_sc[BX_MAX_SCALES][2];
    float* fps_td[2];                   // [max_points] running min-distances between the launches of a tiled FPS run (else null)
    hipStream_t aux_stream;             // latency mode: the FPS launches run here, beside the descriptor work of the caller's stream
    hipStream_t tgt_stream;             // latency mode: the target cloud's descriptor chain (the source cloud's stays on the caller's)
    hipStream_t match_stream;           // latency mode without early exit: matching + CostNet of scale i beside the descriptor work of i+1..
    hipEvent_t ev_fork, ev_tile[BX_MAX_TILES], ev_tgt_go, ev_tgt_done, ev_desc[2][BX_MAX_SCALES], ev_match_done;
    float *patches2, *feat2, *act2[2];  // latency mode: scratch of the target cloud's chain
    float* act3[2];                     // latency mode: conv ping-pong of the source cloud's chain (act0/act1 stay with the CostNet)
    unsigned long long* nn_key[2];      // [K]
    int32_t *s_mids, *t_mids;           // [K]
    float* ind;                         // [K]
    float *R_cat, *t_cat, *ss_cat, *tt_cat;  // [S*K][..]
    int32_t* cons_cnt;                  // [S*K]
    float* cons_thr;                    // [S*K]
    int32_t* inlier_ind;                // [S*K]
    unsigned long long* rad_hist;       // [8200]
    float* fps_dist;                    // [2][max_points]  (unused by register path; kept for generic path)
    unsigned long long* fps_slots;      // cross-workgroup exchange granules
    unsigned long long* fps_hello;      // [2][64] placement handshake granules (k_fps.hip)
    int32_t* fps_ord;                   // [2][max_points] spatial order of the two clouds (bucket pruning, k_fps.hip)
    unsigned short* fps_cell;           // [2][max_points] Morton cell of every point
    int* fps_cnt;                       // [2][4096] cell counters, then fps_bbmax [8]
    unsigned *fps_bbmin, *fps_bbmax;    // [2][4] encoded bounding boxes
    int fps_attr_set;
    int fps_xcd_pair;                   // 0..3: the XCD pair {2p, 2p+1} the co-located FPS launches of this context aim at; -1 until its first FPS launch
    int32_t* ransac_inl;                // [RANSAC_BATCH]
    double* ransac_err;                 // [RANSAC_BATCH]
    double* ransac_T;                   // [RANSAC_BATCH][12]
    float* refine_ws;                   // [8][S*K]
    int32_t* refine_sel;                // [S*K]
    float* sub_pts;                     // [200000][3] subsample buffer for radius estimation
    // neighbour-gather grid (k_ball.hip)
    // ... one "set" per (cloud, scale) of a pair: ball_nsets = 2 * num_scales; element j of a per-set array at base + j * stride
    int ball_nsets;
    size_t ball_st_cnt, ball_st_bsum, ball_st_pts, ball_st_tab, ball_st_num;
    float* ball_bbox_part;              // [2][64][6] per-block bounds of the two clouds
    BallGrid* ball_grid;                // [nsets]
    int32_t *ball_cnt, *ball_start;     // [nsets][BX_BALL_NCELL + 2 tiles]; cnt is all zeros between launches
    int32_t* ball_bsum;                 // [nsets] per scan tile
    int2* ball_cellrank;                // [nsets][max_points]
    int2* ball_ptab;                    // [nsets][num_fps][256] per-keypoint candidate pieces {first slot, count} (ball_rows_kernel)
    int32_t* ball_pnum;                 // [nsets][num_fps] pieces per keypoint, -1: degenerate geometry
    int ball_logpw[2 * BX_MAX_SCALES];  // piece width (log2) of every prepared set
    float4 *ball_pts4, *ball_sorted;    // [nsets][max_points] {x,y,z,0} in permuted order / {x,y,z,bits(i)} sorted by cell
    long long ball_attr_set;
    int ball_waves_hint;                // waves per keypoint of the next neighbour-gather launch (0 = default 2)
    long long* ball_dbg;                // [64][8] cycle stamps (BX_BALL_DEBUG)
    PairState* state;                   // device
    bx_result* result_dev;              // device staging of the result
    int32_t* err_flag;                  // device error flag
    const int32_t* skip;                // device flag: non-zero => pipeline kernels return immediately (early exit)
    struct bx_lane* lane;               // optional: orders the MFMA-heavy sections of the pairs of several contexts (bx_attach_lane)
    char* kiss_ws;                      // KISS-Matcher workspace (k_kiss.hip), allocated when params.pose_estimator == 1
    int kiss_max_C;
    void* pre;                          // bx_pre_ws* (k_pre.hip): workspace of the pre-processing entry points, reserved on demand
    int conv_cap[2][BX_NPOSE];          // persistent-grid size of every conv layer on THIS device (0 = not set up yet)
    int wino_cap[BX_NDESC];             // the same for the Winograd kernels (k_wino.hip)
    int wino43v_cap[BX_NPOSE];          // persistent grid of the valid F(4x4) kernels (k_wino43v.hip)
    int wino_pose_cap[BX_NPOSE];
    int use_wino_pose;                  // bx_params.pose_conv_form: 1 = winograd (valid F(2x2, 3x3), k_wino.hip), 2 = winograd43 (valid F(4x4, 3x3), k_wino43v.hip), 0 = direct -- CostNet layers 1..5
    int use_wino;                       // bx_params.desc_conv_form: 3 = winograd43m (mixed F(4x4) / F(3x4) tiles, every layer; fragments in Wwino43), 2 = winograd43 (F(4x4, 3x3), every layer), 1 = winograd22 (F(2x2, 3x3), layers with >= 64 output channels), 0 = direct
    int conv_persist, conv_cap_override, n_cu;
    int desc_batch;                     // 1 (default): both clouds of a scale in one Cylindrical_Net stack (bx_api.hip::desc_stack_pair); hook BX_DESC_BATCH
    int rad_slices;                     // measurement hook BX_RAD_SLICES: point slices of radius_hist_kernel (0 = the default, 16)
    // bx_register_pair_begin / _finish: the arguments of the pending pair (pointers remembered, not copied) and what the first call consumed
    struct PendingPair {
        bool active;
        const float *src, *tgt;
        int32_t n_src, n_tgt, aligned_z;
        const int32_t *perm_src, *perm_tgt;
        uint64_t seed;
        int ransac_calls;
    } pend;
    double *d_cost_wp, *d_cost_wq;      // collapsed CostNet layer 0 (k_cost.hip): binary64 weights of the P / Q convolutions
    int cost_direct;                    // bx_params.cost_l0_form == direct: layer 0 as the fp32 MFMA convolution of the implicit volume (cost_l1_kernel)
    int32_t* conv_ctr;                  // [2 * BX_NDESC] {next group ticket, departed workgroups} of the 32x32x2 kernels' group walk
    bx_capture cap;                     // bx_set_capture: intermediates of one scale copied to caller buffers
    int cap_on;
    int prof_on;
    void* prof;                         // std::vector<ProfEvt>* (bx_api.hip)
};

constexpr int BX_RANSAC_BATCH = 4096;

// ------------------------------------------------------------------ kernel launchers (one per .hip file)
int bxk_fps(bx_ctx* c, hipStream_t s, const float* const* xyz, const int* n, int nclouds, int m, int32_t* const* idx_out,
            float* const* kpts_out);
int bx_live_contexts(int device);   // contexts alive on the device in this process (bx_api.hip)
int bx_xcd_slot_take(int device);                // least-used XCD pair of the device; released by bx_destroy
int bx_xcd_pair_sharing(int device, int pair);   // live contexts of the device whose co-located FPS launches aim at this XCD pair
int bxk_fps_range(bx_ctx* c, hipStream_t s, const float* const* xyz, const int* n, int nclouds, int j0, int j1, int m,
                  int32_t* const* idx_out, float* const* kpts_out);
int bxk_gather_rows(hipStream_t s, const float* pts, const int32_t* idx, int n, float* out);
int bxk_radius_hist(bx_ctx* c, hipStream_t s, const float* pts, int n_pts, const float* kpts, int nk);
int bxk_radius_bisect(bx_ctx* c, hipStream_t s, int64_t n_orig, int nk, double threshold, double* des_r_out);
int bxk_ball_group(bx_ctx* c, hipStream_t s, const float* pts_perm, int n, const float* kpts, int K, const double* radius,
                   int P, int32_t* idx_out, float* patches_out, int32_t* cnt_out = nullptr);
int bxk_ball_prepare(bx_ctx* c, hipStream_t s, const float* const* clouds, const int* ns, const int32_t* const* perms,
                     const float* const* kpts, int nclouds, int K, const double* radius, int S, const double* pw_hint);
int bxk_ball_grids(bx_ctx* c, hipStream_t s, const float* const* clouds, const int* ns, const int32_t* const* perms, int nclouds,
                   const double* radius, int S, const double* pw_hint, int i0 = 0, int ni = -1);      // scales [i0, i0 + ni) (ni < 0: to S); honours c->skip
int bxk_ball_rows(bx_ctx* c, hipStream_t s, const float* const* kpts, int nclouds, int S, int k0, int K, int i0 = 0, int ni = -1);
int bxk_ball_query(bx_ctx* c, hipStream_t s, int set, int n, const float* kpts, int k0, int K, const double* radius, int P,
                   int32_t* idx_out, float* patches_out, int32_t* cnt_out = nullptr);   // cnt_out: hit-count hand-over (only the real slots are written)
int bxk_radius_bisect_all(bx_ctx* c, hipStream_t s, int64_t n_orig, int nk, const double* thresholds_host, int nthr, double* des_r_out);
int bxk_patch_features(bx_ctx* c, hipStream_t s, const float* patches, int K, int P, const double* radius, int aligned,
                       float* R_out, float* feat_out, const int32_t* cnt = nullptr, const float* kpts = nullptr);   // cnt + kpts: counted patches (hit-count hand-over)
int bxk_conv(bx_ctx* c, hipStream_t s, int net, int layer, const float* in, const int32_t* units_dev, int max_units,
             float* out);
int bxk_wino43_weights(const float* w, int nchunk, int fold, int cout, float** d_out);
int bxk_wino43v(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out);
int bxk_wino43(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out);
int bxk_wino43m(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out);   // mixed tiles (k_wino43m.hip)
int bxk_wino43m_weights(const float* w, int nchunk, int cout, float** d_out);
int bxk_wino(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out);
int bxk_wino_weights(const float* w, int nchunk, int fold, int cout, float** d_out);
int bxk_wino_pose(bx_ctx* c, hipStream_t s, int layer, const float* in, const int32_t* units_dev, int max_units, float* out);
int bxk_cost_l1(bx_ctx* c, hipStream_t s, const float* s_equi, const float* t_equi, const int32_t* s_mids,
                const int32_t* t_mids, const int32_t* m_dev, int max_m, float* out);
int bxk_cost_l0_weights(const float* w0, double** d_wp, double** d_wq);
int bxk_cost_l0(bx_ctx* c, hipStream_t s, const float* s_equi, const float* t_equi, const int32_t* s_mids, const int32_t* t_mids,
                const int32_t* m_dev, int max_m, float* out);
int bxk_desc_head(bx_ctx* c, hipStream_t s, const float* x, int K, float* desc, float* equi);
int bxk_mutual(bx_ctx* c, hipStream_t s, const float* sd, int ns, const float* td, int nt, int32_t* s_mids, int32_t* t_mids,
               int32_t* count_out);
int bxk_soft_argmax(hipStream_t s, const float* logits, const int32_t* m_dev, int max_m, float* ind, const int32_t* skip);
int bxk_hypotheses(hipStream_t s, const float* ind, const int32_t* s_mids, const int32_t* t_mids, const int32_t* m_dev,
                   int max_m, const float* s_R, const float* t_R, const float* s_k, const float* t_k, float* R_out,
                   float* t_out, float* ss_out, float* tt_out, const int32_t* base_dev, const int32_t* skip);
int bx_permute_launch(hipStream_t s, const float* pts, const int32_t* perm, int n, float* out, const int32_t* skip);
int bxk_consensus(bx_ctx* c, hipStream_t s, const float* R, const float* t, const float* ss, const float* tt,
                  const int32_t* M_dev, int max_M, int32_t* inlier_out, int32_t* count_out, int32_t* best_out);
int bxk_ransac(bx_ctx* c, hipStream_t s, const float* ss, const float* tt, const int32_t* corr, const int32_t* C_dev,
               int max_C, uint64_t seed, double* T_out, int32_t* info_out, const int32_t* skip_flag);
size_t bxk_kiss_workspace_bytes(int max_C);
int bxk_kiss(bx_ctx* c, hipStream_t s, const float* ss, const float* tt, const int32_t* corr, const int32_t* C_dev, int max_C,
             double* T_out, int32_t* info_out, const int32_t* skip_flag);
int bxk_pre_reserve(bx_ctx* c, int64_t max_points);
void bxk_pre_release(bx_ctx* c);
int bxk_pre_voxel_downsample(bx_ctx* c, hipStream_t s, const float* pts, int n, double voxel_size, float* out, int32_t* count_out);
int bxk_random_perm(hipStream_t s, int n, uint64_t seed, int32_t* out);
int bxk_pre_pca(bx_ctx* c, hipStream_t s, const float* pts, int n, const int32_t* sample_idx, int ns, double* out17);
int bxk_refine(bx_ctx* c, hipStream_t s, const float* ss, const float* tt, const int32_t* M_dev, int max_M, float* T_io,
               int32_t* iters_out);

#ifdef __HIPCC__
// ------------------------------------------------------------------ device helpers
// Streamed-once activations (a layer's input map, its output map) bypass the L2 retention policy: with plain loads / stores the ~360 MB
// a layer moves evict the few MB of weight fragments every workgroup re-reads, and the MFMA loops wait for them at Infinity-Cache
// latency (round 3: -12 % per convolution kernel, measured on k_wino43.hip first).
__device__ __forceinline__ float4 bx_ld_stream(const float4* p)
{
    typedef float bx_f32x4 __attribute__((ext_vector_type(4)));
    const bx_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bx_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void bx_st_stream(float4* p, const float4& y)
{
    typedef float bx_f32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((bx_f32x4){y.x, y.y, y.z, y.w}, reinterpret_cast<bx_f32x4*>(p));
}
__host__ __device__ __forceinline__ uint64_t bx_mix64(uint64_t seed, uint64_t ctr)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (ctr + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// xor-butterfly all-reduce sums over the 64 lanes ("wave order" of the arithmetic contract)
__device__ __forceinline__ float bx_wave_sum(float v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
    return v;
}
__device__ __forceinline__ double bx_wave_sum(double v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
    return v;
}
__device__ __forceinline__ unsigned long long bx_wave_max(unsigned long long v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        unsigned long long o = __shfl_xor(v, s, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int bx_wave_sum_i(int v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes on the DPP network (no LDS round trips): Hillis-Steele inside each row of
// 16 lanes (row_shr 1,2,4,8), then row_bcast15 / row_bcast31 carry the row totals (gfx9 DPP broadcasts)
__device__ __forceinline__ int bx_wave_incl_scan_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// exclusive prefix sum over the wave
__device__ __forceinline__ int bx_wave_excl_scan(int v, int lane)
{
    int x = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        int y = __shfl_up(x, s, 64);
        if (lane >= s) x += y;
    }
    return x - v;
}

// ---- deterministic binary64 elementary functions (same series as oracle/bxo_detmath.h, written for the device)
__device__ __forceinline__ double bxd_pow2i(int k) { return __longlong_as_double((long long)(k + 1023) << 52); }

__device__ inline double bxd_exp(double x)
{
    if (x < -700.0) return 0.0;
    if (x > 700.0) x = 700.0;
    double kf = floor(x * 1.4426950408889634074 + 0.5);
    double r = (x - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p * bxd_pow2i((int)kf);
}

__device__ inline void bxd_sincos(double x, double* s, double* c)
{
    double kf = floor(x * 0.63661977236758134308 + 0.5);
    double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
    r = r - kf * 2.02226624879595063154e-21;
    double r2 = r * r;
    double ps = -1.0 / 121645100408832000.0;
    ps = ps * r2 + 1.0 / 355687428096000.0;
    ps = ps * r2 - 1.0 / 1307674368000.0;
    ps = ps * r2 + 1.0 / 6227020800.0;
    ps = ps * r2 - 1.0 / 39916800.0;
    ps = ps * r2 + 1.0 / 362880.0;
    ps = ps * r2 - 1.0 / 5040.0;
    ps = ps * r2 + 1.0 / 120.0;
    ps = ps * r2 - 1.0 / 6.0;
    ps = ps * r2 + 1.0;
    double sn = ps * r;
    double pc = -1.0 / 6402373705728000.0;
    pc = pc * r2 + 1.0 / 20922789888000.0;
    pc = pc * r2 - 1.0 / 87178291200.0;
    pc = pc * r2 + 1.0 / 479001600.0;
    pc = pc * r2 - 1.0 / 3628800.0;
    pc = pc * r2 + 1.0 / 40320.0;
    pc = pc * r2 - 1.0 / 720.0;
    pc = pc * r2 + 1.0 / 24.0;
    pc = pc * r2 - 0.5;
    pc = pc * r2 + 1.0;
    double cs = pc;
    long long k = (long long)kf;
    int q = (int)(k & 3);
    if (q == 0) { *s = sn; *c = cs; }
    else if (q == 1) { *s = cs; *c = -sn; }
    else if (q == 2) { *s = -sn; *c = -cs; }
    else { *s = -cs; *c = sn; }
}

__device__ inline double bxd_asin_small(double t)
{
    double t2 = t * t;
    double term = t;
    double sum = t;
    for (int n = 1; n <= 30; ++n) {
        term = term * t2 * ((double)(2 * n - 1) / (double)(2 * n));
        sum = sum + term / (double)(2 * n + 1);
    }
    return sum;
}

__device__ inline double bxd_acos(double x)
{
    const double PI = 3.14159265358979323846;
    if (x > 1.0) x = 1.0;
    if (x < -1.0) x = -1.0;
    double ax = x < 0 ? -x : x;
    if (ax <= 0.5) return PI * 0.5 - bxd_asin_small(x);
    double z = (1.0 - ax) * 0.5;
    double a = 2.0 * bxd_asin_small(sqrt(z));
    return x > 0 ? a : PI - a;
}

__device__ inline double bxd_log(double x)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
    double f = (m - 1.0) / (m + 1.0);
    double f2 = f * f;
    double sum = 0.0;
    for (int n = 14; n >= 0; --n) sum = sum * f2 + 1.0 / (double)(2 * n + 1);
    double lm = 2.0 * f * sum;
    return ((double)e * 6.93147180369123816490e-01 + lm) + (double)e * 1.90821492927058770002e-10;
}

// symmetric 3x3 Jacobi (binary64, 10 fixed sweeps); v = eigenvectors as columns
__device__ inline void bxd_jacobi3(double a[9], double v[9], double w[3])
{
    for (int i = 0; i < 9; ++i) v[i] = 0.0;
    v[0] = v[4] = v[8] = 1.0;
    for (int sweep = 0; sweep < 10; ++sweep) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = (r == 2) ? 1 : 0;
            const int q = (r == 0) ? 1 : 2;
            double apq = a[p * 3 + q];
            if (apq == 0.0) continue;
            double app = a[p * 3 + p], aqq = a[q * 3 + q];
            double theta = (aqq - app) / (2.0 * apq);
            double at = theta < 0 ? -theta : theta;
            double t = 1.0 / (at + sqrt(theta * theta + 1.0));
            if (theta < 0) t = -t;
            double c = 1.0 / sqrt(t * t + 1.0);
            double s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double akp = a[k * 3 + p], akq = a[k * 3 + q];
                a[k * 3 + p] = c * akp - s * akq;
                a[k * 3 + q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                a[p * 3 + k] = c * apk - s * aqk;
                a[q * 3 + k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
                v[k * 3 + p] = c * vkp - s * vkq;
                v[k * 3 + q] = s * vkp + c * vkq;
            }
        }
    }
    w[0] = a[0]; w[1] = a[4]; w[2] = a[8];
}

// Kabsch rotation from H = sum a b^T (a source, b target); returns 0 on rank < 2.
__device__ inline int bxd_kabsch_from_H(const double H[9], double R[9])
{
    double hth[9], V[9], w[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            hth[i * 3 + j] = (H[0 * 3 + i] * H[0 * 3 + j] + H[1 * 3 + i] * H[1 * 3 + j]) + H[2 * 3 + i] * H[2 * 3 + j];
    bxd_jacobi3(hth, V, w);
    int o0 = 0, o1 = 1, o2 = 2, tmp;
    if (w[o1] > w[o0]) { tmp = o0; o0 = o1; o1 = tmp; }
    if (w[o2] > w[o0]) { tmp = o0; o0 = o2; o2 = tmp; }
    if (w[o2] > w[o1]) { tmp = o1; o1 = o2; o2 = tmp; }
    (void)o2;
    double v1[3] = {V[0 * 3 + o0], V[1 * 3 + o0], V[2 * 3 + o0]};
    double v2[3] = {V[0 * 3 + o1], V[1 * 3 + o1], V[2 * 3 + o1]};
    double l1 = w[o0], l2 = w[o1];
    if (!(l1 > 0.0) || !(l2 > l1 * 1e-24)) return 0;
    double u1[3], u2[3];
    for (int i = 0; i < 3; ++i) {
        u1[i] = (H[i * 3 + 0] * v1[0] + H[i * 3 + 1] * v1[1]) + H[i * 3 + 2] * v1[2];
        u2[i] = (H[i * 3 + 0] * v2[0] + H[i * 3 + 1] * v2[1]) + H[i * 3 + 2] * v2[2];
    }
    double n1 = sqrt((u1[0] * u1[0] + u1[1] * u1[1]) + u1[2] * u1[2]);
    if (!(n1 > 0.0)) return 0;
    for (int i = 0; i < 3; ++i) u1[i] = u1[i] / n1;
    double d12 = (u1[0] * u2[0] + u1[1] * u2[1]) + u1[2] * u2[2];
    for (int i = 0; i < 3; ++i) u2[i] = u2[i] - d12 * u1[i];
    double n2 = sqrt((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]);
    if (!(n2 > 0.0)) return 0;
    for (int i = 0; i < 3; ++i) u2[i] = u2[i] / n2;
    double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = (v1[i] * u1[j] + v2[i] * u2[j]) + v3[i] * u3[j];
    return 1;
}
#endif  // __HIPCC__
