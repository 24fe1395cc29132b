/*
 * include/bufferx.h -- C-ABI of libbufferx_hip.so: the MI355X-native BUFFER-X inference hot path.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain
 * pointers / sizes (no torch types), returns an int status (0 = BX_OK) and never throws or aborts.
 * Device pointers are caller-owned HIP device allocations (e.g. torch tensor .data_ptr()); `stream`
 * is a hipStream_t passed as void* (NULL = default stream).  All work is stream-ordered; no entry
 * point synchronises the device except bx_create/bx_destroy/bx_load_weights and the *_host helpers.
 *
 * Each entry point replaces an operator the reference reaches through un-vendored CUDA packages or
 * torch (reference file:line in the comment above it).  The reference binds those through Python
 * C-extensions; the binding a maintainer would add is the ctypes stub in buffer-x_amd/lib.py
 * (see INTEGRATION.md).
 *
 * Arithmetic contract: see oracle/bx_oracle.c header -- fp32 (fp64 for RANSAC), no implicit FMA
 * contraction, documented accumulation orders, deterministic transcendentals.  Randomness that the
 * reference leaves to unseeded global RNGs (models/patch_embedder.py:96 permutation, Open3D RANSAC
 * sampling, models/BUFFERX.py:665 subsample) is explicit here: permutations and a 64-bit seed are inputs.
 *
 * Layouts:
 *   clouds / keypoints          float32 [N][3]
 *   patches                     float32 [K][P][3]
 *   chunked feature maps        float32 [unit][C/16][pos][16]  (channel c of chunk at slot 4*(c%4)+c/4;
 *                               bx_chunk_slot() below -- the order MFMA f32 16x16x4 consumes 4 k-steps
 *                               from one 128-bit LDS read)
 *   descriptors                 float32 [K][32];   equivariant maps float32 [K][ele*azi][32]
 *   rotations                   float32 [.][9] row-major;  poses double/float [16] row-major 4x4
 */
#ifndef BUFFERX_H
#define BUFFERX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BX_OK 0
#define BX_ERR_ARG 1       /* bad argument */
#define BX_ERR_HIP 2       /* HIP runtime error (bx_last_error has the string) */
#define BX_ERR_STATE 3     /* weights not loaded, workspace too small, ... */
#define BX_ERR_DEVICE 4    /* device-side failure flag (spin timeout, ...) */

#define BX_MAX_SCALES 8
#define BX_MAX_TILES 8

typedef struct bx_ctx bx_ctx;
typedef struct bx_lane bx_lane;
typedef struct bx_prefetch bx_prefetch;

/* Hot-path knobs == reference cfg.patch / cfg.match / cfg.test fields (config/indoor_config.py:49-80,
 * config/outdoor_config.py:49-82; CLI overrides utils/test_args.py:46-63). */
typedef struct bx_params {
    int32_t num_fps;                      /* cfg.patch.num_fps */
    int32_t num_points_per_patch;         /* cfg.patch.num_points_per_patch */
    int32_t num_scales;                   /* cfg.patch.num_scales */
    int32_t rad_n, azi_n, ele_n;          /* 3, 20, 7 (only these are supported by the conv kernels) */
    int32_t voxel_sample;                 /* 10 */
    int32_t num_points_radius_estimate;   /* 2000 */
    double delta;                         /* 0.8 */
    double search_radius_thresholds[BX_MAX_SCALES];
    double dist_th, inlier_th, similar_th; /* cfg.match.* (Python floats in the reference => binary64 here) */
    double confidence;                    /* cfg.match.confidence */
    int32_t iter_n;                       /* cfg.match.iter_n */
    int32_t enable_early_exit;            /* cfg.match.enable_early_exit */
    int32_t early_exit_min_inliers;       /* cfg.match.early_exit_min_inliers */
    int32_t pose_refine;                  /* cfg.test.pose_refine */
    int32_t max_points;                   /* workspace sizing: largest cloud this context will see */
    int32_t pose_estimator;               /* cfg.match.pose_estimator: 0 = "ransac", 1 = "kiss_matcher" (utils/test_args.py:74-80) */
    double kiss_resolution;               /* cfg.match.kiss_resolution (KISSMatcherConfig(resolution), models/pose_estimator.py:61) */
    int32_t keypoint_tiles;               /* no reference counterpart.  0/1: throughput form of bx_register_pair (everything on the
                                           * caller's stream).  2..BX_MAX_TILES: latency form -- furthest point sampling runs on an
                                           * internal stream in that many launches (the first ends at num_points_radius_estimate)
                                           * and the descriptors of a tile of keypoints are computed while the next tile is still
                                           * being sampled.  Results are bit-identical in both forms. */
    /* Arithmetic forms of the three stages that have more than one (no reference counterpart: the reference leaves the summation order
     * of its convolutions to cuDNN / ATen).  Every form has its own restatement in the oracle (oracle/oracle.py takes the same
     * names) and is bit-exact against it; the value in force is echoed in bx_result.arith_forms.  0 = the default of each. */
    int32_t desc_conv_form;               /* Cylindrical_Net: BX_DESC_CONV_WINOGRAD43 (F(4x4,3x3), all 8 layers) | _WINOGRAD43M (mixed F(4x4) / F(3x4) tiles) | _WINOGRAD22 (F(2x2,3x3),
                                           * the 6 layers with >= 64 output channels) | _DIRECT (fp32 fmaf chain chunk > tap > channel) */
    int32_t pose_conv_form;               /* CostNet layers 1..5: BX_POSE_CONV_WINOGRAD43 (valid F(4x4,3x3)) | _WINOGRAD22 (valid F(2x2,3x3)) |
                                           * _DIRECT */
    int32_t cost_l0_form;                 /* CostNet layer 0: BX_COST_L0_COLLAPSED (binary64 P - Q form) | BX_COST_L0_DIRECT (fp32
                                           * convolution of the implicit cost volume) */
} bx_params;
#define BX_DESC_CONV_WINOGRAD43 0
#define BX_DESC_CONV_WINOGRAD22 1
#define BX_DESC_CONV_DIRECT 2
#define BX_DESC_CONV_WINOGRAD43M 3   /* mixed tiles (round 6): F(4x4,3x3) on the output rows 0..3, F(3x4,3x3) on the rows 4..6 -- the all-F(4x4)
                                      * form's phantom 8th output row is never multiplied (-8 % MFMAs); own restatement bxo_conv_wino43m */
#define BX_POSE_CONV_WINOGRAD43 0
#define BX_POSE_CONV_WINOGRAD22 1
#define BX_POSE_CONV_DIRECT 2
#define BX_COST_L0_COLLAPSED 0
#define BX_COST_L0_DIRECT 1

/* BatchNorm-folded weights in kernel layout, HOST pointers (buffer-x_amd/weights.py: fold_and_pack).
 * Replaces test.py:86-94 load_state_dict + the nn.Conv/BatchNorm modules of models/patch_embedder.py:26-41
 * and models/patchnet.py:68-84,192-210.  Conv weights: [cin/16][taps][16][cout], bias [cout]. */
typedef struct bx_weights {
    const float *pnt_w;    /* [16][3] */
    const float *pnt_b;    /* [16]    */
    const float *pool_w1;  /* [16][32] */
    const float *pool_b1;  /* [16] */
    const float *pool_w2;  /* [16] */
    const float *pool_b2;  /* [1] */
    const float *desc_w[8];
    const float *desc_b[8];
    const float *pose_w[10];
    const float *pose_b[10];
} bx_weights;

/* Result record of one registered pair == return tuple of BufferX.forward (models/BUFFERX.py:466-467)
 * and the per-pair payload of the multi-GPU all-gather. */
typedef struct bx_result {
    double pose[16];          /* src -> tgt, row-major 4x4 (float32 values widened when pose_refine) */
    int32_t num_inliers;      /* len(result.correspondence_set) of the RANSAC call */
    int32_t num_mutual;       /* accumulated mutual matches  */
    int32_t num_inlier_ind;   /* |inlier_ind| of the last consensus step */
    int32_t scales_used;
    int32_t ransac_iters;     /* iterations visited by the last RANSAC call */
    int32_t refine_iters;
    int32_t status;           /* 0 ok; device-side error bits otherwise */
    int32_t arith_forms;      /* the arithmetic forms the pair was computed in: desc_conv_form | pose_conv_form << 8 | cost_l0_form << 16 */
    float des_r[BX_MAX_SCALES];
} bx_result;

/* channel c (0..15) of a 16-chunk lives in slot 4*(c%4) + c/4 */
static inline int bx_chunk_slot(int c) { return 4 * (c & 3) + (c >> 2); }

/* ---- lifecycle ------------------------------------------------------------------------------- */
int bx_create(int device_id, const bx_params *params, bx_ctx **out);
int bx_destroy(bx_ctx *ctx);
const char *bx_last_error(void);
int bx_load_weights(bx_ctx *ctx, const bx_weights *w);
/* bytes of the per-context workspace arena (for DESIGN.md / capacity planning) */
int64_t bx_workspace_bytes(const bx_ctx *ctx);

/* ---- measurement hooks (hipEvent timers; reference utils/gpu_timer.py:3-33 semantics) -------------------
 * When enabled, bx_register_pair brackets each stage with hipEvents on the caller's stream.  After the stream
 * is synchronised bx_profile_read() adds up elapsed milliseconds / launch counts per stage tag and clears the
 * event list.  Tags: 0 fps, 1 radius, 2 neighbour-gather (ball_group), 3 patch features, 4 Desc conv stack,
 * 5 desc head, 6 mutual matching, 7 CostNet + soft-argmax, 8 hypotheses + consensus, 9 RANSAC, 10 refinement,
 * 11 permute (stage entry point only: the whole-pair path permutes on the fly), 12 the neighbour-gather query kernel alone
 * (inside tag 2), 13 the batched grid + row-table build of all (cloud, scale) sets of a pair (once per pair; tag 2 then holds
 * the query launches only).  ms_out / count_out: HOST arrays of BX_PROF_TAGS entries.                     */
#define BX_PROF_TAGS 16
int bx_profile_enable(bx_ctx *ctx, int32_t on);
int bx_profile_read(bx_ctx *ctx, double *ms_out, int32_t *count_out);
/* diagnostics: in-kernel cycle stamps of the neighbour-gather query kernel (collected when the environment
 * variable BX_BALL_DEBUG is set): up to 64 records of 8 int64 {t0, setup, rows, scan, expand, output, drained, T}. */
int bx_debug_read(bx_ctx *ctx, int64_t *out, int32_t n);

/* ---- whole pair: replaces BufferX.forward's inference branch, models/BUFFERX.py:257-467 -------
 * src/tgt: device float32 [n][3].  perm_src/perm_tgt: device int32 permutations, one per scale
 * ([num_scales][n], stands in for np.random.choice at models/patch_embedder.py:96).  seed drives the
 * RANSAC sampling (and the >200k subsample).  result: HOST pointer (pinned memory recommended),
 * valid after the stream is synchronised.                                                      */
int bx_register_pair(bx_ctx *ctx, void *stream, const float *src, int32_t n_src, const float *tgt, int32_t n_tgt,
                     int32_t aligned_z, const int32_t *perm_src, const int32_t *perm_tgt, uint64_t seed,
                     bx_result *result);

/* ---- the same in two calls, with the early-exit decision taken on the HOST (models/BUFFERX.py:424-457: the reference tests
 * `should_exit` on the host after scale 0 and breaks out of its scale loop).  bx_register_pair enqueues every launch of every scale
 * and lets the kernels of the later scales return on a device flag: right for one pair alone (no host round trip), but with many
 * pairs in flight the ~100 empty launches of a pair that left -- a third of them asking for a whole CU's LDS per workgroup -- still
 * queue behind the other pairs' kernels: 2 % of the throughput of an early-exit configuration (BASELINE configs[4] geometry:
 * 89.4 -> 91.3 pairs/s, profiles/r06_overlap.txt), and a pair that left holds its context until they have drained.
 *   bx_register_pair_begin : everything up to and including the exit test of scale 0 (ALL scales when the early exit is off or
 *                            num_scales == 1), then an asynchronous copy of the decision to *exited_host (HOST int32, pinned memory
 *                            recommended: 1 = the pair left at scale 0).  The input pointers are remembered, not copied: they must
 *                            stay valid until bx_register_pair_finish has been enqueued.
 *   bx_register_pair_finish: called once the stream has passed the copy (event / stream synchronisation by the caller) with the
 *                            value read: enqueues the later scales unless `exited`, the final pose estimation, the refinement and
 *                            the result copy.  Same kernels in the same order as bx_register_pair on a pair that takes the same
 *                            way: results are bit-identical (tests/test_gpu_pair.py).  Throughput form only (keypoint_tiles <= 1);
 *                            BX_ERR_STATE without a pending begin.                                                              */
int bx_register_pair_begin(bx_ctx *ctx, void *stream, const float *src, int32_t n_src, const float *tgt, int32_t n_tgt,
                           int32_t aligned_z, const int32_t *perm_src, const int32_t *perm_tgt, uint64_t seed,
                           int32_t *exited_host);
int bx_register_pair_finish(bx_ctx *ctx, void *stream, int32_t exited, bx_result *result);

/* ---- capture of intermediates (parity tests at sizes the CPU oracle cannot run end to end) ----------------------------------
 * While a capture is set, bx_register_pair copies (stream-ordered, device to device) the intermediates of ONE scale into
 * caller-owned device buffers; every pointer is nullable.  The per-cloud transients (permuted cloud, patches, voxel features,
 * last conv map) are taken from cloud `cloud` (0 = src, 1 = tgt); the per-scale tensors from both clouds.  The cumulative tensors
 * are copied right after the consensus step of scale `scale`.  T_ransac: the binary64 pose after the last pose-estimation call (the returned
 * pose itself is bx_result.pose).  Not available in the latency form (keypoint_tiles > 1: bx_set_capture returns BX_ERR_STATE).  A parity test feeds these tensors, a sampled subset at a time, to the oracle stage that consumes them.
 * bx_set_capture(ctx, NULL) switches the capture off.  The struct is copied; the buffers must stay alive while it is set.      */
typedef struct bx_capture {
    int32_t scale, cloud;
    float *pts_perm;                  /* [n][3]           permuted cloud (models/patch_embedder.py:96-97)                  */
    float *patches;                   /* [K][P][3]        select_patches output                                             */
    float *feat;                      /* [K][3][140][16]  pnt_layer + max (chunked)                                         */
    float *x;                         /* [K][2][140][16]  Cylindrical_Net output (chunked)                                  */
    float *kpts[2];                   /* [K][3]                                                                             */
    float *desc[2];                   /* [K][32]                                                                            */
    float *equi[2];                   /* [K][140][32]                                                                       */
    float *R[2];                      /* [K][9]           patch rotations                                                   */
    int32_t *s_mids, *t_mids;         /* [K]              mutual matches of the scale                                       */
    float *ind;                       /* [K]              soft-argmax of CostNet                                            */
    float *R_cat, *t_cat, *ss_cat, *tt_cat;   /* [S*K][9], [S*K][3] x3   accumulated hypotheses / matched keypoints        */
    int32_t *cons_cnt;                /* [S*K]            inlier count of every hypothesis                                  */
    int32_t *inlier_ind;              /* [S*K]                                                                              */
    int32_t *counts;                  /* [4] = {m of the scale, M, C, best}                                                 */
    double *T_ransac;                 /* [16]                                                                               */
} bx_capture;
int bx_set_capture(bx_ctx *ctx, const bx_capture *cap);

/* Keypoint tiles of the latency form (bx_params.keypoint_tiles): writes the T + 1 tile boundaries 0 = b[0] < ... < b[T] into
 * bounds[BX_MAX_TILES + 1] and returns T (1 = not tiled: keypoint_tiles <= 1, or num_fps <= num_points_radius_estimate + 4).
 * Tile 0 ends at num_points_radius_estimate rounded up to a multiple of 4 (the radius estimation needs exactly those keypoints);
 * the rest is split evenly in multiples of 4.  Pure host arithmetic (no device call): what bx_register_pair uses. */
int bx_keypoint_tile_bounds(const bx_params *params, int32_t *bounds);

/* ---- stage entry points (parity tests + operator-level drop-ins) ----------------------------- */

/* pointnet2_ops.furthest_point_sample + gather_operation (models/BUFFERX.py:286-290,338-346).
 * idx_out int32 [m]; kpts_out float32 [m][3] (nullable). */
int bx_fps(bx_ctx *ctx, void *stream, const float *xyz, int32_t n, int32_t m, int32_t *idx_out, float *kpts_out);

/* density_aware_radius_estimation (models/BUFFERX.py:627-696) for `nthr` thresholds.
 * pts float32 [n_pts][3] (already subsampled if n_orig > 200000), kpts [nk][3].
 * des_r_out: device double [nthr].                                                              */
int bx_radius(bx_ctx *ctx, void *stream, const float *pts, int32_t n_pts, int64_t n_orig, const float *kpts,
              int32_t nk, const double *thresholds_host, int32_t nthr, double *des_r_out);

/* out[i] = pts[perm[i]]  (models/patch_embedder.py:96-97) */
int bx_permute(bx_ctx *ctx, void *stream, const float *pts, const int32_t *perm, int32_t n, float *out);

/* MiniSpinNet.select_patches: ball_query + grouping_operation + mask arithmetic
 * (models/patch_embedder.py:92-120).  pts_perm [n][3] already permuted; radius: device double (as
 * produced by bx_radius).  idx_out int32 [K][P] (nullable), patches_out float32 [K][P][3].       */
int bx_ball_group(bx_ctx *ctx, void *stream, const float *pts_perm, int32_t n, const float *kpts, int32_t K,
                  const double *radius, int32_t P, int32_t *idx_out, float *patches_out);

/* The COUNTED form of the two stages above (round 5; what bx_register_pair runs internally): slots [count, P) of a patch are
 * copies of the keypoint -- the padded slots (group_idx == group_idx[:, :, 0]) and slot P - 1 of models/patch_embedder.py:105-111
 * -- so bx_ball_group_counted writes only the first count_out[k] = clamp(hits, 1, P - 1) slots of patches_out [K][P][3] (the rest
 * of a row is left untouched) and bx_patch_features_counted takes (patches, counts, kpts) and produces R_out / feat_out
 * bit-identical to bx_patch_features on the padded patch (tests/test_gpu_counted.py).  counts[k] must lie in [1, P - 1]; values
 * outside that range are clamped into it by the kernels (never trusted as an index).                                            */
int bx_ball_group_counted(bx_ctx *ctx, void *stream, const float *pts_perm, int32_t n, const float *kpts, int32_t K,
                          const double *radius, int32_t P, float *patches_out, int32_t *count_out);
int bx_patch_features_counted(bx_ctx *ctx, void *stream, const float *patches, const int32_t *counts, const float *kpts,
                              int32_t K, int32_t P, const double *radius, int32_t aligned_z, float *R_out, float *feat_out);

/* axis_align + normalize + SPT + pnt_layer + max-pool (models/patch_embedder.py:122-170, 26-30, 73-77;
 * utils/common.py:431-498, 501-525, 709-726).  R_out float32 [K][9]; feat_out chunked [K][rad_n][ele*azi][16]. */
int bx_patch_features(bx_ctx *ctx, void *stream, const float *patches, int32_t K, int32_t P, const double *radius,
                      int32_t aligned_z, float *R_out, float *feat_out);

/* Cylindrical_Net conv stack + pool_layer + weighted pooling + norms
 * (models/patchnet.py:49-84, models/patch_embedder.py:78-83).  feat chunked [K][3][140][16];
 * desc_out [K][32]; equi_out [K][140][32].  x_out (nullable): last conv output, chunked [K][2][140][16]. */
int bx_desc_net(bx_ctx *ctx, void *stream, const float *feat, int32_t K, float *desc_out, float *equi_out, float *x_out);

/* one convolution layer of either stack (parity of the MFMA kernel): net 0 = Desc (layers 0..7),
 * net 1 = Pose (layers 1..9; layer 0 consumes the implicit cost volume, see bx_pose_net).
 * in/out chunked layout; units = patches or matches.                                            */
int bx_conv_layer(bx_ctx *ctx, void *stream, int32_t net, int32_t layer, const float *in, int32_t units, float *out);

/* BufferX.mutual_matching (models/BUFFERX.py:469-496; knn_cuda.KNN k=1).  s_mids/t_mids int32 [ns],
 * count_out device int32 [1]. */
int bx_mutual(bx_ctx *ctx, void *stream, const float *src_des, int32_t ns, const float *tgt_des, int32_t nt,
              int32_t *s_mids, int32_t *t_mids, int32_t *count_out);

/* CostVolume.forward (models/BUFFERX.py:51-69) incl. CostNet (models/patchnet.py:192-210), softmax and
 * soft-argmax.  m_dev: device int32 count of matches (<= max_m).  ind_out float32 [max_m].
 * logits_out (nullable) float32 chunked [max_m][2][1][16].                                      */
int bx_pose_net(bx_ctx *ctx, void *stream, const float *s_equi, const float *t_equi, const int32_t *s_mids,
                const int32_t *t_mids, const int32_t *m_dev, int32_t max_m, float *ind_out, float *logits_out);

/* hypothesis recovery (models/BUFFERX.py:382-389; kornia axis_angle_to_rotation_matrix).
 * s_R/t_R [K][9], kpts [K][3] indexed through s_mids/t_mids; writes R_out [m][9], t_out [m][3],
 * ss_out/tt_out [m][3] (matched keypoints).                                                     */
int bx_hypotheses(bx_ctx *ctx, void *stream, const float *ind, const int32_t *s_mids, const int32_t *t_mids,
                  const int32_t *m_dev, int32_t max_m, const float *s_R, const float *t_R, const float *s_kpts,
                  const float *t_kpts, float *R_out, float *t_out, float *ss_out, float *tt_out);

/* cross-scale consensus maximisation (models/BUFFERX.py:405-417).  M_dev: device count.
 * inlier_out int32 [max_M], count_out device int32 [1], best_out device int32 [1] (nullable).    */
int bx_consensus(bx_ctx *ctx, void *stream, const float *R, const float *t, const float *ss, const float *tt,
                 const int32_t *M_dev, int32_t max_M, int32_t *inlier_out, int32_t *count_out, int32_t *best_out);

/* PoseEstimator._estimate_ransac (models/pose_estimator.py:84-117; Open3D 0.18
 * registration_ransac_based_on_correspondence), seeded.  corr int32 [C_dev]; T_out device double [16];
 * info_out device int32 [2] = {num_inliers, iterations visited}.                                */
int bx_ransac(bx_ctx *ctx, void *stream, const float *ss, const float *tt, const int32_t *corr, const int32_t *C_dev,
              int32_t max_C, uint64_t seed, double *T_out, int32_t *info_out);

/* PoseEstimator._estimate_kiss_matcher (models/pose_estimator.py:50-82): KISSMatcher(KISSMatcherConfig(kiss_resolution)).solve on
 * the correspondences corr[0..*C_dev) -- maximum-k-core pruning of the compatibility graph, GNC-TLS rotation, component-wise TLS
 * translation; restated from the published algorithm (the package is not vendored by the reference: see oracle/bx_oracle.c).
 * The context must have been created with params.pose_estimator = 1.  T_out device double [16]; info_out device int32 [4] =
 * {final inliers (get_num_final_inliers), core size, rotation inliers, GNC iterations}.                                    */
int bx_kiss_solve(bx_ctx *ctx, void *stream, const float *ss, const float *tt, const int32_t *corr, const int32_t *C_dev,
                  int32_t max_C, double *T_out, int32_t *info_out);

/* BufferX.post_refinement + rigid_transform_3d (models/BUFFERX.py:522-603).  T_io device float [16];
 * iters_out device int32 [1] (nullable).                                                        */
int bx_refine(bx_ctx *ctx, void *stream, const float *ss, const float *tt, const int32_t *M_dev, int32_t max_M,
              float *T_io, int32_t *iters_out);

/* ---- lane: ordering of the pairs in flight on ONE GPU (optional) -----------------------------------------------------------
 * The reference runs one pair at a time (test.py:132-146).  Here several contexts / streams keep pairs in flight so that the
 * latency-bound furthest-point sampling of one pair overlaps the convolutions of another.  Contexts attached to the same lane
 * additionally run their MFMA-bound sections one after the other, in submission order, instead of interleaving them workgroup by
 * workgroup: mode 1 = everything after the FPS, mode 2 = the two convolution stacks only.  Results do not depend on it.
 * Calls that use a lane must come from one host thread (or be serialised by the caller).                                     */
int bx_lane_create(int32_t mode, bx_lane **out);
int bx_lane_destroy(bx_lane *lane);
int bx_attach_lane(bx_ctx *ctx, bx_lane *lane);          /* lane == NULL detaches */

/* ---- pre-processing in front of the hot path (optional: the hot path does not depend on it) ----------
 * The reference prepares every pair on the host: utils/tools.py:152-198 sphericity_based_voxel_analysis (scikit-learn PCA of a
 * 10 % subsample, z-range in the PCA frame -> voxel size) and open3d voxel_down_sample (dataset/threedmatch.py:90-102,
 * dataset/kitti.py, dataset/tiers.py).  These entry points move both onto the GPU; buffer-x_amd/preprocess.py is the mirror of
 * the two reference functions on top of them.
 *
 * bx_pre_reserve: workspace for clouds of up to max_points RAW points (<= 2^26; memory O(max_points) whatever the extent of the
 * scene -- the voxels live in a hash table; allocates, not stream-ordered; call once).
 * bx_pre_voxel_downsample: pts float32 [n][3] -> out float32 [<= n][3] voxel centroids (binary64 accumulation in input order like
 * Open3D's AddPoint loop, emitted in order of first appearance); count_out device int32[2] = {number of voxels, status
 * (1 = more than 2^21 voxels along an axis, i.e. the voxel size is too small for the extent: nothing written)}.
 * bx_pre_pca: PCA (covariance + symmetric eigen-decomposition, sklearn sign convention) of pts[sample_idx[0..ns)] and the extent of
 * ALL n points along the 3rd component.  out17 device double[17] = {explained_variance[3] descending, components[3][3] rows,
 * mean[3], zmin, zmax}.                                                                                              */
int bx_pre_reserve(bx_ctx *ctx, int64_t max_points);
int bx_pre_voxel_downsample(bx_ctx *ctx, void *stream, const float *pts, int32_t n, double voxel_size, float *out,
                            int32_t *count_out);
int bx_pre_pca(bx_ctx *ctx, void *stream, const float *pts, int32_t n, const int32_t *sample_idx, int32_t ns, double *out17);
/* bx_random_perm: out[0..n) = a permutation of [0, n) determined by seed, computed on the device (4-round Feistel network keyed by
 * the library's mix64 + cycle walking; oracle/pre_oracle.py::random_perm).  Stands in for the loaders' / the descriptor's host-side
 * np.random.shuffle / np.random.choice(N, N, replace=False) (dataset/threedmatch.py:99,109, models/patch_embedder.py:96) when the
 * reference's exact NumPy stream is not needed; the result of bx_register_pair is defined for ANY permutations it is given. */
int bx_random_perm(bx_ctx *ctx, void *stream, int32_t n, uint64_t seed, int32_t *out);

/* ---- data ingest in front of the hot path (optional) ----------------------------------------------
 * The reference opens every pair synchronously on the main thread: open3d.io.read_point_cloud for .ply
 * (dataset/threedmatch.py:75-79) and .pcd (dataset/tiers.py:72-73), np.fromfile(float32).reshape(-1, 4)[:, :3] for KITTI .bin
 * (dataset/kitti.py:76-80), then a blocking .cuda().
 *
 * bx_io_probe: number of points of a .ply / .pcd / .bin file (header only for ply / pcd).
 * bx_io_read_xyz: xyz of every point as float32 [n][3] into a HOST buffer of `capacity` points (PLY ascii / binary little / big
 * endian with any scalar or list properties; PCD ascii / binary / binary_compressed; float64 coordinates are rounded to
 * float32, which is what the loaders hand to the model).
 * bx_prefetch_*: `slots` pairs of pinned host + device buffers of max_points points each and one worker thread.
 *   submit  queues the two files of a pair; the worker parses them into pinned memory and uploads them with hipMemcpyAsync on
 *           its own stream (returns BX_ERR_STATE when every slot is in use).
 *   wait    blocks the HOST until the upload has been issued, makes `stream` wait for the DMA (hipStreamWaitEvent) and returns
 *           the device pointers (owned by the prefetcher, valid until release).
 *   release the slot may be refilled once everything queued on `stream` so far has run.
 * One submitting / waiting host thread per prefetcher.                                                                       */
int bx_io_probe(const char *path, int64_t *n_points);
int bx_io_read_xyz(const char *path, float *xyz_out, int64_t capacity, int64_t *n_points);
int bx_prefetch_create(int32_t device, int32_t slots, int64_t max_points, bx_prefetch **out);
int bx_prefetch_submit(bx_prefetch *p, const char *src_path, const char *tgt_path, int64_t *ticket);
int bx_prefetch_wait(bx_prefetch *p, int64_t ticket, void *stream, const float **src_dev, int64_t *n_src, const float **tgt_dev,
                     int64_t *n_tgt);
int bx_prefetch_release(bx_prefetch *p, int64_t ticket, void *stream);
int bx_prefetch_destroy(bx_prefetch *p);

#ifdef __cplusplus
}
#endif
#endif /* BUFFERX_H */
