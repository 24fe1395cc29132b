// bx_api.hip -- C-ABI of libbufferx_hip.so (include/bufferx.h): context, weights, stage entry points and
// the whole-pair pipeline (reference BufferX.forward inference branch, models/BUFFERX.py:257-467).
#include "bx_common.h"
#include <atomic>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>

static thread_local char g_err[1024] = "";
void bx_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

struct ProfEvt { hipEvent_t a, b; int tag; int weight; };      // weight: launches the bracket stands for (2: both clouds of a scale in one convolution stack)
struct ProfScope {
    bx_ctx* c; hipStream_t s; ProfEvt e; bool on;
    ProfScope(bx_ctx* c_, hipStream_t s_, int tag, int weight = 1) : c(c_), s(s_), on(c_->prof_on != 0)
    {
        if (!on) return;
        e.tag = tag;
        e.weight = weight;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(e.a, s);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(e.b, s);
        static_cast<std::vector<ProfEvt>*>(c->prof)->push_back(e);
    }
};

}  // namespace

// contexts alive per device (this process) and per XCD pair: the FPS launcher aims the co-located launches of a context at one of the
// four XCD pairs and sizes the co-location by the contexts that REALLY share that pair (k_fps.hip).  A slot is released in bx_destroy:
// contexts that come and go (model.py re-creates its context for a larger cloud, bench.py opens and closes one) do not pile up on one pair.
static std::atomic<int> g_live[64];
static std::mutex g_slot_mu;
static int g_slot_use[64][4];
int bx_live_contexts(int device) { return device >= 0 && device < 64 ? g_live[device].load() : 1 << 20; }
int bx_xcd_pair_sharing(int device, int pair)
{
    if (device < 0 || device >= 64 || pair < 0 || pair > 3) return 1 << 20;
    std::lock_guard<std::mutex> lk(g_slot_mu);
    return g_slot_use[device][pair];
}
int bx_xcd_slot_take(int device)      // called by the FIRST furthest-point-sampling launch of a context (k_fps.hip): contexts that never sample
{                                     // (harness.py's preparation-only context) do not count towards the sharing of an XCD pair
    if (device < 0 || device >= 64) return 0;
    std::lock_guard<std::mutex> lk(g_slot_mu);
    int best = 0;
    for (int p = 1; p < 4; ++p) if (g_slot_use[device][p] < g_slot_use[device][best]) best = p;
    ++g_slot_use[device][best];
    return best;
}
static void xcd_slot_release(int device, int pair)
{
    std::lock_guard<std::mutex> lk(g_slot_mu);
    if (g_slot_use[device][pair] > 0) --g_slot_use[device][pair];
}

// event bracket usable from the other translation units (tag 12 = the neighbour-gather query kernel alone)
void bx_prof_mark(bx_ctx* c, hipStream_t s, int tag, int begin)
{
    if (!c->prof_on) return;
    static thread_local ProfEvt cur;
    if (begin) {
        cur.tag = tag;
        cur.weight = 1;
        if (hipEventCreate(&cur.a) != hipSuccess || hipEventCreate(&cur.b) != hipSuccess) { cur.tag = -1; return; }
        (void)hipEventRecord(cur.a, s);
    } else if (cur.tag == tag) {
        (void)hipEventRecord(cur.b, s);
        static_cast<std::vector<ProfEvt>*>(c->prof)->push_back(cur);
        cur.tag = -1;
    }
}

namespace {

struct Carver {
    char* base;
    size_t off;
    template <typename T>
    T* take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

void carve(bx_ctx* c, char* base, size_t* total)
{
    const size_t K = (size_t)c->p.num_fps, P = (size_t)c->p.num_points_per_patch, S = (size_t)c->p.num_scales;
    const size_t NK = (size_t)c->p.num_points_radius_estimate;
    const size_t KM = K > NK ? K : NK;
    const size_t NMAX = (size_t)c->p.max_points;
    const size_t SK = S * K;
    Carver cv{base, 0};
    // largest map: CostNet layer-0 output (K x 2 x 972 x 16) or the 128-channel Desc maps of BOTH clouds of a scale (2 K x 8 x 140 x 16:
    // the whole-pair path runs the two clouds' stacks as one launch per layer -- round 6)
    const size_t act = std::max(K * 2 * 972 * 16, 2 * K * 8 * BX_EA * 16);
    c->act0 = cv.take<float>(act);
    c->act1 = cv.take<float>(act);
    c->patches = cv.take<float>(K * P * 3);
    c->pcnt = cv.take<int32_t>(K);
    c->feat = cv.take<float>(2 * K * BX_RAD * BX_EA * 16);      // both clouds of a scale
    c->pts_perm = cv.take<float>(NMAX * 3);
    for (int i = 0; i < 2; ++i) {
        c->fps_idx[i] = cv.take<int32_t>(KM);
        c->kpts[i] = cv.take<float>(KM * 3);
        c->kpts_r[i] = c->kpts[i];
        c->desc_out[i] = cv.take<float>(K * 32);
        c->equi[i] = cv.take<float>(K * BX_EA * 32);
        c->Rpatch[i] = cv.take<float>(K * 9);
        c->nn_key[i] = cv.take<unsigned long long>(K);
    }
    const bool tiled = c->p.keypoint_tiles > 1;
    for (size_t sc = 0; sc < (size_t)BX_MAX_SCALES; ++sc)
        for (int i = 0; i < 2; ++i) {
            const bool own = tiled && sc > 0 && sc < S;     // scale 0 uses the shared arrays
            c->desc_sc[sc][i] = own ? cv.take<float>(K * 32) : c->desc_out[i];
            c->equi_sc[sc][i] = own ? cv.take<float>(K * BX_EA * 32) : c->equi[i];
            c->R_sc[sc][i] = own ? cv.take<float>(K * 9) : c->Rpatch[i];
        }
    for (int i = 0; i < 2; ++i) c->fps_td[i] = tiled ? cv.take<float>(NMAX) : nullptr;
    c->patches2 = tiled ? cv.take<float>(K * P * 3) : nullptr;
    c->pcnt2 = tiled ? cv.take<int32_t>(K) : nullptr;
    c->feat2 = tiled ? cv.take<float>(K * BX_RAD * BX_EA * 16) : nullptr;
    for (int i = 0; i < 2; ++i) c->act2[i] = tiled ? cv.take<float>(K * 8 * BX_EA * 16) : nullptr;   // largest Cylindrical_Net map: 128 channels
    for (int i = 0; i < 2; ++i) c->act3[i] = tiled ? cv.take<float>(K * 8 * BX_EA * 16) : nullptr;
    c->s_mids = cv.take<int32_t>(K);
    c->t_mids = cv.take<int32_t>(K);
    c->ind = cv.take<float>(K);
    c->R_cat = cv.take<float>(SK * 9);
    c->t_cat = cv.take<float>(SK * 3);
    c->ss_cat = cv.take<float>(SK * 3);
    c->tt_cat = cv.take<float>(SK * 3);
    c->cons_cnt = cv.take<int32_t>(SK);
    c->cons_thr = cv.take<float>(SK);
    c->inlier_ind = cv.take<int32_t>(SK);
    c->rad_hist = cv.take<unsigned long long>(8200);
    c->fps_dist = nullptr;
    c->fps_slots = cv.take<unsigned long long>(2 * 2 * 64 * 64);     // k_fps.hip: [cloud][parity][FPS_MAX_G][FPS_REC = 64]
    c->fps_hello = cv.take<unsigned long long>(2 * 64);
    c->fps_ord = cv.take<int32_t>(2 * NMAX);
    c->fps_cell = cv.take<unsigned short>(2 * NMAX);
    c->fps_cnt = cv.take<int>(2 * 4096 + 8);
    c->fps_bbmax = reinterpret_cast<unsigned*>(c->fps_cnt + 2 * 4096);     // zeroed together with the counters
    c->fps_bbmin = cv.take<unsigned>(8);
    c->ransac_inl = cv.take<int32_t>(BX_RANSAC_BATCH);
    c->ransac_err = cv.take<double>(BX_RANSAC_BATCH);
    c->ransac_T = cv.take<double>((size_t)BX_RANSAC_BATCH * 12);
    c->refine_ws = cv.take<float>(SK * 2);
    c->refine_sel = cv.take<int32_t>(SK);
    c->sub_pts = cv.take<float>(NMAX > 200000 ? (size_t)200000 * 3 : 16);
    const size_t NS = 2 * S;                      // grid sets: (cloud, scale)
    c->ball_nsets = (int)NS;
    c->ball_st_cnt = (size_t)BX_BALL_NCELL + 2 * 2048;
    c->ball_st_bsum = BX_BALL_NCELL / 2048 + 2;
    c->ball_st_pts = (NMAX + 63) & ~(size_t)63;
    c->ball_st_tab = KM * 256;                    // k_ball.hip NPMAX pieces per keypoint
    c->ball_st_num = KM;
    c->ball_bbox_part = cv.take<float>(2 * 64 * 6);
    c->ball_grid = cv.take<BallGrid>(NS);
    c->ball_cnt = cv.take<int32_t>(NS * c->ball_st_cnt);
    c->ball_start = cv.take<int32_t>(NS * c->ball_st_cnt);
    c->ball_bsum = cv.take<int32_t>(NS * c->ball_st_bsum);
    c->ball_cellrank = cv.take<int2>(NS * c->ball_st_pts);
    c->ball_ptab = cv.take<int2>(NS * c->ball_st_tab);
    c->ball_pnum = cv.take<int32_t>(NS * c->ball_st_num);
    c->ball_pts4 = cv.take<float4>(NS * c->ball_st_pts);
    c->ball_sorted = cv.take<float4>(NS * c->ball_st_pts);
    c->ball_dbg = cv.take<long long>(64 * 8);
    c->state = cv.take<PairState>(1);
    c->result_dev = cv.take<bx_result>(1);
    c->err_flag = cv.take<int32_t>(4);
    c->conv_ctr = cv.take<int32_t>(2 * BX_NDESC);
    *total = (cv.off + 255) & ~(size_t)255;
}

// host restatement of get_voxel_coordinate / var_to_invar tables (reference utils/common.py:248-262, 390-405,
// 422-428, 483-493, 117-128): binary64 libm like numpy, then rounded to fp32 like torch.FloatTensor.
void voxel_tables(std::vector<float>& cen, std::vector<float>& rot, std::vector<float>& rowc)
{
    const double PI = 3.14159265358979323846;
    cen.resize((size_t)BX_VOX * 3);
    rot.resize((size_t)BX_AZI * 4);
    rowc.resize((size_t)BX_RAD * BX_ELE * 2);
    for (int s = 0; s < BX_RAD; ++s) {
        double scale = (double)s / (double)BX_RAD + 1.0 / (double)(2 * BX_RAD);
        for (int e = 0; e < BX_ELE; ++e) {
            double beta = (double)e * (PI / (double)BX_ELE) + PI / (double)BX_ELE / 2.0;
            rowc[(size_t)(s * BX_ELE + e) * 2] = (float)(scale * std::sin(beta));
            rowc[(size_t)(s * BX_ELE + e) * 2 + 1] = (float)(scale * std::cos(beta));
            for (int a = 0; a < BX_AZI; ++a) {
                double alpha = (double)a * (2.0 * PI / (double)BX_AZI) + PI / (double)BX_AZI;
                double x = std::sin(beta) * std::cos(alpha), y = std::sin(beta) * std::sin(alpha), z = std::cos(beta);
                size_t o = ((size_t)(s * BX_ELE + e) * BX_AZI + a) * 3;
                cen[o] = (float)(scale * x); cen[o + 1] = (float)(scale * y); cen[o + 2] = (float)(scale * z);
            }
        }
    }
    for (int a = 0; a < BX_AZI; ++a) {
        double ang = -1.0 * (double)a * (2.0 * PI / (double)BX_AZI);
        rot[a * 4 + 0] = (float)std::cos(ang);
        rot[a * 4 + 1] = (float)(-std::sin(ang));
        rot[a * 4 + 2] = (float)std::sin(ang);
        rot[a * 4 + 3] = (float)std::cos(ang);
    }
}

// LDS staging geometry of the convolution kernel (k_conv.hip): cylindrical maps (reference pad_image / pad_image_3d,
// utils/common.py:265-310: circular in azimuth, zero in elevation) are staged with a halo so that a tap is a constant
// row offset; the un-padded ("valid") CostNet convolutions (models/patchnet.py:192-210) need no halo.
struct ConvGeo { std::vector<int32_t> lrow, lrow2, obase, toff; int p_lds; };

ConvGeo cyl_geo()
{
    ConvGeo g;
    const int WP = BX_AZI + 2;
    g.p_lds = (BX_ELE + 2) * WP;
    g.lrow.resize(BX_EA); g.lrow2.resize(BX_EA); g.obase.resize(BX_EA); g.toff.resize(9);
    for (int h = 0; h < BX_ELE; ++h)
        for (int w = 0; w < BX_AZI; ++w) {
            const int p = h * BX_AZI + w;
            g.lrow[p] = (h + 1) * WP + (w + 1);
            g.lrow2[p] = w == 0 ? (h + 1) * WP + (BX_AZI + 1) : (w == BX_AZI - 1 ? (h + 1) * WP : -1);
            g.obase[p] = h * WP + w;
        }
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) g.toff[kh * 3 + kw] = kh * WP + kw;
    return g;
}

ConvGeo valid_geo(const int d[3], const int k[3], int o[3])
{
    for (int i = 0; i < 3; ++i) o[i] = d[i] - k[i] + 1;
    ConvGeo g;
    const int pin = d[0] * d[1] * d[2];
    g.p_lds = pin;
    g.lrow.resize(pin); g.lrow2.assign(pin, -1);
    for (int p = 0; p < pin; ++p) g.lrow[p] = p;
    g.obase.resize((size_t)o[0] * o[1] * o[2]);
    for (int z = 0; z < o[0]; ++z)
        for (int y = 0; y < o[1]; ++y)
            for (int x = 0; x < o[2]; ++x) g.obase[(size_t)(z * o[1] + y) * o[2] + x] = (z * d[1] + y) * d[2] + x;
    for (int a = 0; a < k[0]; ++a)
        for (int b = 0; b < k[1]; ++b)
            for (int c = 0; c < k[2]; ++c) g.toff.push_back((a * d[1] + b) * d[2] + c);
    return g;
}

template <typename T>
int upload(T** dst, const T* src, size_t n)
{
    BX_HIP(hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(T)));
    BX_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return BX_OK;
}

__global__ void subsample_kernel(const float* __restrict__ pts, int n, unsigned long long seed, int cnt, float* __restrict__ out)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cnt) return;
    uint64_t k = bx_mix64(seed, (0x200ULL << 32) + (uint64_t)j) % (uint64_t)n;
    out[(size_t)j * 3] = pts[k * 3]; out[(size_t)j * 3 + 1] = pts[k * 3 + 1]; out[(size_t)j * 3 + 2] = pts[k * 3 + 2];
}

__global__ void state_reset_kernel(PairState* st, int32_t* err)
{
    if (threadIdx.x == 0) {
        st->m_scale = 0; st->M = 0; st->C = 0; st->best = -1; st->done = 0; st->scales_used = 0; st->num_inliers = 0;
        st->ransac_iters = 0; st->refine_iters = 0; st->status = 0;
        for (int i = 0; i < BX_MAX_SCALES; ++i) st->des_r[i] = 0.0;
        for (int i = 0; i < 16; ++i) { st->T[i] = (i % 5 == 0) ? 1.0 : 0.0; st->Tf[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
        err[0] = 0;
    }
}

__global__ void accumulate_kernel(PairState* st, int scale, const int32_t* skip)
{
    if (threadIdx.x == 0 && !(skip && *skip)) { st->M += st->m_scale; st->scales_used = scale + 1; }
}

__global__ void early_exit_kernel(PairState* st, int min_inliers)
{
    if (threadIdx.x == 0) st->done = st->num_inliers >= min_inliers ? 1 : 0;
}

__global__ void pose_to_float_kernel(PairState* st)
{
    if (threadIdx.x < 16) st->Tf[threadIdx.x] = (float)st->T[threadIdx.x];
}

__global__ void finalize_kernel(const PairState* st, const int32_t* err, int refine, int nscales, int forms, bx_result* out)
{
    if (threadIdx.x == 0) {
        for (int i = 0; i < 16; ++i) out->pose[i] = refine ? (double)st->Tf[i] : st->T[i];
        out->num_inliers = st->num_inliers;
        out->num_mutual = st->M;
        out->num_inlier_ind = st->C;
        out->scales_used = st->scales_used;
        out->ransac_iters = st->ransac_iters;
        out->refine_iters = st->refine_iters;
        out->status = err[0];
        out->arith_forms = forms;
        for (int i = 0; i < BX_MAX_SCALES; ++i) out->des_r[i] = i < nscales ? (float)st->des_r[i] : 0.0f;
    }
}

// bx_set_capture: stream-ordered device-to-device copy into a caller buffer (no-op for a null destination)
template <typename T>
int cap_copy(hipStream_t s, T* dst, const T* src, size_t n)
{
    if (dst && n) BX_HIP(hipMemcpyAsync(dst, src, sizeof(T) * n, hipMemcpyDeviceToDevice, s));
    return BX_OK;
}

__global__ void cap_counts_kernel(const PairState* st, int32_t* counts, double* T)
{
    if (threadIdx.x == 0 && counts) { counts[0] = st->m_scale; counts[1] = st->M; counts[2] = st->C; counts[3] = st->best; }
    if (T && threadIdx.x < 16) T[threadIdx.x] = st->T[threadIdx.x];
}

#define BX_ENTER(c, need_w) int rc; if ((rc = check_ctx((c), (need_w))) != BX_OK) return rc; BxDevScope ds_((c)->device)

int check_ctx(bx_ctx* c, bool need_weights)
{
    if (!c) { bx_set_error("null context"); return BX_ERR_ARG; }
    if (need_weights && !c->weights_loaded) { bx_set_error("weights not loaded (bx_load_weights)"); return BX_ERR_STATE; }
    return BX_OK;
}

int desc_stack(bx_ctx* c, hipStream_t s, const float* feat, int K, float* desc, float* equi, float* x_out, float* const* scratch = nullptr)
{
    const float* in = feat;
    float* bufs[2] = {scratch ? scratch[0] : c->act0, scratch ? scratch[1] : c->act1};
    int rc;
    for (int l = 0; l < BX_NDESC; ++l) {
        float* out = bufs[l & 1];
        if ((rc = bxk_conv(c, s, 0, l, in, nullptr, K, out)) != BX_OK) return rc;
        in = out;
    }
    if (x_out) BX_HIP(hipMemcpyAsync(x_out, in, sizeof(float) * (size_t)K * 2 * BX_EA * 16, hipMemcpyDeviceToDevice, s));
    return bxk_desc_head(c, s, in, K, desc, equi);
}

// both clouds of a scale as ONE stack of 2 K units (round 6): a unit's arithmetic does not depend on what shares its launch, so the
// descriptors are those of two K-unit stacks bit for bit; one pair alone pays 13 instead of 2 x 7 persistent rounds on the 64-column
// layers (25 instead of 2 x 13 on the 128-column ones).  feat [2 K][3][140][16], heads per cloud.
int desc_stack_pair(bx_ctx* c, hipStream_t s, const float* feat, int K, float* const* desc, float* const* equi)
{
    const float* in = feat;
    float* bufs[2] = {c->act0, c->act1};
    int rc;
    for (int l = 0; l < BX_NDESC; ++l) {
        float* out = bufs[l & 1];
        if ((rc = bxk_conv(c, s, 0, l, in, nullptr, 2 * K, out)) != BX_OK) return rc;
        in = out;
    }
    for (int cl = 0; cl < 2; ++cl)
        if ((rc = bxk_desc_head(c, s, in + (size_t)cl * K * 2 * BX_EA * 16, K, desc[cl], equi[cl])) != BX_OK) return rc;
    return BX_OK;
}

int pose_stack(bx_ctx* c, hipStream_t s, const float* s_equi, const float* t_equi, const int32_t* s_mids, const int32_t* t_mids,
               const int32_t* m_dev, int max_m, float* ind, float* logits_out)
{
    int rc;
    // layer 0 on the implicit cost volume: the collapsed binary64 form (k_cost.hip) unless bx_params.cost_l0_form = BX_COST_L0_DIRECT asks for the fp32 MFMA
    // convolution of the volume (round-1/2 kernel, kept for A/B measurements; its arithmetic contract is the oracle's "direct" form)
    if (c->cost_direct) rc = bxk_cost_l1(c, s, s_equi, t_equi, s_mids, t_mids, m_dev, max_m, c->act0);
    else rc = bxk_cost_l0(c, s, s_equi, t_equi, s_mids, t_mids, m_dev, max_m, c->act0);
    if (rc != BX_OK) return rc;
    float* bufs[2] = {c->act0, c->act1};
    const float* in = c->act0;
    for (int l = 1; l < BX_NPOSE; ++l) {
        float* out = bufs[l & 1];
        if ((rc = bxk_conv(c, s, 1, l, in, m_dev, max_m, out)) != BX_OK) return rc;
        in = out;
    }
    if (logits_out) BX_HIP(hipMemcpyAsync(logits_out, in, sizeof(float) * (size_t)max_m * 2 * 16, hipMemcpyDeviceToDevice, s));
    return bxk_soft_argmax(s, in, m_dev, max_m, ind, c->skip);
}

}  // namespace

extern "C" {

const char* bx_last_error(void) { return g_err; }

static int create_impl(bx_ctx* c, int device_id)
{
    const bx_params& p = c->p;
    {
        hipDeviceProp_t prop;
        BX_HIP(hipGetDeviceProperties(&prop, device_id));
        c->n_cu = prop.multiProcessorCount;
        const char* e = getenv("BX_CONV_PERSIST");
        c->conv_persist = (!e || atoi(e) != 0) ? 1 : 0;
        e = getenv("BX_CONV_PERSIST_CAP");
        c->conv_cap_override = e ? atoi(e) : 0;
        e = getenv("BX_DESC_BATCH");                 // measurement hook: 0 = one Cylindrical_Net stack per cloud (rounds 1-5); results do not depend on it
        c->desc_batch = (!e || atoi(e) != 0) ? 1 : 0;
        e = getenv("BX_RAD_SLICES");                 // measurement hook (k_radius.hip); results do not depend on it
        c->rad_slices = e ? atoi(e) : 0;
        // arithmetic forms: bx_params (validated by bx_create), never the environment
        c->use_wino = p.desc_conv_form == BX_DESC_CONV_DIRECT ? 0 : (p.desc_conv_form == BX_DESC_CONV_WINOGRAD22 ? 1 : (p.desc_conv_form == BX_DESC_CONV_WINOGRAD43M ? 3 : 2));
        c->use_wino_pose = p.pose_conv_form == BX_POSE_CONV_DIRECT ? 0 : (p.pose_conv_form == BX_POSE_CONV_WINOGRAD22 ? 1 : 2);
        c->cost_direct = p.cost_l0_form == BX_COST_L0_DIRECT ? 1 : 0;
    }
    (void)p;
    c->prof = new std::vector<ProfEvt>();
    size_t total = 0;
    carve(c, nullptr, &total);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->arena), total);
    if (e != hipSuccess) {
        c->arena = nullptr;
        bx_set_error("bx_create: workspace hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
        return BX_ERR_HIP;
    }
    c->arena_bytes = (int64_t)total;
    carve(c, c->arena, &total);
    BX_HIP(hipMemset(c->state, 0, sizeof(PairState)));
    BX_HIP(hipMemset(c->err_flag, 0, 4 * sizeof(int32_t)));
    BX_HIP(hipMemset(c->conv_ctr, 0, 2 * BX_NDESC * sizeof(int32_t)));
    BX_HIP(hipMemset(c->ball_cnt, 0, (size_t)c->ball_nsets * c->ball_st_cnt * sizeof(int32_t)));   // kept zero by scan_apply_kernel
    std::vector<float> cen, rot, rowc;
    voxel_tables(cen, rot, rowc);
    int rc;
    if ((rc = upload(&c->d_centres, cen.data(), cen.size())) != BX_OK) return rc;
    if ((rc = upload(&c->d_rot, rot.data(), rot.size())) != BX_OK) return rc;
    if ((rc = upload(&c->d_rowc, rowc.data(), rowc.size())) != BX_OK) return rc;
    std::vector<float> thr(8193);
    for (int m = 0; m <= 8192; ++m) {
        // the bisection of models/BUFFERX.py:675-692 only visits des_r = 5 m / 8192; compare value = fp32(des_r*des_r)
        double r = 5.0 * (double)m / 8192.0;
        thr[m] = (float)(r * r);
    }
    if ((rc = upload(&c->d_rad_thr, thr.data(), thr.size())) != BX_OK) return rc;
    if (p.keypoint_tiles > 1) {
        int lo = 0, hi = 0;
        BX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));      // hi = numerically lowest = most urgent
        BX_HIP(hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, hi));
        BX_HIP(hipStreamCreateWithFlags(&c->tgt_stream, hipStreamNonBlocking));
        BX_HIP(hipStreamCreateWithFlags(&c->match_stream, hipStreamNonBlocking));
        BX_HIP(hipEventCreateWithFlags(&c->ev_match_done, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < BX_MAX_SCALES; ++j) BX_HIP(hipEventCreateWithFlags(&c->ev_desc[i][j], hipEventDisableTiming));
        BX_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        BX_HIP(hipEventCreateWithFlags(&c->ev_tgt_go, hipEventDisableTiming));
        BX_HIP(hipEventCreateWithFlags(&c->ev_tgt_done, hipEventDisableTiming));
        for (int i = 0; i < BX_MAX_TILES; ++i) BX_HIP(hipEventCreateWithFlags(&c->ev_tile[i], hipEventDisableTiming));
    }
    if (p.pose_estimator == 1) {
        if (!(p.kiss_resolution > 0.0)) { bx_set_error("bx_create: kiss_resolution must be positive"); return BX_ERR_ARG; }
        c->kiss_max_C = p.num_fps * p.num_scales;
        const size_t kb = bxk_kiss_workspace_bytes(c->kiss_max_C);
        hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&c->kiss_ws), kb);
        if (e2 != hipSuccess) { c->kiss_ws = nullptr; bx_set_error("bx_create: KISS-Matcher workspace hipMalloc(%zu) failed: %s", kb, hipGetErrorString(e2)); return BX_ERR_HIP; }
    } else if (p.pose_estimator != 0) { bx_set_error("bx_create: pose_estimator must be 0 (ransac) or 1 (kiss_matcher)"); return BX_ERR_ARG; }
    return BX_OK;
}

int bx_create(int device_id, const bx_params* params, bx_ctx** out)
{
    if (!params || !out) { bx_set_error("bx_create: null argument"); return BX_ERR_ARG; }
    const bx_params& p = *params;
    if (p.rad_n != BX_RAD || p.ele_n != BX_ELE || p.azi_n != BX_AZI) {
        bx_set_error("bx_create: only rad_n=3, ele_n=7, azi_n=20 are supported (got %d %d %d)", p.rad_n, p.ele_n, p.azi_n);
        return BX_ERR_ARG;
    }
    if (p.num_fps < 1 || p.num_points_per_patch < 2 || p.num_scales < 1 || p.num_scales > BX_MAX_SCALES || p.max_points < 1 ||
        p.num_points_radius_estimate < 1 || p.voxel_sample < 1 || p.voxel_sample > 16 || p.keypoint_tiles < 0 ||
        p.keypoint_tiles > BX_MAX_TILES) {
        bx_set_error("bx_create: invalid parameters");
        return BX_ERR_ARG;
    }
    if (p.desc_conv_form < 0 || p.desc_conv_form > BX_DESC_CONV_WINOGRAD43M || p.pose_conv_form < 0 || p.pose_conv_form > BX_POSE_CONV_DIRECT ||
        p.cost_l0_form < 0 || p.cost_l0_form > BX_COST_L0_DIRECT) {
        bx_set_error("bx_create: unknown arithmetic form (desc_conv_form %d, pose_conv_form %d, cost_l0_form %d)", p.desc_conv_form,
                     p.pose_conv_form, p.cost_l0_form);
        return BX_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
        bx_set_error("bx_create: device %d not available (%d devices)", device_id, ndev);
        return BX_ERR_HIP;
    }
    BxDevScope ds(device_id);
    bx_ctx* c = new bx_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device_id;
    c->p = p;
    c->fps_xcd_pair = -1;              // taken lazily by the first FPS launch
    if (device_id < 64) g_live[device_id].fetch_add(1);
    const int rc = create_impl(c, device_id);
    if (rc != BX_OK) {
        // bx_destroy releases whatever had been allocated; it must not clobber the message of the failure
        char msg[1024];
        snprintf(msg, sizeof(msg), "%s", g_err);
        bx_destroy(c);
        bx_set_error("%s", msg);
        return rc;
    }
    *out = c;
    return BX_OK;
}

int bx_destroy(bx_ctx* c)
{
    if (!c) return BX_OK;
    BxDevScope ds(c->device);
    (void)hipDeviceSynchronize();
    if (c->device >= 0 && c->device < 64) { g_live[c->device].fetch_sub(1); if (c->fps_xcd_pair >= 0) xcd_slot_release(c->device, c->fps_xcd_pair); }
    bxk_pre_release(c);
    if (c->prof) {
        auto* v = static_cast<std::vector<ProfEvt>*>(c->prof);
        for (auto& e : *v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        delete v;
    }
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->tgt_stream) (void)hipStreamDestroy(c->tgt_stream);
    if (c->match_stream) (void)hipStreamDestroy(c->match_stream);
    if (c->ev_match_done) (void)hipEventDestroy(c->ev_match_done);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < BX_MAX_SCALES; ++j) if (c->ev_desc[i][j]) (void)hipEventDestroy(c->ev_desc[i][j]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_tgt_go) (void)hipEventDestroy(c->ev_tgt_go);
    if (c->ev_tgt_done) (void)hipEventDestroy(c->ev_tgt_done);
    for (int i = 0; i < BX_MAX_TILES; ++i) if (c->ev_tile[i]) (void)hipEventDestroy(c->ev_tile[i]);
    (void)hipFree(c->arena);
    (void)hipFree(c->kiss_ws);
    (void)hipFree(c->d_cost_wp); (void)hipFree(c->d_cost_wq);
    (void)hipFree(c->d_centres); (void)hipFree(c->d_rot); (void)hipFree(c->d_rowc); (void)hipFree(c->d_rad_thr);
    (void)hipFree(c->d_pnt_w); (void)hipFree(c->d_pnt_b); (void)hipFree(c->d_pool_w1); (void)hipFree(c->d_pool_b1); (void)hipFree(c->d_pool_w2); (void)hipFree(c->d_pool_b2);
    for (int i = 0; i < BX_NDESC; ++i) { (void)hipFree(c->desc[i].W); (void)hipFree(c->desc[i].Wwino); (void)hipFree(c->desc[i].Wwino43); (void)hipFree(c->desc[i].b); (void)hipFree(c->desc[i].lrow); (void)hipFree(c->desc[i].lrow2); (void)hipFree(c->desc[i].obase); (void)hipFree(c->desc[i].toff); }
    for (int i = 0; i < BX_NPOSE; ++i) { (void)hipFree(c->pose[i].W); (void)hipFree(c->pose[i].Wwino); (void)hipFree(c->pose[i].Wwino43); (void)hipFree(c->pose[i].b); (void)hipFree(c->pose[i].lrow); (void)hipFree(c->pose[i].lrow2); (void)hipFree(c->pose[i].obase); (void)hipFree(c->pose[i].toff); }
    delete c;
    return BX_OK;
}

int64_t bx_workspace_bytes(const bx_ctx* c) { return c ? c->arena_bytes : 0; }

int bx_debug_read(bx_ctx* c, int64_t* out, int32_t n)
{
    if (!c || !out || n < 0 || n > 64 * 8) { bx_set_error("bx_debug_read: bad argument"); return BX_ERR_ARG; }
    BX_HIP(hipMemcpy(out, c->ball_dbg, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost));
    return BX_OK;
}

int bx_keypoint_tile_bounds(const bx_params* params, int32_t* bounds)
{
    if (!params || !bounds) { bx_set_error("bx_keypoint_tile_bounds: null argument"); return 0; }
    const int K = params->num_fps, NK = params->num_points_radius_estimate, want = params->keypoint_tiles;
    for (int i = 0; i <= BX_MAX_TILES; ++i) bounds[i] = 0;
    int T = 1;
    if (want > 1 && want <= BX_MAX_TILES && K > NK + 4) {
        const int first = (NK + 3) & ~3;
        bounds[1] = first;
        for (int t = 2; t <= want; ++t) {
            const int e = t == want ? K : first + (int)((int64_t)(K - first) * (t - 1) / (want - 1)) / 4 * 4;
            if (e > bounds[T]) bounds[++T] = e;
        }
    }
    if (T == 1) bounds[1] = K > 0 ? K : 0;
    return T;
}

int bx_set_capture(bx_ctx* c, const bx_capture* cap)
{
    if (!c) { bx_set_error("null context"); return BX_ERR_ARG; }
    if (!cap) { c->cap_on = 0; return BX_OK; }
    if (cap->scale < 0 || cap->scale >= c->p.num_scales || cap->cloud < 0 || cap->cloud > 1) {
        bx_set_error("bx_set_capture: scale %d / cloud %d out of range", cap->scale, cap->cloud);
        return BX_ERR_ARG;
    }
    if (c->p.keypoint_tiles > 1) { bx_set_error("bx_set_capture: the latency form (keypoint_tiles > 1) cannot be captured"); return BX_ERR_STATE; }
    c->cap = *cap;
    c->cap_on = 1;
    return BX_OK;
}

int bx_profile_enable(bx_ctx* c, int32_t on)
{
    if (!c) { bx_set_error("null context"); return BX_ERR_ARG; }
    c->prof_on = on ? 1 : 0;
    return BX_OK;
}

int bx_profile_read(bx_ctx* c, double* ms_out, int32_t* count_out)
{
    if (!c || !ms_out || !count_out) { bx_set_error("bx_profile_read: null"); return BX_ERR_ARG; }
    for (int i = 0; i < BX_PROF_TAGS; ++i) { ms_out[i] = 0.0; count_out[i] = 0; }
    auto* v = static_cast<std::vector<ProfEvt>*>(c->prof);
    for (auto& e : *v) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess && e.tag >= 0 && e.tag < BX_PROF_TAGS) { ms_out[e.tag] += ms; count_out[e.tag] += e.weight; }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    v->clear();
    return BX_OK;
}

int bx_load_weights(bx_ctx* c, const bx_weights* w)
{
    BX_ENTER(c, false);
    if (!w) { bx_set_error("bx_load_weights: null"); return BX_ERR_ARG; }
    if (c->weights_loaded) { bx_set_error("bx_load_weights: already loaded"); return BX_ERR_STATE; }
    if ((rc = upload(&c->d_pnt_w, w->pnt_w, 48)) != BX_OK) return rc;
    if ((rc = upload(&c->d_pnt_b, w->pnt_b, 16)) != BX_OK) return rc;
    if ((rc = upload(&c->d_pool_w1, w->pool_w1, 512)) != BX_OK) return rc;
    if ((rc = upload(&c->d_pool_b1, w->pool_b1, 16)) != BX_OK) return rc;
    if ((rc = upload(&c->d_pool_w2, w->pool_w2, 16)) != BX_OK) return rc;
    if ((rc = upload(&c->d_pool_b2, w->pool_b2, 1)) != BX_OK) return rc;
    // Desc: Cylindrical_Net (models/patchnet.py:72-84)
    static const int dc[BX_NDESC][2] = {{3, 64}, {4, 64}, {4, 128}, {8, 128}, {8, 64}, {4, 64}, {4, 32}, {2, 32}};
    // B-fragment order of the MFMA kernels: [chunk*taps][column tile][lane = kk*16 + li][4] with element i =
    // W[chunk][tap][kk + 4 i][tile*16 + li] (zero beyond cout): one 16-byte load per lane feeds the 4 MFMAs of a tile
    auto upload_w = [&](ConvLayerDev& L, const float* wsrc) -> int {
        const int nct = L.nchunk * L.ntaps, nt = (L.cout + 15) / 16;
        std::vector<float> wp((size_t)nct * nt * 64 * 4, 0.0f);
        for (int ct = 0; ct < nct; ++ct)
            for (int t = 0; t < nt; ++t)
                for (int kk = 0; kk < 4; ++kk)
                    for (int li = 0; li < 16; ++li)
                        for (int i = 0; i < 4; ++i) {
                            const int col = t * 16 + li;
                            if (col < L.cout)
                                wp[((((size_t)ct * nt + t) * 4 + kk) * 16 + li) * 4 + i] = wsrc[((size_t)ct * 16 + kk + 4 * i) * L.cout + col];
                        }
        return upload(&L.W, wp.data(), wp.size());
    };
    const ConvGeo cg = cyl_geo();
    auto upload_geo = [&](ConvLayerDev& L, const ConvGeo& g) -> int {
        int r;
        L.p_lds = g.p_lds;
        if ((r = upload(&L.lrow, g.lrow.data(), g.lrow.size())) != BX_OK) return r;
        if ((r = upload(&L.lrow2, g.lrow2.data(), g.lrow2.size())) != BX_OK) return r;
        if ((r = upload(&L.obase, g.obase.data(), g.obase.size())) != BX_OK) return r;
        return upload(&L.toff, g.toff.data(), g.toff.size());
    };
    for (int l = 0; l < BX_NDESC; ++l) {
        ConvLayerDev& L = c->desc[l];
        L.nchunk = dc[l][0]; L.ntaps = 9; L.p_in = BX_EA; L.p_out = BX_EA; L.cout = dc[l][1]; L.relu = l < BX_NDESC - 1;
        if ((rc = upload_w(L, w->desc_w[l])) != BX_OK) return rc;
        if (c->use_wino == 1 && L.cout >= 64 && (rc = bxk_wino_weights(w->desc_w[l], L.nchunk, 1, L.cout, &L.Wwino)) != BX_OK) return rc;
        if (c->use_wino == 2 && (rc = bxk_wino43_weights(w->desc_w[l], L.nchunk, 1, L.cout, &L.Wwino43)) != BX_OK) return rc;
        if (c->use_wino == 3 && (rc = bxk_wino43m_weights(w->desc_w[l], L.nchunk, L.cout, &L.Wwino43)) != BX_OK) return rc;
        if ((rc = upload(&L.b, w->desc_b[l], (size_t)L.cout)) != BX_OK) return rc;
        if ((rc = upload_geo(L, cg)) != BX_OK) return rc;
    }
    // Pose: CostNet (models/patchnet.py:196-210) on the [azi, ele-2, azi] cost volume
    static const int pc[BX_NPOSE][2] = {{2, 32}, {2, 64}, {4, 64}, {4, 128}, {8, 128}, {8, 64}, {4, 64}, {4, 32}, {2, 32}, {2, 20}};
    int dims[3] = {BX_AZI, BX_ELE - 2, BX_AZI};
    for (int l = 0; l < BX_NPOSE; ++l) {
        int k[3] = {3, l < 2 ? 3 : 1, 3};
        if (l == BX_NPOSE - 1) { k[0] = 2; k[1] = 1; k[2] = 2; }
        int o[3];
        const ConvGeo vg = valid_geo(dims, k, o);
        ConvLayerDev& L = c->pose[l];
        L.nchunk = pc[l][0]; L.ntaps = k[0] * k[1] * k[2]; L.p_in = dims[0] * dims[1] * dims[2]; L.p_out = o[0] * o[1] * o[2];
        L.cout = pc[l][1]; L.relu = l < BX_NPOSE - 1;
        if ((rc = upload_w(L, w->pose_w[l])) != BX_OK) return rc;
        if ((rc = upload(&L.b, w->pose_b[l], (size_t)L.cout)) != BX_OK) return rc;
        if ((rc = upload_geo(L, vg)) != BX_OK) return rc;
        if (c->use_wino_pose == 1 && l >= 1 && l <= 5 && (rc = bxk_wino_weights(w->pose_w[l], L.nchunk, k[1], L.cout, &L.Wwino)) != BX_OK) return rc;
        if (c->use_wino_pose == 2 && l >= 1 && l <= 5 && (rc = bxk_wino43_weights(w->pose_w[l], L.nchunk, k[1], L.cout, &L.Wwino43)) != BX_OK) return rc;
        for (int i = 0; i < 3; ++i) dims[i] = o[i];
    }
    if ((rc = bxk_cost_l0_weights(w->pose_w[0], &c->d_cost_wp, &c->d_cost_wq)) != BX_OK) return rc;
    c->weights_loaded = true;
    return BX_OK;
}

// ------------------------------------------------------------------------------------------------ stages
int bx_fps(bx_ctx* c, void* stream, const float* xyz, int32_t n, int32_t m, int32_t* idx_out, float* kpts_out)
{
    BX_ENTER(c, false);
    if (!xyz || !idx_out || n < 1 || m < 1) { bx_set_error("bx_fps: bad argument"); return BX_ERR_ARG; }
    const float* xs[1] = {xyz};
    int ns[1] = {n};
    int32_t* is[1] = {idx_out};
    float* ks[1] = {kpts_out};
    return bxk_fps(c, (hipStream_t)stream, xs, ns, 1, m, is, ks);
}

int bx_radius(bx_ctx* c, void* stream, const float* pts, int32_t n_pts, int64_t n_orig, const float* kpts, int32_t nk,
              const double* thresholds_host, int32_t nthr, double* des_r_out)
{
    BX_ENTER(c, false);
    if (!pts || !kpts || !thresholds_host || !des_r_out || n_pts < 1 || nk < 1) { bx_set_error("bx_radius: bad argument"); return BX_ERR_ARG; }
    if ((rc = bxk_radius_hist(c, (hipStream_t)stream, pts, n_pts, kpts, nk)) != BX_OK) return rc;
    return bxk_radius_bisect_all(c, (hipStream_t)stream, n_orig, nk, thresholds_host, nthr, des_r_out);
}

int bx_permute(bx_ctx* c, void* stream, const float* pts, const int32_t* perm, int32_t n, float* out)
{
    BX_ENTER(c, false);
    return bx_permute_launch((hipStream_t)stream, pts, perm, n, out, nullptr);
}

int bx_ball_group(bx_ctx* c, void* stream, const float* pts_perm, int32_t n, const float* kpts, int32_t K, const double* radius,
                  int32_t P, int32_t* idx_out, float* patches_out)
{
    BX_ENTER(c, false);
    if (!pts_perm || !kpts || !radius || !patches_out) { bx_set_error("bx_ball_group: null argument"); return BX_ERR_ARG; }
    c->skip = nullptr;
    c->ball_waves_hint = 0;
    return bxk_ball_group(c, (hipStream_t)stream, pts_perm, n, kpts, K, radius, P, idx_out, patches_out);
}

int bx_ball_group_counted(bx_ctx* c, void* stream, const float* pts_perm, int32_t n, const float* kpts, int32_t K, const double* radius,
                          int32_t P, float* patches_out, int32_t* count_out)
{
    BX_ENTER(c, false);
    if (!pts_perm || !kpts || !radius || !patches_out || !count_out) { bx_set_error("bx_ball_group_counted: null argument"); return BX_ERR_ARG; }
    c->skip = nullptr;
    c->ball_waves_hint = 0;
    return bxk_ball_group(c, (hipStream_t)stream, pts_perm, n, kpts, K, radius, P, nullptr, patches_out, count_out);
}

int bx_patch_features_counted(bx_ctx* c, void* stream, const float* patches, const int32_t* counts, const float* kpts, int32_t K, int32_t P,
                              const double* radius, int32_t aligned_z, float* R_out, float* feat_out)
{
    BX_ENTER(c, true);
    if (!patches || !counts || !kpts || !radius || !R_out || !feat_out) { bx_set_error("bx_patch_features_counted: null argument"); return BX_ERR_ARG; }
    c->skip = nullptr;
    return bxk_patch_features(c, (hipStream_t)stream, patches, K, P, radius, aligned_z, R_out, feat_out, counts, kpts);
}

int bx_patch_features(bx_ctx* c, void* stream, const float* patches, int32_t K, int32_t P, const double* radius, int32_t aligned_z,
                      float* R_out, float* feat_out)
{
    BX_ENTER(c, true);
    if (!patches || !radius || !R_out || !feat_out) { bx_set_error("bx_patch_features: null argument"); return BX_ERR_ARG; }
    c->skip = nullptr;
    return bxk_patch_features(c, (hipStream_t)stream, patches, K, P, radius, aligned_z, R_out, feat_out);
}

int bx_desc_net(bx_ctx* c, void* stream, const float* feat, int32_t K, float* desc_out, float* equi_out, float* x_out)
{
    BX_ENTER(c, true);
    if (K > c->p.num_fps) { bx_set_error("bx_desc_net: K=%d exceeds context num_fps=%d", K, c->p.num_fps); return BX_ERR_ARG; }
    c->skip = nullptr;
    return desc_stack(c, (hipStream_t)stream, feat, K, desc_out, equi_out, x_out);
}

int bx_conv_layer(bx_ctx* c, void* stream, int32_t net, int32_t layer, const float* in, int32_t units, float* out)
{
    BX_ENTER(c, true);
    if (net == 1 && layer == 0) { bx_set_error("bx_conv_layer: Pose layer 0 consumes the implicit cost volume; use bx_pose_net"); return BX_ERR_ARG; }
    c->skip = nullptr;
    return bxk_conv(c, (hipStream_t)stream, net, layer, in, nullptr, units, out);
}

int bx_mutual(bx_ctx* c, void* stream, const float* src_des, int32_t ns, const float* tgt_des, int32_t nt, int32_t* s_mids,
              int32_t* t_mids, int32_t* count_out)
{
    BX_ENTER(c, false);
    if (ns > c->p.num_fps || nt > c->p.num_fps) { bx_set_error("bx_mutual: more descriptors than num_fps"); return BX_ERR_ARG; }
    c->skip = nullptr;
    return bxk_mutual(c, (hipStream_t)stream, src_des, ns, tgt_des, nt, s_mids, t_mids, count_out);
}

int bx_pose_net(bx_ctx* c, void* stream, const float* s_equi, const float* t_equi, const int32_t* s_mids, const int32_t* t_mids,
                const int32_t* m_dev, int32_t max_m, float* ind_out, float* logits_out)
{
    BX_ENTER(c, true);
    if (max_m > c->p.num_fps) { bx_set_error("bx_pose_net: max_m exceeds num_fps"); return BX_ERR_ARG; }
    c->skip = nullptr;
    return pose_stack(c, (hipStream_t)stream, s_equi, t_equi, s_mids, t_mids, m_dev, max_m, ind_out, logits_out);
}

int bx_hypotheses(bx_ctx* c, void* stream, const float* ind, const int32_t* s_mids, const int32_t* t_mids, const int32_t* m_dev,
                  int32_t max_m, const float* s_R, const float* t_R, const float* s_kpts, const float* t_kpts, float* R_out,
                  float* t_out, float* ss_out, float* tt_out)
{
    BX_ENTER(c, false);
    return bxk_hypotheses((hipStream_t)stream, ind, s_mids, t_mids, m_dev, max_m, s_R, t_R, s_kpts, t_kpts, R_out, t_out, ss_out,
                          tt_out, nullptr, nullptr);
}

int bx_consensus(bx_ctx* c, void* stream, const float* R, const float* t, const float* ss, const float* tt, const int32_t* M_dev,
                 int32_t max_M, int32_t* inlier_out, int32_t* count_out, int32_t* best_out)
{
    BX_ENTER(c, false);
    if (max_M > c->p.num_fps * c->p.num_scales) { bx_set_error("bx_consensus: max_M exceeds num_fps*num_scales"); return BX_ERR_ARG; }
    c->skip = nullptr;
    return bxk_consensus(c, (hipStream_t)stream, R, t, ss, tt, M_dev, max_M, inlier_out, count_out, best_out);
}

int bx_ransac(bx_ctx* c, void* stream, const float* ss, const float* tt, const int32_t* corr, const int32_t* C_dev, int32_t max_C,
              uint64_t seed, double* T_out, int32_t* info_out)
{
    BX_ENTER(c, false);
    return bxk_ransac(c, (hipStream_t)stream, ss, tt, corr, C_dev, max_C, seed, T_out, info_out, nullptr);
}

int bx_kiss_solve(bx_ctx* c, void* stream, const float* ss, const float* tt, const int32_t* corr, const int32_t* C_dev, int32_t max_C,
                  double* T_out, int32_t* info_out)
{
    BX_ENTER(c, false);
    if (!ss || !tt || !corr || !C_dev) { bx_set_error("bx_kiss_solve: null argument"); return BX_ERR_ARG; }
    return bxk_kiss(c, (hipStream_t)stream, ss, tt, corr, C_dev, max_C, T_out, info_out, nullptr);
}

int bx_refine(bx_ctx* c, void* stream, const float* ss, const float* tt, const int32_t* M_dev, int32_t max_M, float* T_io,
              int32_t* iters_out)
{
    BX_ENTER(c, false);
    if (max_M > c->p.num_fps * c->p.num_scales) { bx_set_error("bx_refine: max_M exceeds num_fps*num_scales"); return BX_ERR_ARG; }
    return bxk_refine(c, (hipStream_t)stream, ss, tt, M_dev, max_M, T_io, iters_out);
}

// ------------------------------------------------------------------------------------------------ pre-processing
int bx_pre_reserve(bx_ctx* c, int64_t max_points)
{
    BX_ENTER(c, false);
    return bxk_pre_reserve(c, max_points);
}

int bx_pre_voxel_downsample(bx_ctx* c, void* stream, const float* pts, int32_t n, double voxel_size, float* out, int32_t* count_out)
{
    BX_ENTER(c, false);
    if (!pts || !out || !count_out) { bx_set_error("bx_pre_voxel_downsample: null argument"); return BX_ERR_ARG; }
    return bxk_pre_voxel_downsample(c, (hipStream_t)stream, pts, n, voxel_size, out, count_out);
}

int bx_random_perm(bx_ctx* c, void* stream, int32_t n, uint64_t seed, int32_t* out)
{
    BX_ENTER(c, false);
    if (!out || n < 0 || n > (1 << 30)) { bx_set_error("bx_random_perm: bad argument"); return BX_ERR_ARG; }
    return bxk_random_perm((hipStream_t)stream, n, seed, out);
}

int bx_pre_pca(bx_ctx* c, void* stream, const float* pts, int32_t n, const int32_t* sample_idx, int32_t ns, double* out17)
{
    BX_ENTER(c, false);
    if (!pts || !sample_idx || !out17) { bx_set_error("bx_pre_pca: null argument"); return BX_ERR_ARG; }
    return bxk_pre_pca(c, (hipStream_t)stream, pts, n, sample_idx, ns, out17);
}

// ------------------------------------------------------------------------------------------------ lane
int bx_lane_create(int32_t mode, bx_lane** out)
{
    if (!out || mode < 1 || mode > 2) { bx_set_error("bx_lane_create: mode must be 1 or 2"); return BX_ERR_ARG; }
    bx_lane* l = new bx_lane();
    for (int i = 0; i < bx_lane::NEV; ++i) {
        hipError_t e = hipEventCreateWithFlags(&l->ev[i], hipEventDisableTiming);
        if (e != hipSuccess) { bx_set_error("bx_lane_create: hipEventCreate failed: %s", hipGetErrorString(e)); delete l; return BX_ERR_HIP; }
    }
    l->next = 0; l->last = -1; l->mode = mode;
    *out = l;
    return BX_OK;
}

int bx_lane_destroy(bx_lane* l)
{
    if (!l) return BX_OK;
    for (int i = 0; i < bx_lane::NEV; ++i) (void)hipEventDestroy(l->ev[i]);
    delete l;
    return BX_OK;
}

int bx_attach_lane(bx_ctx* c, bx_lane* lane)
{
    if (!c) { bx_set_error("bx_attach_lane: null context"); return BX_ERR_ARG; }
    c->lane = lane;
    return BX_OK;
}

namespace {
struct LaneScope {   // section of a pair that takes its turn on the lane
    bx_lane* l; hipStream_t s;
    LaneScope(bx_ctx* c, hipStream_t st, int mode) : l(c->lane && c->lane->mode == mode ? c->lane : nullptr), s(st)
    {
        if (l && l->last >= 0) (void)hipStreamWaitEvent(s, l->ev[l->last], 0);
    }
    ~LaneScope()
    {
        if (!l) return;
        (void)hipEventRecord(l->ev[l->next], s);
        l->last = l->next;
        l->next = (l->next + 1) % bx_lane::NEV;
    }
};
}  // namespace

// ------------------------------------------------------------------------------------------------ whole pair
}  // extern "C"

// phase 0: the whole pair (bx_register_pair).  phase 1: up to and including the exit test of scale 0 -- every scale when no early exit can
// follow -- and the copy of the decision to the host (bx_register_pair_begin).  phase 2: the rest, the host knowing whether the pair
// left (bx_register_pair_finish).  One body: the two-call form enqueues the very launches of the one-call form, minus those a pair that
// left would only have returned from.
static int register_pair_impl(bx_ctx* c, void* stream, const float* src, int32_t n_src, const float* tgt, int32_t n_tgt, int32_t aligned_z,
                              const int32_t* perm_src, const int32_t* perm_tgt, uint64_t seed, bx_result* result, int phase, bool exited,
                              int32_t* exited_host)
{
    BX_ENTER(c, true);
    if (!src || !tgt || !perm_src || !perm_tgt || (!result && phase != 1)) { bx_set_error("bx_register_pair: null argument"); return BX_ERR_ARG; }
    const bx_params& p = c->p;
    if (n_src < 1 || n_tgt < 1 || n_src > p.max_points || n_tgt > p.max_points) {
        bx_set_error("bx_register_pair: cloud sizes %d/%d outside [1, max_points=%d]", n_src, n_tgt, p.max_points);
        return BX_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int K = p.num_fps, P = p.num_points_per_patch, S = p.num_scales, NK = p.num_points_radius_estimate;
    const int KM = K > NK ? K : NK;
    PairState* st = c->state;
    c->skip = nullptr;
    const bool early = p.enable_early_exit != 0;
    if (phase != 0 && p.keypoint_tiles > 1) { bx_set_error("bx_register_pair_begin / _finish: throughput form only (keypoint_tiles <= 1)"); return BX_ERR_STATE; }
    int ransac_calls = phase == 2 ? c->pend.ransac_calls : 0;
    const float* clouds[2] = {src, tgt};
    const int ns[2] = {n_src, n_tgt};
    const int32_t* perms[2] = {perm_src, perm_tgt};
    int32_t tb[BX_MAX_TILES + 1] = {0};
    const int T = bx_keypoint_tile_bounds(&p, tb);
    if (T == 1) tb[1] = KM;                       // one launch covers the radius-estimation prefix too
    const bool tiled = T > 1;                       // FPS in several launches on the context's own stream
    const bool multi = p.keypoint_tiles > 1;        // source / target / matching chains on the context's streams (also when K <= nk
                                                    // leaves nothing to tile: the reference's default num_fps = 1500 < 2000)
    if (phase != 2) {
    hipLaunchKernelGGL(state_reset_kernel, dim3(1), dim3(64), 0, s, st, c->err_flag);

    // (1) keypoints: ONE furthest-point sampling per cloud; FPS(nk) is a prefix of FPS(K).
    // Latency form (params.keypoint_tiles > 1): the run is cut into tiles of keypoints on the context's own stream; tile 0 ends
    // where the radius estimation has its keypoints, and the descriptor work of a tile runs on the caller's stream while the next
    // tile is still being sampled (a descriptor depends on its own keypoint only).
    if (multi && c->cap_on) { bx_set_error("bx_register_pair: bx_set_capture needs keypoint_tiles <= 1"); return BX_ERR_STATE; }
    hipStream_t fs = tiled ? c->aux_stream : s;
    if (tiled) {
        BX_HIP(hipEventRecord(c->ev_fork, s));
        BX_HIP(hipStreamWaitEvent(fs, c->ev_fork, 0));
    }
    for (int t = 0; t < T; ++t) {
        { ProfScope ps(c, fs, 0); if ((rc = bxk_fps_range(c, fs, clouds, ns, 2, tb[t], tb[t + 1], KM, c->fps_idx, c->kpts)) != BX_OK) return rc; }
        if (tiled) BX_HIP(hipEventRecord(c->ev_tile[t], fs));
    }
    if (tiled) BX_HIP(hipStreamWaitEvent(s, c->ev_tile[0], 0));
    }   // phase != 2

    LaneScope lane_main(c, s, 1);
    if (phase != 2) {
    // (2) radius estimation histogram: the LARGER cloud and its keypoints (models/BUFFERX.py:654-665), once per pair
    const int big = n_src > n_tgt ? 0 : 1;
    const float* rpts = clouds[big];
    int rn = ns[big];
    if (rn > 200000) {
        hipLaunchKernelGGL(subsample_kernel, dim3((200000 + 255) / 256), dim3(256), 0, s, clouds[big], rn, (unsigned long long)seed, 200000, c->sub_pts);
        rpts = c->sub_pts;
        rn = 200000;
    }
    { ProfScope ps(c, s, 1); if ((rc = bxk_radius_hist(c, s, rpts, rn, c->kpts[big], NK)) != BX_OK) return rc; }

    // every scale's radius up front (the bisections share the histogram: one launch), then the grids of all 2 x S (cloud, scale)
    // sets in one batch of six launches; the per-scale permutation is applied on the fly
    { ProfScope ps(c, s, 1); if ((rc = bxk_radius_bisect_all(c, s, (int64_t)ns[big], NK, p.search_radius_thresholds, S, st->des_r)) != BX_OK) return rc; }
    // (with the early exit on, only scale 0's two grids: the later scales' are built behind the exit test and skipped with the pair)
    c->skip = nullptr;
    { ProfScope ps(c, s, 13); if ((rc = bxk_ball_grids(c, s, clouds, ns, perms, 2, st->des_r, S, p.search_radius_thresholds, 0, early ? 1 : S)) != BX_OK) return rc; }
    }   // phase != 2

    // descriptors of keypoints [k0, k0 + kn) of one (scale, cloud): neighbour gather -> patch features -> Cylindrical_Net
    // Latency form: the target cloud's chain runs on the context's second stream with its own scratch, so that the tail of one
    // chain's launch is filled by the other's (the same overlap several pairs in flight give the throughput form).
    auto describe = [&](int i, int cl, int k0, int kn) -> int {
        const bool capc = c->cap_on && c->cap.scale == i && c->cap.cloud == cl;
        const bool side = multi && cl == 1;
        hipStream_t ds = side ? c->tgt_stream : s;
        float* patches = side ? c->patches2 : c->patches;
        // hit-count hand-over: the query kernel writes only the real slots of a patch and their number, the two patch kernels take the
        // count (k_ball.hip / k_patch.hip; bit-identical features).  A captured (scale, cloud) keeps the padded form: the capture
        // buffer holds the reference's [K][P][3] tensor.
        int32_t* pcnt = capc ? nullptr : (side ? c->pcnt2 : c->pcnt);
        float* feat = side ? c->feat2 : c->feat;
        // expected neighbourhood = threshold % of the cloud: large ones get 4 waves per keypoint, small ones 2 (measured)
        c->ball_waves_hint = p.search_radius_thresholds[i] >= 1.5 ? 4 : 2;
        // the whole-pair path does not need the ball_query index list (nothing downstream reads it): idx_out = nullptr
        { ProfScope ps(c, ds, 2); if ((rc = bxk_ball_query(c, ds, cl * S + i, ns[cl], c->kpts[cl], k0, kn, &st->des_r[i], P, nullptr, patches, pcnt)) != BX_OK) return rc; }
        if (capc && c->cap.pts_perm) {      // the permuted cloud is never materialised on the hot path
            if ((rc = bx_permute_launch(ds, clouds[cl], perms[cl] + (size_t)i * ns[cl], ns[cl], c->pts_perm, nullptr)) != BX_OK) return rc;
        }
        { ProfScope ps(c, ds, 3); if ((rc = bxk_patch_features(c, ds, patches, kn, P, &st->des_r[i], aligned_z, c->R_sc[i][cl] + (size_t)k0 * 9, feat, pcnt, pcnt ? c->kpts[cl] + (size_t)k0 * 3 : nullptr)) != BX_OK) return rc; }
        if (capc) {
            if ((rc = cap_copy(ds, c->cap.pts_perm, c->pts_perm, (size_t)ns[cl] * 3)) != BX_OK) return rc;
            if ((rc = cap_copy(ds, c->cap.patches, patches, (size_t)K * P * 3)) != BX_OK) return rc;
            if ((rc = cap_copy(ds, c->cap.feat, feat, (size_t)K * BX_RAD * BX_EA * 16)) != BX_OK) return rc;
        }
        if (multi) {
            ProfScope ps(c, ds, 4);
            if ((rc = desc_stack(c, ds, feat, kn, c->desc_sc[i][cl] + (size_t)k0 * 32, c->equi_sc[i][cl] + (size_t)k0 * BX_EA * 32, nullptr, side ? c->act2 : c->act3)) != BX_OK) return rc;
            if (k0 + kn == K) BX_HIP(hipEventRecord(c->ev_desc[cl][i], ds));     // this (cloud, scale) is complete
        } else {
            LaneScope ls(c, ds, 2); ProfScope ps(c, ds, 4);
            if ((rc = desc_stack(c, ds, feat, kn, c->desc_sc[i][cl] + (size_t)k0 * 32, c->equi_sc[i][cl] + (size_t)k0 * BX_EA * 32, capc ? c->cap.x : nullptr)) != BX_OK) return rc;
        }
        return BX_OK;
    };
    // the target chain starts behind whatever the caller's stream has enqueued so far / the caller's stream waits for it
    auto tgt_go = [&]() -> int {
        if (!multi) return BX_OK;
        BX_HIP(hipEventRecord(c->ev_tgt_go, s));
        BX_HIP(hipStreamWaitEvent(c->tgt_stream, c->ev_tgt_go, 0));
        return BX_OK;
    };
    auto tgt_join = [&]() -> int {
        if (!multi) return BX_OK;
        BX_HIP(hipEventRecord(c->ev_tgt_done, c->tgt_stream));
        BX_HIP(hipStreamWaitEvent(s, c->ev_tgt_done, 0));
        return BX_OK;
    };
    // tile by tile: candidate tables of the tile (all sets, one launch), then its descriptors -- of every scale when no early exit
    // can skip the later ones, else of scale 0 only (the later scales then run over all keypoints behind the exit test)
    c->skip = nullptr;
    for (int t = 0; t < T && phase != 2; ++t) {
        const int k0 = tb[t], kn = (t == T - 1 ? K : tb[t + 1]) - k0;
        if (tiled && t > 0) BX_HIP(hipStreamWaitEvent(s, c->ev_tile[t], 0));
        { ProfScope ps(c, s, 13); if ((rc = bxk_ball_rows(c, s, c->kpts, 2, S, k0, kn, 0, early ? 1 : S)) != BX_OK) return rc; }
        if (!multi) break;
        if ((rc = tgt_go()) != BX_OK) return rc;
        for (int i = 0; i < (early ? 1 : S); ++i)
            for (int cl = 0; cl < 2; ++cl)
                if ((rc = describe(i, cl, k0, kn)) != BX_OK) return rc;
    }
    // no early exit: nothing of a scale's matching decides what the later scales do -- it runs on the third stream, beside the
    // descriptor work still queued on the other two, and only the last scale's matching is exposed
    hipStream_t s_caller = s;
    const bool split_match = multi && !early;
    if (!split_match) { if ((rc = tgt_join()) != BX_OK) return rc; }
    if (split_match) s = c->match_stream;
    // (second call: scale 0 is done; a pair that left has nothing more to describe)
    for (int i = phase == 2 ? (early ? (exited ? S : 1) : S) : 0; i < S; ++i) {
        if (split_match) {
            BX_HIP(hipStreamWaitEvent(s, c->ev_desc[0][i], 0));
            BX_HIP(hipStreamWaitEvent(s, c->ev_desc[1][i], 0));
        }
        c->skip = (early && i > 0) ? &st->done : nullptr;
        const bool capi = c->cap_on && c->cap.scale == i;
        if (early && i == 1) {
            // the grids and candidate tables of scales 1 .. S - 1, behind the exit test of scale 0 (nothing is built for a pair that left)
            ProfScope ps(c, s, 13);
            if ((rc = bxk_ball_grids(c, s, clouds, ns, perms, 2, st->des_r, S, p.search_radius_thresholds, 1, S - 1)) != BX_OK) return rc;
            if ((rc = bxk_ball_rows(c, s, c->kpts, 2, S, 0, K, 1, S - 1)) != BX_OK) return rc;
        }
        // (2 K units must still fit the 32-bit byte offsets of the F(4x4) kernels: beyond ~14 900 keypoints the clouds run one by one)
        const bool batch_fits = (long long)2 * K * 8 * BX_EA * 64 < 0x7fffffffLL - (1LL << 24);
        if (!multi && c->desc_batch && batch_fits && !(c->cap_on && c->cap.scale == i)) {
            // throughput form: neighbour gather + patch features per cloud, then BOTH clouds' Cylindrical_Net stacks as one launch per layer
            for (int cl = 0; cl < 2; ++cl) {
                c->ball_waves_hint = p.search_radius_thresholds[i] >= 1.5 ? 4 : 2;
                { ProfScope ps(c, s, 2); if ((rc = bxk_ball_query(c, s, cl * S + i, ns[cl], c->kpts[cl], 0, K, &st->des_r[i], P, nullptr, c->patches, c->pcnt)) != BX_OK) return rc; }
                { ProfScope ps(c, s, 3); if ((rc = bxk_patch_features(c, s, c->patches, K, P, &st->des_r[i], aligned_z, c->R_sc[i][cl], c->feat + (size_t)cl * K * BX_RAD * BX_EA * 16, c->pcnt, c->kpts[cl])) != BX_OK) return rc; }
            }
            LaneScope ls(c, s, 2); ProfScope ps(c, s, 4);      // ONE bracket = one launch sequence of 2 K units (bench.py prices it as such)
            if ((rc = desc_stack_pair(c, s, c->feat, K, c->desc_sc[i], c->equi_sc[i])) != BX_OK) return rc;
        } else if (!multi || (early && i > 0)) {
            if ((rc = tgt_go()) != BX_OK) return rc;
            for (int cl = 0; cl < 2; ++cl)
                if ((rc = describe(i, cl, 0, K)) != BX_OK) return rc;
            if ((rc = tgt_join()) != BX_OK) return rc;
        }
        float* const* dsc = c->desc_sc[i];
        float* const* eqv = c->equi_sc[i];
        float* const* Rp = c->R_sc[i];
        if (capi) {
            for (int cl = 0; cl < 2; ++cl) {
                if ((rc = cap_copy(s, c->cap.kpts[cl], c->kpts[cl], (size_t)K * 3)) != BX_OK) return rc;
                if ((rc = cap_copy(s, c->cap.desc[cl], dsc[cl], (size_t)K * 32)) != BX_OK) return rc;
                if ((rc = cap_copy(s, c->cap.equi[cl], eqv[cl], (size_t)K * BX_EA * 32)) != BX_OK) return rc;
                if ((rc = cap_copy(s, c->cap.R[cl], Rp[cl], (size_t)K * 9)) != BX_OK) return rc;
            }
        }
        { ProfScope ps(c, s, 6); if ((rc = bxk_mutual(c, s, dsc[0], K, dsc[1], K, c->s_mids, c->t_mids, &st->m_scale)) != BX_OK) return rc; }
        { LaneScope ls(c, s, 2); ProfScope ps(c, s, 7); if ((rc = pose_stack(c, s, eqv[0], eqv[1], c->s_mids, c->t_mids, &st->m_scale, K, c->ind, nullptr)) != BX_OK) return rc; }
        if ((rc = bxk_hypotheses(s, c->ind, c->s_mids, c->t_mids, &st->m_scale, K, Rp[0], Rp[1], c->kpts[0], c->kpts[1],
                                 c->R_cat, c->t_cat, c->ss_cat, c->tt_cat, &st->M, c->skip)) != BX_OK) return rc;
        hipLaunchKernelGGL(accumulate_kernel, dim3(1), dim3(64), 0, s, st, i, c->skip);
        { ProfScope ps(c, s, 8); if ((rc = bxk_consensus(c, s, c->R_cat, c->t_cat, c->ss_cat, c->tt_cat, &st->M, (i + 1) * K, c->inlier_ind, &st->C, &st->best)) != BX_OK) return rc; }
        if (capi) {
            const size_t MK = (size_t)(i + 1) * K;
            if ((rc = cap_copy(s, c->cap.s_mids, c->s_mids, (size_t)K)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.t_mids, c->t_mids, (size_t)K)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.ind, c->ind, (size_t)K)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.R_cat, c->R_cat, MK * 9)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.t_cat, c->t_cat, MK * 3)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.ss_cat, c->ss_cat, MK * 3)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.tt_cat, c->tt_cat, MK * 3)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.cons_cnt, c->cons_cnt, MK)) != BX_OK) return rc;
            if ((rc = cap_copy(s, c->cap.inlier_ind, c->inlier_ind, MK)) != BX_OK) return rc;
            if (c->cap.counts) hipLaunchKernelGGL(cap_counts_kernel, dim3(1), dim3(64), 0, s, st, c->cap.counts, (double*)nullptr);
        }
        if (early && i == 0) {
            if (p.pose_estimator == 1) rc = bxk_kiss(c, s, c->ss_cat, c->tt_cat, c->inlier_ind, &st->C, K, nullptr, nullptr, nullptr);
            else rc = bxk_ransac(c, s, c->ss_cat, c->tt_cat, c->inlier_ind, &st->C, K, bx_mix64(seed, 0x5AC0000ULL + ransac_calls), nullptr, nullptr, nullptr);
            if (rc != BX_OK) return rc;
            ++ransac_calls;
            hipLaunchKernelGGL(early_exit_kernel, dim3(1), dim3(64), 0, s, st, p.early_exit_min_inliers);
            if (phase == 1 && S > 1) break;      // the host decides about the later scales
        }
    }
    if (phase == 1) {
        // the decision, stream-ordered (0 when nothing can be skipped: the second call then only finishes the pair)
        if (early && S > 1) BX_HIP(hipMemcpyAsync(exited_host, &st->done, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        else *exited_host = 0;
        BX_LAUNCH_CHECK();
        c->pend = {true, src, tgt, n_src, n_tgt, aligned_z, perm_src, perm_tgt, seed, ransac_calls};
        return BX_OK;
    }
    c->pend.active = false;
    if (split_match) {
        BX_HIP(hipEventRecord(c->ev_match_done, s));
        s = s_caller;
        BX_HIP(hipStreamWaitEvent(s, c->ev_match_done, 0));
        if ((rc = tgt_join()) != BX_OK) return rc;
    }
    // final pose estimation unless the early exit was taken (models/BUFFERX.py:449-457)
    if (!(phase == 2 && early && exited)) { ProfScope ps(c, s, 9);
    if (p.pose_estimator == 1) rc = bxk_kiss(c, s, c->ss_cat, c->tt_cat, c->inlier_ind, &st->C, S * K, nullptr, nullptr, early ? &st->done : nullptr);
    else rc = bxk_ransac(c, s, c->ss_cat, c->tt_cat, c->inlier_ind, &st->C, S * K, bx_mix64(seed, 0x5AC0000ULL + ransac_calls), nullptr, nullptr,
                         early ? &st->done : nullptr);
    if (rc != BX_OK) return rc; }
    c->skip = nullptr;
    if (c->cap_on && c->cap.T_ransac) hipLaunchKernelGGL(cap_counts_kernel, dim3(1), dim3(64), 0, s, st, (int32_t*)nullptr, c->cap.T_ransac);
    if (p.pose_refine) {
        hipLaunchKernelGGL(pose_to_float_kernel, dim3(1), dim3(64), 0, s, st);
        { ProfScope ps(c, s, 10); if ((rc = bxk_refine(c, s, c->ss_cat, c->tt_cat, &st->M, S * K, st->Tf, &st->refine_iters)) != BX_OK) return rc; }
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, s, st, c->err_flag, p.pose_refine, S,
                       p.desc_conv_form | (p.pose_conv_form << 8) | (p.cost_l0_form << 16), c->result_dev);
    BX_LAUNCH_CHECK();
    BX_HIP(hipMemcpyAsync(result, c->result_dev, sizeof(bx_result), hipMemcpyDeviceToHost, s));
    return BX_OK;
}

extern "C" {

int bx_register_pair(bx_ctx* c, void* stream, const float* src, int32_t n_src, const float* tgt, int32_t n_tgt, int32_t aligned_z,
                     const int32_t* perm_src, const int32_t* perm_tgt, uint64_t seed, bx_result* result)
{
    return register_pair_impl(c, stream, src, n_src, tgt, n_tgt, aligned_z, perm_src, perm_tgt, seed, result, 0, false, nullptr);
}

int bx_register_pair_begin(bx_ctx* c, void* stream, const float* src, int32_t n_src, const float* tgt, int32_t n_tgt, int32_t aligned_z,
                           const int32_t* perm_src, const int32_t* perm_tgt, uint64_t seed, int32_t* exited_host)
{
    if (!exited_host) { bx_set_error("bx_register_pair_begin: null argument"); return BX_ERR_ARG; }
    return register_pair_impl(c, stream, src, n_src, tgt, n_tgt, aligned_z, perm_src, perm_tgt, seed, nullptr, 1, false, exited_host);
}

int bx_register_pair_finish(bx_ctx* c, void* stream, int32_t exited, bx_result* result)
{
    if (!c || !c->pend.active) { bx_set_error("bx_register_pair_finish: no pending bx_register_pair_begin on this context"); return BX_ERR_STATE; }
    const bx_ctx::PendingPair a = c->pend;
    return register_pair_impl(c, stream, a.src, a.n_src, a.tgt, a.n_tgt, a.aligned_z, a.perm_src, a.perm_tgt, a.seed, result, 2, exited != 0, nullptr);
}

}  // extern "C"
