// k_patch.hip -- patch -> cylindrical voxel features, one workgroup per patch, fused:
//   axis_align   (reference models/patch_embedder.py:122-148; utils/common.py:709-726 cal_Z_axis,
//                 :501-525 RodsRotatFormula, :111-114 l2_norm)
//   normalize    (models/patch_embedder.py:167-170)
//   SPT          (models/patch_embedder.py:150-165; utils/common.py:431-469 sphere_query, :472-498 var_to_invar)
//   pnt_layer + max over the voxel samples (models/patch_embedder.py:26-30, 73-77)
// The reference materialises [K,P,3] x4 temporaries, a [K,420,10,3] gather, the constant voxel grid and 20
// rotation matrices per call; here the patch lives in LDS (16 B/point), the 3x3 covariance is a wave
// reduction (xor-butterfly, the arithmetic contract's "wave order"), the eigenvector comes from a binary64
// Jacobi on one wave, and each thread owns voxels: it scans the patch in order with LDS broadcast reads,
// keeps the first `voxel_sample` hits, and applies mask, azimuth de-rotation, 3->16 conv + ReLU and the max
// in registers.  Output: feat [K][rad][ele*azi][16] in chunk-slot order (bx_chunk_slot).
#include "bx_common.h"

namespace {
constexpr int PF_THREADS = 256;
constexpr int MAX_NS = 16;

__global__ __launch_bounds__(PF_THREADS) void patch_features_kernel(
    const float* __restrict__ patches, int K, int P, const double* __restrict__ radius, int aligned,
    const float* __restrict__ centres, const float* __restrict__ rot, int nsample, float voxel_r,
    const float* __restrict__ pnt_w, const float* __restrict__ pnt_b, float* __restrict__ R_out, float* __restrict__ feat,
    const int32_t* __restrict__ skip)
{
    if (skip && *skip) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sp = reinterpret_cast<float4*>(smem);                            // [P]
    unsigned short* shit = reinterpret_cast<unsigned short*>(sp + P);       // [nsample][BX_VOX]
    float* sR = reinterpret_cast<float*>(shit + (size_t)MAX_NS * BX_VOX);   // [9] (+pad)

    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pp = patches + (size_t)q * P * 3;
    const float des_r = (float)(*radius);
    const float cx = pp[(size_t)(P - 1) * 3], cy = pp[(size_t)(P - 1) * 3 + 1], cz = pp[(size_t)(P - 1) * 3 + 2];

    for (int i = tid; i < P; i += PF_THREADS) {
        float x = pp[(size_t)i * 3] - cx, y = pp[(size_t)i * 3 + 1] - cy, z = pp[(size_t)i * 3 + 2] - cz;
        sp[i] = make_float4(x, y, z, 0.f);
    }
    __syncthreads();

    if (!aligned) {
        if (wave == 0) {
            float c00 = 0.f, c01 = 0.f, c02 = 0.f, c11 = 0.f, c12 = 0.f, c22 = 0.f;
            for (int i = lane; i < P; i += 64) {
                float4 d = sp[i];
                c00 = fmaf(d.x, d.x, c00); c01 = fmaf(d.x, d.y, c01); c02 = fmaf(d.x, d.z, c02);
                c11 = fmaf(d.y, d.y, c11); c12 = fmaf(d.y, d.z, c12); c22 = fmaf(d.z, d.z, c22);
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
                        sR[j * 3 + i] = rr;  // transpose(-1,-2)
                    }
            }
        }
        __syncthreads();
        float R[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = sR[i];
        for (int i = tid; i < P; i += PF_THREADS) {
            float4 d = sp[i];
            float nx = fmaf(d.z, R[6], fmaf(d.y, R[3], d.x * R[0]));
            float ny = fmaf(d.z, R[7], fmaf(d.y, R[4], d.x * R[1]));
            float nzc = fmaf(d.z, R[8], fmaf(d.y, R[5], d.x * R[2]));
            sp[i] = make_float4(nx / des_r, ny / des_r, nzc / des_r, 0.f);
        }
        if (tid < 9) R_out[(size_t)q * 9 + tid] = sR[tid];
    } else {
        for (int i = tid; i < P; i += PF_THREADS) {
            float4 d = sp[i];
            sp[i] = make_float4(d.x / des_r, d.y / des_r, d.z / des_r, 0.f);
        }
        if (tid < 9) R_out[(size_t)q * 9 + tid] = (tid % 4 == 0) ? 1.0f : 0.0f;
    }
    __syncthreads();

    const float vr2 = voxel_r * voxel_r;
    for (int v0 = 0; v0 < BX_VOX; v0 += PF_THREADS) {
        const int v = v0 + tid;
        const bool act = v < BX_VOX;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (act) { qx = centres[v * 3]; qy = centres[v * 3 + 1]; qz = centres[v * 3 + 2]; }
        int cnt = act ? 0 : nsample;
        for (int k0 = 0; k0 < P; k0 += 8) {
            if (__all(cnt >= nsample)) break;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                int k = k0 + kk;
                if (k < P) {
                    float4 d = sp[k];
                    float dx = qx - d.x, dy = qy - d.y, dz = qz - d.z;
                    float dd = (dx * dx + dy * dy) + dz * dz;
                    if (dd < vr2 && cnt < nsample) {
                        shit[cnt * BX_VOX + v] = (unsigned short)k;
                        ++cnt;
                    }
                }
            }
        }
        if (!act) continue;
        const int a = v % BX_AZI;
        const float r00 = rot[a * 4], r01 = rot[a * 4 + 1], r10 = rot[a * 4 + 2], r11 = rot[a * 4 + 3];
        const int first = cnt > 0 ? (int)shit[v] : 0;
        float mx[16];
        for (int j = 0; j < nsample; ++j) {
            int id = j < cnt ? (int)shit[j * BX_VOX + v] : first;
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
                acc = acc > 0.0f ? acc : 0.0f;
                mx[c] = (j == 0 || acc > mx[c]) ? acc : mx[c];
            }
        }
        const int s = v / BX_EA, pos = v % BX_EA;
        float4* fo = reinterpret_cast<float4*>(feat + (((size_t)q * BX_RAD + s) * BX_EA + pos) * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) fo[u] = make_float4(mx[u], mx[4 + u], mx[8 + u], mx[12 + u]);
    }
}
}  // namespace

int bxk_patch_features(bx_ctx* c, hipStream_t s, const float* patches, int K, int P, const double* radius, int aligned,
                       float* R_out, float* feat_out)
{
    if (K <= 0) return BX_OK;
    const int ns = c->p.voxel_sample;
    if (ns < 1 || ns > MAX_NS || P < 2 || P > 8192) { bx_set_error("bxk_patch_features: voxel_sample=%d P=%d unsupported", ns, P); return BX_ERR_ARG; }
    size_t lds = (size_t)P * 16 + (size_t)MAX_NS * BX_VOX * 2 + 64;
    const float voxel_r = (float)(c->p.delta / (double)c->p.rad_n);
    hipLaunchKernelGGL(patch_features_kernel, dim3(K), dim3(PF_THREADS), lds, s, patches, K, P, radius, aligned, c->d_centres,
                       c->d_rot, ns, voxel_r, c->d_pnt_w, c->d_pnt_b, R_out, feat_out, c->skip);
    BX_LAUNCH_CHECK();
    return BX_OK;
}
