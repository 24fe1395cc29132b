/*
 * oracle/bxo_detmath.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Deterministic elementary functions and tiny linear-algebra helpers used by the CPU oracle.
 *
 * Why they exist: the reference evaluates acos/sin/cos (utils/common.py:501-525 RodsRotatFormula),
 * cos/sin (kornia axis_angle_to_rotation_matrix, call site models/BUFFERX.py:386), exp (softmax,
 * models/BUFFERX.py:67) and log (Open3D RANSAC termination, call site models/pose_estimator.py:94-112)
 * through whatever libm / CUDA libdevice the host happens to have, i.e. their last-ulp behaviour is
 * *unspecified* by the reference.  To make "oracle == HIP kernel" a bit-exact statement, both sides
 * evaluate these functions with IEEE-754 binary64 +,-,*,/,sqrt only (no libm, no FMA contraction),
 * using the fixed series below, and round the result to binary32 where the reference works in fp32.
 * The result is within 1 ulp(fp32) of a correctly rounded libm result (checked in
 * tests/test_oracle_math.py against numpy).
 *
 * Compile with -ffp-contract=off.
 */
#ifndef BXO_DETMATH_H
#define BXO_DETMATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline double bxo_u64_as_f64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t bxo_f64_as_u64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

/* 2^k for integer k in [-1022, 1023] */
static inline double bxo_pow2i(int k) { return bxo_u64_as_f64((uint64_t)(k + 1023) << 52); }

/* exp(x), x <= 0 expected (softmax after max subtraction); valid for |x| < 700 */
static inline double bxo_exp(double x)
{
    if (x < -700.0) return 0.0;
    if (x > 700.0) x = 700.0;
    double kf = floor(x * 1.4426950408889634074 + 0.5);
    double r = (x - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    /* Taylor, |r| <= 0.3466: degree 14 -> error < 1e-18 */
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
    return p * bxo_pow2i((int)kf);
}

/* sin and cos of x, |x| < 1e5 */
static inline void bxo_sincos(double x, double *s, double *c)
{
    double kf = floor(x * 0.63661977236758134308 + 0.5);
    double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
    r = r - kf * 2.02226624879595063154e-21;
    double r2 = r * r;
    /* sin Taylor deg 19, cos Taylor deg 18 on |r| <= pi/4 */
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
    long k = (long)kf;
    int q = (int)(k & 3);
    if (q == 0) { *s = sn; *c = cs; }
    else if (q == 1) { *s = cs; *c = -sn; }
    else if (q == 2) { *s = -sn; *c = -cs; }
    else { *s = -cs; *c = sn; }
}

/* asin(t) for |t| <= 0.5 : sum_{n} C(2n,n)/(4^n (2n+1)) t^(2n+1), 30 terms */
static inline double bxo_asin_small(double t)
{
    double t2 = t * t;
    double term = t; /* coefficient*t^(2n+1) without the 1/(2n+1) */
    double sum = t;
    for (int n = 1; n <= 30; ++n) {
        term = term * t2 * ((double)(2 * n - 1) / (double)(2 * n));
        sum = sum + term / (double)(2 * n + 1);
    }
    return sum;
}

static inline double bxo_acos(double x)
{
    const double PI = 3.14159265358979323846;
    if (x > 1.0) x = 1.0;
    if (x < -1.0) x = -1.0;
    double ax = x < 0 ? -x : x;
    if (ax <= 0.5) return PI * 0.5 - bxo_asin_small(x);
    double z = (1.0 - ax) * 0.5;
    double a = 2.0 * bxo_asin_small(sqrt(z));
    return x > 0 ? a : PI - a;
}

/* natural log, x > 0 finite normal */
static inline double bxo_log(double x)
{
    uint64_t u = bxo_f64_as_u64(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = bxo_u64_as_f64((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
    double f = (m - 1.0) / (m + 1.0);
    double f2 = f * f;
    double sum = 0.0;
    for (int n = 14; n >= 0; --n) sum = sum * f2 + 1.0 / (double)(2 * n + 1);
    double lm = 2.0 * f * sum;
    return ((double)e * 6.93147180369123816490e-01 + lm) + (double)e * 1.90821492927058770002e-10;
}

/* ---- symmetric 3x3 eigen-decomposition, cyclic Jacobi, fixed 10 sweeps (binary64) ----
 * a: in  symmetric matrix (row-major 9), destroyed; v: out eigenvectors as COLUMNS (row-major 9);
 * w: out eigenvalues a[0],a[4],a[8] (unsorted).                                               */
static inline void bxo_jacobi3(double a[9], double v[9], double w[3])
{
    for (int i = 0; i < 9; ++i) v[i] = 0.0;
    v[0] = v[4] = v[8] = 1.0;
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 10; ++sweep) {
        for (int r = 0; r < 3; ++r) {
            int p = PQ[r][0], q = PQ[r][1];
            double apq = a[p * 3 + q];
            if (apq == 0.0) continue;
            double app = a[p * 3 + p], aqq = a[q * 3 + q];
            double theta = (aqq - app) / (2.0 * apq);
            double at = theta < 0 ? -theta : theta;
            double t = 1.0 / (at + sqrt(theta * theta + 1.0));
            if (theta < 0) t = -t;
            double c = 1.0 / sqrt(t * t + 1.0);
            double s = t * c;
            /* A <- J^T A J */
            for (int k = 0; k < 3; ++k) {
                double akp = a[k * 3 + p], akq = a[k * 3 + q];
                a[k * 3 + p] = c * akp - s * akq;
                a[k * 3 + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {
                double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                a[p * 3 + k] = c * apk - s * aqk;
                a[q * 3 + k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {
                double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
                v[k * 3 + p] = c * vkp - s * vkq;
                v[k * 3 + q] = s * vkp + c * vkq;
            }
        }
    }
    w[0] = a[0]; w[1] = a[4]; w[2] = a[8];
}

/* Proper rotation R (row-major 9) maximising trace(R * H^T)... i.e. Kabsch/Umeyama:
 * given H = sum_p  a_p b_p^T (3x3, row-major; a = source, b = target),  returns R with  b ~ R a.
 * SVD H = U S V^T  =>  R = V diag(1,1,det(V U^T)) U^T.
 * Built from the eigen-decomposition of H^T H (Jacobi above); rank-2 H (3-point samples) handled by
 * completing the third singular vectors with cross products.  Returns 0 on rank < 2.            */
static inline int bxo_kabsch_from_H(const double H[9], double R[9])
{
    double hth[9], V[9], w[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            hth[i * 3 + j] = (H[0 * 3 + i] * H[0 * 3 + j] + H[1 * 3 + i] * H[1 * 3 + j]) + H[2 * 3 + i] * H[2 * 3 + j];
    /* exact symmetrisation is implicit: (i,j) and (j,i) use the same products in the same order */
    bxo_jacobi3(hth, V, w);
    /* order eigenvalues descending (stable) */
    int o0 = 0, o1 = 1, o2 = 2, tmp;
    if (w[o1] > w[o0]) { tmp = o0; o0 = o1; o1 = tmp; }
    if (w[o2] > w[o0]) { tmp = o0; o0 = o2; o2 = tmp; }
    if (w[o2] > w[o1]) { tmp = o1; o1 = o2; o2 = tmp; }
    double v1[3] = {V[0 * 3 + o0], V[1 * 3 + o0], V[2 * 3 + o0]};
    double v2[3] = {V[0 * 3 + o1], V[1 * 3 + o1], V[2 * 3 + o1]};
    double l1 = w[o0], l2 = w[o1];
    if (!(l1 > 0.0) || !(l2 > l1 * 1e-24)) return 0;
    /* u_i = H v_i / |H v_i| */
    double u1[3], u2[3];
    for (int i = 0; i < 3; ++i) {
        u1[i] = (H[i * 3 + 0] * v1[0] + H[i * 3 + 1] * v1[1]) + H[i * 3 + 2] * v1[2];
        u2[i] = (H[i * 3 + 0] * v2[0] + H[i * 3 + 1] * v2[1]) + H[i * 3 + 2] * v2[2];
    }
    double n1 = sqrt((u1[0] * u1[0] + u1[1] * u1[1]) + u1[2] * u1[2]);
    if (!(n1 > 0.0)) return 0;
    for (int i = 0; i < 3; ++i) u1[i] = u1[i] / n1;
    /* Gram-Schmidt u2 against u1 (numerically they are already orthogonal) */
    double d12 = (u1[0] * u2[0] + u1[1] * u2[1]) + u1[2] * u2[2];
    for (int i = 0; i < 3; ++i) u2[i] = u2[i] - d12 * u1[i];
    double n2 = sqrt((u2[0] * u2[0] + u2[1] * u2[1]) + u2[2] * u2[2]);
    if (!(n2 > 0.0)) return 0;
    for (int i = 0; i < 3; ++i) u2[i] = u2[i] / n2;
    /* third vectors by cross products: det(U)=det(V)=+1, so the det-fix is the identity and
     * R = V U^T = sum_i v_i u_i^T  is the proper rotation closest to H (reflection case handled
     * because the smallest singular direction is the one whose sign is flipped).               */
    double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    /* H maps source (rows index a) to target?  H = sum a b^T ; H v = sum a (b.v): u lives in the
     * SOURCE space and v in the TARGET space, so  R = V U^T  (target <- source).               */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = (v1[i] * u1[j] + v2[i] * u2[j]) + v3[i] * u3[j];
    return 1;
}

#endif
