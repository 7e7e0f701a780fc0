// Kernels of the Initializer (libs/initializer.cpp) and of the square-marker pose solver it calls, aruco::solvePnP_
// (3rdparty/aruco/aruco/ippe.cpp:118-223).  gfx950 only; no CPU path.
//
//   k_ippe          one thread per detection: undistort -> homography of the square -> the two IPPE rotations -> translations ->
//                   float reprojection errors -> float-rounded 4x4 poses.  Streaming: 32 B in, ~270 B out per detection.
//   k_pair_cands    one thread per candidate of a camera-pair / marker-pair set: (T, T1^-1, T2^-1) of
//                   fill_transformation_sets (:95-125) from two stored poses.  Streaming.
//   k_object_cands  the same for fill_transformation_set (:73-93), the per-frame sets of init_object_transforms.
//   k_vote          find_best_transformation (:151-193): cost_i = sum_j sum_corners |p - T2inv_j T_i T1inv_j p|.  The n^2 part:
//                   one wavefront per 64 candidates i of a set, T_i in registers, the j-side (24 doubles) read through the
//                   scalar cache because it is uniform across the wavefront; ~160 fp64 operations per (i,j): VALU-bound.
//
// All matrices are the 3x4 top of the reference's 4x4 CV_64F matrices (bottom row 0 0 0 1 stays exact under products and
// inverses), row-major.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <cmath>
#include <limits>
#include <vector>

#include "../host/init_device.h"
#include "hostcopy.h"

namespace aar {

struct CamTab {
    double K[9];
    double k[AAR_MAX_DIST];
};

// ---------------------------------------------------------------------------------------------------------------------
// IPPE.  Contraction is off: the float error below is a chain of individually rounded operations (ippe.cpp:289-321), and the
// double part then rounds like the reference's scalar code as well.
// ---------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

__device__ __forceinline__ void d_mat3mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// least-squares translation for a fixed rotation (ippe.cpp:347-425): [n 0 Sa; 0 n Sb; Sa Sb Sq] t = B
__device__ __forceinline__ void d_ippe_translation(float hf, const float *q, const double *R, double *t) {
    const float mx[4] = {-hf, hf, hf, -hf}, my[4] = {hf, hf, -hf, -hf};
    double Sa = 0, Sb = 0, Sq = 0, B0 = 0, B1 = 0, B2 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double X = mx[i], Y = my[i], Z = 0.0;
        const double rx = R[0] * X + R[1] * Y + R[2] * Z, ry = R[3] * X + R[4] * Y + R[5] * Z, rz = R[6] * X + R[7] * Y + R[8] * Z;
        const double a = -(double)q[2 * i], b = -(double)q[2 * i + 1];
        Sa += a; Sb += b; Sq += a * a + b * b;
        const double bx = (double)q[2 * i] * rz - rx, by = (double)q[2 * i + 1] * rz - ry;
        B0 += bx; B1 += by; B2 += a * bx + b * by;
    }
    const double n = 4;
    const double dinv = 1.0 / (n * n * Sq - n * Sb * Sb - Sa * n * Sa);
    t[0] = dinv * ((n * Sq - Sb * Sb) * B0 + (Sa * Sb) * B1 + (-Sa * n) * B2);
    t[1] = dinv * ((Sb * Sa) * B0 + (n * Sq - Sa * Sa) * B1 + (-n * Sb) * B2);
    t[2] = dinv * ((-n * Sa) * B0 + (-n * Sb) * B1 + (n * n) * B2);
}

__device__ __forceinline__ float d_ippe_error(float hf, const float *q, const double *R, const double *t) {
    const float mx[4] = {-hf, hf, hf, -hf}, my[4] = {hf, hf, -hf, -hf};
    float err = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float px = (float)(R[0] * mx[i]) + (float)(R[1] * my[i]) + (float)(R[2] * 0.0f + t[0]);
        const float py = (float)(R[3] * mx[i]) + (float)(R[4] * my[i]) + (float)(R[5] * 0.0f + t[1]);
        const float pz = (float)(R[6] * mx[i]) + (float)(R[7] * my[i]) + (float)(R[8] * 0.0f + t[2]);
        const float dx = px / pz - q[2 * i], dy = py / pz - q[2 * i + 1];
        err = err + sqrtf(dx * dx + dy * dy);
    }
    return err;
}

// IPPERot2vec (ippe.cpp:323-345), cv::Rodrigues back to a matrix and the CV_32F conversion of getRTMatrix (:40-93)
__device__ __forceinline__ void d_store_pose(const double *R, const double *t, double *out) {
    const double w = acos((R[0] + R[4] + R[8] - 1.0) / 2.0);
    double rv0 = 0, rv1 = 0, rv2 = 0;
    if (!(w < DBL_EPSILON)) {
        const double d = 1 / (2 * sin(w)) * w;
        rv0 = d * (R[7] - R[5]); rv1 = d * (R[2] - R[6]); rv2 = d * (R[3] - R[1]);
    }
    double M[9];
    const double th = sqrt(rv0 * rv0 + rv1 * rv1 + rv2 * rv2);
    if (th < DBL_EPSILON) {
#pragma unroll
        for (int i = 0; i < 9; i++) M[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        const double c = cos(th), s = sin(th), c1 = 1. - c, ith = 1. / th;
        const double x = rv0 * ith, y = rv1 * ith, z = rv2 * ith;
        M[0] = c + c1 * x * x;     M[1] = c1 * x * y - s * z; M[2] = c1 * x * z + s * y;
        M[3] = c1 * x * y + s * z; M[4] = c + c1 * y * y;     M[5] = c1 * y * z - s * x;
        M[6] = c1 * x * z - s * y; M[7] = c1 * y * z + s * x; M[8] = c + c1 * z * z;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) out[i * 4 + j] = (double)(float)M[i * 3 + j];
        out[i * 4 + 3] = (double)(float)t[i];
    }
}

__global__ void __launch_bounds__(64) k_ippe(long long n, const float *__restrict__ uv, const int *__restrict__ det_cam,
                                             const CamTab *__restrict__ cams, float hf, double *__restrict__ poses,
                                             float *__restrict__ e1, float *__restrict__ e2, float *__restrict__ uvK) {
    const long long d = (long long)blockIdx.x * 64 + threadIdx.x;
    if (d >= n) return;
    const CamTab &cm = cams[det_cam[d]];
    const double *K = cm.K, *k = cm.k;
    const double ifx = 1.0 / K[0], ify = 1.0 / K[4];
    float q[8];
    // cv::undistortPoints: five fixed-point iterations of the inverse distortion model; normalised output for IPPE
    // (ippe.cpp:167), P = K output for the data set (libs/multicam_mapper.cpp:554-578)
#pragma unroll
    for (int c = 0; c < 4; c++) {
        double x = ((double)uv[8 * d + 2 * c] - K[2]) * ifx, y = ((double)uv[8 * d + 2 * c + 1] - K[5]) * ify;
        const double x0 = x, y0 = y;
#pragma unroll 1
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = (1.0 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double dx = 2.0 * k[2] * x * y + k[3] * (r2 + 2.0 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dy = k[2] * (r2 + 2.0 * y * y) + 2.0 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        q[2 * c] = (float)x;
        q[2 * c + 1] = (float)y;
        const double xx = K[0] * x + K[1] * y + K[2], yy = K[3] * x + K[4] * y + K[5], ww = 1.0 / (K[6] * x + K[7] * y + K[8]);
        uvK[8 * d + 2 * c] = (float)(xx * ww);
        uvK[8 * d + 2 * c + 1] = (float)(yy * ww);
    }
    // homography of the square (-h,h),(h,h),(h,-h),(-h,-h) onto q, h22 = 1: unit square -> quadrilateral, composed with the
    // affine map of the marker frame (the closed form ippe.cpp:538-578 expands)
    double H[9];
    {
        const double h = (double)hf;
        const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
        const double sx = x0 - x1 + x2 - x3, sy = y0 - y1 + y2 - y3;
        const double dx1 = x1 - x2, dx2 = x3 - x2, dy1 = y1 - y2, dy2 = y3 - y2;
        const double den = dx1 * dy2 - dy1 * dx2;
        const double g = (sx * dy2 - sy * dx2) / den, kk = (dx1 * sy - dy1 * sx) / den;
        const double U[9] = {x1 - x0 + g * x1, x3 - x0 + kk * x3, x0, y1 - y0 + g * y1, y3 - y0 + kk * y3, y0, g, kk, 1.0};
        const double s = 1.0 / (2.0 * h);
        double Hn[9];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Hn[r * 3 + 0] = U[r * 3 + 0] * s;
            Hn[r * 3 + 1] = -U[r * 3 + 1] * s;
            Hn[r * 3 + 2] = 0.5 * (U[r * 3 + 0] + U[r * 3 + 1]) + U[r * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) H[i] = Hn[i] / Hn[8];
    }
    // the two rotations (ippe.cpp:427-536; IPPE paper, Algorithm 1)
    double Ra[9], Rb[9];
    {
        const double J0 = H[0] - H[6] * H[2], J1 = H[1] - H[7] * H[2], J2 = H[3] - H[6] * H[5], J3 = H[4] - H[7] * H[5];
        const double p = H[2], qq = H[5];
        const double s = sqrt(p * p + qq * qq + 1), t = sqrt(p * p + qq * qq);
        const double ct = 1 / s, st = sqrt(1 - 1 / (s * s));
        const double kx = p / t, ky = qq / t;
        const double Rv[9] = {(ct - 1) * kx * kx + 1, kx * ky * (ct - 1),     kx * st,
                              kx * ky * (ct - 1),     (ct - 1) * ky * ky + 1, ky * st,
                              -kx * st,               -ky * st,               (ct - 1) * (kx * kx + ky * ky) + 1};
        const double b00 = Rv[0] - p * Rv[6], b01 = Rv[1] - p * Rv[7], b10 = Rv[3] - qq * Rv[6], b11 = Rv[4] - qq * Rv[7];
        const double di = 1.0 / (b00 * b11 - b01 * b10);
        const double i00 = di * b11, i01 = -di * b01, i10 = -di * b10, i11 = di * b00;
        const double a00 = i00 * J0 + i01 * J2, a01 = i00 * J1 + i01 * J3;
        const double a10 = i10 * J0 + i11 * J2, a11 = i10 * J1 + i11 * J3;
        const double n00 = a00 * a00 + a01 * a01, n01 = a00 * a10 + a01 * a11, n11 = a10 * a10 + a11 * a11;
        const double gamma = sqrt(0.5 * (n00 + n11 + sqrt((n00 - n11) * (n00 - n11) + 4.0 * n01 * n01)));
        const double r00 = a00 / gamma, r01 = a01 / gamma, r10 = a10 / gamma, r11 = a11 / gamma;
        const double b0 = sqrt(-r00 * r00 - r10 * r10 + 1);
        double b1 = sqrt(-r01 * r01 - r11 * r11 + 1);
        if (-r00 * r01 - r10 * r11 < 0) b1 = -b1;
        const double Qa[9] = {r00, r01, b1 * r10 - b0 * r11, r10, r11, b0 * r01 - b1 * r00, b0, b1, r00 * r11 - r01 * r10};
        const double Qb[9] = {r00, r01, b0 * r11 - b1 * r10, r10, r11, b1 * r00 - b0 * r01, -b0, -b1, r00 * r11 - r01 * r10};
        d_mat3mul(Rv, Qa, Ra);
        d_mat3mul(Rv, Qb, Rb);
    }
    double ta[3], tb[3];
    d_ippe_translation(hf, q, Ra, ta);
    d_ippe_translation(hf, q, Rb, tb);
    const float ea = d_ippe_error(hf, q, Ra, ta), eb = d_ippe_error(hf, q, Rb, tb);
    const bool a_first = ea < eb;
    d_store_pose(Ra, ta, poses + (2 * d + (a_first ? 0 : 1)) * 12);
    d_store_pose(Rb, tb, poses + (2 * d + (a_first ? 1 : 0)) * 12);
    e1[d] = a_first ? ea : eb;
    e2[d] = a_first ? eb : ea;
}

#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------------------------------
// 3x4 affine helpers
// ---------------------------------------------------------------------------------------------------------------------
struct Aff {
    double m[12];
};

__device__ __forceinline__ Aff aff_load(const double *__restrict__ p) {
    Aff a;
#pragma unroll
    for (int i = 0; i < 12; i++) a.m[i] = p[i];
    return a;
}
__device__ __forceinline__ void aff_store(double *__restrict__ p, const Aff &a) {
#pragma unroll
    for (int i = 0; i < 12; i++) p[i] = a.m[i];
}
__device__ __forceinline__ Aff aff_identity() {
    Aff a;
#pragma unroll
    for (int i = 0; i < 12; i++) a.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return a;
}
// x * y, every element accumulated over k = 0..3 in order as a 4x4 cv::Mat product does
__device__ __forceinline__ Aff aff_mul(const Aff &x, const Aff &y) {
    Aff r;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++)
            r.m[i * 4 + j] = x.m[i * 4] * y.m[j] + x.m[i * 4 + 1] * y.m[4 + j] + x.m[i * 4 + 2] * y.m[8 + j];
        r.m[i * 4 + 3] = x.m[i * 4] * y.m[3] + x.m[i * 4 + 1] * y.m[7] + x.m[i * 4 + 2] * y.m[11] + x.m[i * 4 + 3];
    }
    return r;
}
// general inverse (cv::Mat::inv() of the 4x4): the poses are float-rounded, so R^T is NOT the inverse
__device__ __forceinline__ Aff aff_inv(const Aff &x) {
    const double a = x.m[0], b = x.m[1], c = x.m[2], d = x.m[4], e = x.m[5], f = x.m[6], g = x.m[8], h = x.m[9], i = x.m[10];
    const double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const double idet = 1.0 / (a * c00 + b * c01 + c * c02);
    Aff r;
    r.m[0] = c00 * idet; r.m[1] = (c * h - b * i) * idet; r.m[2] = (b * f - c * e) * idet;
    r.m[4] = c01 * idet; r.m[5] = (a * i - c * g) * idet; r.m[6] = (c * d - a * f) * idet;
    r.m[8] = c02 * idet; r.m[9] = (b * g - a * h) * idet; r.m[10] = (a * e - b * d) * idet;
#pragma unroll
    for (int k = 0; k < 3; k++) r.m[k * 4 + 3] = -(r.m[k * 4] * x.m[3] + r.m[k * 4 + 1] * x.m[7] + r.m[k * 4 + 2] * x.m[11]);
    return r;
}

// j-side record of the vote: T2_inv (12) followed by T1_inv * corners (3 x 4, column c = corner c)
__device__ __forceinline__ void store_jside(double *__restrict__ bj, const Aff &T1inv, const Aff &T2inv, double h) {
    aff_store(bj, T2inv);
    const double px[4] = {-h, h, h, -h}, py[4] = {h, h, -h, -h};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) bj[12 + r * 4 + c] = T1inv.m[r * 4] * px[c] + T1inv.m[r * 4 + 1] * py[c] + T1inv.m[r * 4 + 3];
}

__global__ void __launch_bounds__(256) k_pair_cands(long long n, int type, const int *__restrict__ ca, const int *__restrict__ cb,
                                                    const double *__restrict__ poses, double h, double *__restrict__ Tc,
                                                    double *__restrict__ BJ) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const Aff P1 = aff_load(poses + 12LL * ca[k]), P2 = aff_load(poses + 12LL * cb[k]);
    if (type == 0) {  // cameras: T = P2 * P1^-1, T1_inv = P1, T2_inv = P2^-1
        const Aff P2i = aff_inv(P2);
        aff_store(Tc + 12 * k, aff_mul(P2, aff_inv(P1)));
        store_jside(BJ + 24 * k, P1, P2i, h);
    } else {          // markers: T = P2^-1 * P1, T1_inv = P1^-1, T2_inv = P2
        const Aff P2i = aff_inv(P2);
        aff_store(Tc + 12 * k, aff_mul(P2i, P1));
        store_jside(BJ + 24 * k, aff_inv(P1), P2, h);
    }
}

__global__ void __launch_bounds__(256) k_object_cands(long long n, const int *__restrict__ cpose, const int *__restrict__ ccam,
                                                      const int *__restrict__ cmk, const double *__restrict__ poses,
                                                      const double *__restrict__ Tcr, const double *__restrict__ Tmr, double h,
                                                      double *__restrict__ Tc, double *__restrict__ BJ) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const Aff T_mc = aff_load(poses + 12LL * cpose[k]);
    const Aff T_cr = aff_load(Tcr + 12LL * ccam[k]), T_mr = aff_load(Tmr + 12LL * cmk[k]);
    const Aff T_rm = aff_inv(T_mr), T_rc = aff_inv(T_cr), T_cm = aff_inv(T_mc);
    aff_store(Tc + 12 * k, aff_mul(aff_mul(T_cr, T_mc), T_rm));   // root marker -> root camera through this detection
    store_jside(BJ + 24 * k, aff_mul(T_mr, T_cm), T_rc, h);
}

// generic entry (aar_vote_transforms): j-side records from explicit T1_inv / T2_inv
__global__ void __launch_bounds__(256) k_prep_jside(long long n, const double *__restrict__ A, const double *__restrict__ B,
                                                    double h, double *__restrict__ BJ) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    store_jside(BJ + 24 * k, aff_load(A + 12 * k), aff_load(B + 12 * k), h);
}

// sqrt for the vote: v_rsq_f64 seed (~2^-26) + one coupled Goldschmidt step + one residual correction = full double accuracy
// (<= 1 ulp) in 8 instructions, without the range scaling of the library sqrt (the arguments are squared distances of
// metre-sized scenes, nowhere near the denormals); an exact zero stays an exact zero.
__device__ __forceinline__ double vote_sqrt(double q) {
    const double y = __builtin_amdgcn_rsq(q);
    double g = q * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, q);
    g = fma(d, h, g);
    return q == 0.0 ? 0.0 : g;   // (a NaN argument stays NaN: such a candidate must never win the vote)
}

// work item: candidates [i0, i0+64) of the set [begin, end)
__global__ void __launch_bounds__(64) k_vote(const int4 *__restrict__ items, const double *__restrict__ Tc,
                                             const double *__restrict__ BJ, double h, double *__restrict__ cost) {
    const int4 it = items[blockIdx.x];
    const int begin = it.x, end = it.y;
    const int i = it.z + (int)threadIdx.x;
    const int il = i < end ? i : end - 1;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = Tc[12LL * il + k];
    const double px[4] = {-h, h, h, -h}, py[4] = {h, h, -h, -h};
    double acc = 0;
#pragma unroll 1
    for (int j = begin; j < end; j++) {
        const double *__restrict__ b = BJ + 24LL * j;   // uniform across the wavefront: scalar loads
        double s = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double ax = b[12 + c], ay = b[16 + c], az = b[20 + c];
            // three-term rows as FMA chains ending in the translation (one instruction per product, none for the additions)
            const double rx = fma(T[0], ax, fma(T[1], ay, fma(T[2], az, T[3])));
            const double ry = fma(T[4], ax, fma(T[5], ay, fma(T[6], az, T[7])));
            const double rz = fma(T[8], ax, fma(T[9], ay, fma(T[10], az, T[11])));
            const double dx = px[c] - fma(b[0], rx, fma(b[1], ry, fma(b[2], rz, b[3])));
            const double dy = py[c] - fma(b[4], rx, fma(b[5], ry, fma(b[6], rz, b[7])));
            const double dz = -fma(b[8], rx, fma(b[9], ry, fma(b[10], rz, b[11])));
            s += vote_sqrt(fma(dx, dx, fma(dy, dy, dz * dz)));
        }
        acc += s;
    }
    if (i < end) cost[i] = acc;
}

__global__ void __launch_bounds__(256) k_gather12(int n, const int *__restrict__ idx, const double *__restrict__ src,
                                                  double *__restrict__ dst) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n * 12) return;
    const int s = k / 12, e = k - s * 12;
    dst[k] = idx[s] >= 0 ? src[12LL * idx[s] + e] : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T *p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)); }
    hipError_t upload(const T *h, size_t n) {
        hipError_t e = alloc(n);
        if (e == hipSuccess && n) e = (hipError_t)h2d(p, h, sizeof(T) * n, nullptr);   // (through page-locked staging: hostcopy.h)
        return e;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

struct InitDevice {
    int32_t device_id = 0;
    double *poses = nullptr;   // [2 * n_det][12]
    int64_t n_poses = 0;
    ~InitDevice() { if (poses) (void)hipFree(poses); }
};

static int hip_fail(const char *what, hipError_t e) { return set_error(AAR_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); }

#define HIPCHK(expr, what)                                  \
    do {                                                    \
        hipError_t e__ = (expr);                            \
        if (e__ != hipSuccess) return hip_fail(what, e__);  \
    } while (0)

static int select_device(int32_t device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(AAR_ERR_NO_DEVICE, "no HIP device available; this library has no CPU path");
    if (device_id < 0 || device_id >= ndev) return set_error(AAR_ERR_INVALID, "device_id %d out of range (%d devices)", device_id, ndev);
    if (hipSetDevice(device_id) != hipSuccess) return set_error(AAR_ERR_HIP, "hipSetDevice failed");
    return AAR_OK;
}

int initdev_create(int32_t device_id, InitDevice **out) {
    if (int rc = select_device(device_id)) return rc;
    *out = new InitDevice;
    (*out)->device_id = device_id;
    return AAR_OK;
}

void initdev_destroy(InitDevice *d) { delete d; }

int initdev_ippe(InitDevice *D, const aar_cam_model *cams, int32_t n_cams, float marker_size, int64_t n, const float *uv,
                 const int32_t *det_cam, float *e1, float *e2, float *uv_undistorted) {
    if (D->poses) { (void)hipFree(D->poses); D->poses = nullptr; }
    D->n_poses = 2 * n;
    HIPCHK(hipMalloc((void **)&D->poses, sizeof(double) * 12 * (size_t)(D->n_poses ? D->n_poses : 1)), "hipMalloc(poses)");
    if (n == 0) return AAR_OK;
    std::vector<CamTab> tab((size_t)n_cams);
    for (int c = 0; c < n_cams; c++) {
        for (int i = 0; i < 9; i++) tab[c].K[i] = cams[c].K[i];
        for (int i = 0; i < AAR_MAX_DIST; i++) tab[c].k[i] = i < cams[c].n_dist ? cams[c].dist[i] : 0.0;
    }
    DevBuf<CamTab> d_tab;
    DevBuf<float> d_uv, d_e1, d_e2, d_uvK;
    DevBuf<int> d_cam;
    HIPCHK(d_tab.upload(tab.data(), tab.size()), "upload(cameras)");
    HIPCHK(d_uv.upload(uv, 8 * (size_t)n), "upload(corners)");
    HIPCHK(d_cam.upload(det_cam, (size_t)n), "upload(det_cam)");
    HIPCHK(d_e1.alloc((size_t)n), "hipMalloc");
    HIPCHK(d_e2.alloc((size_t)n), "hipMalloc");
    HIPCHK(d_uvK.alloc(8 * (size_t)n), "hipMalloc");
    hipLaunchKernelGGL(k_ippe, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, (long long)n, d_uv.p, d_cam.p, d_tab.p,
                       marker_size / 2.0f, D->poses, d_e1.p, d_e2.p, d_uvK.p);
    HIPCHK(hipGetLastError(), "k_ippe");
    HIPCHK((hipError_t)d2h(e1, d_e1.p, sizeof(float) * n, nullptr), "download(e1)");
    HIPCHK((hipError_t)d2h(e2, d_e2.p, sizeof(float) * n, nullptr), "download(e2)");
    HIPCHK((hipError_t)d2h(uv_undistorted, d_uvK.p, sizeof(float) * 8 * n, nullptr), "download(corners)");
    return AAR_OK;
}

// vote every set of device-resident candidates; first minimum per set (NaN never wins, as `curr_error < min_error`)
static int vote_sets(int64_t n_cand, const double *Tc, const double *BJ, int64_t n_sets, const int64_t *set_begin,
                     double marker_size, int64_t *best, double *weight, double *best_T, double *cost_out) {
    std::vector<int4> items;
    for (int64_t s = 0; s < n_sets; s++)
        for (int64_t i0 = set_begin[s]; i0 < set_begin[s + 1]; i0 += 64)
            items.push_back(make_int4((int)set_begin[s], (int)set_begin[s + 1], (int)i0, 0));
    std::vector<double> cost((size_t)n_cand);
    if (!items.empty()) {
        DevBuf<int4> d_items;
        DevBuf<double> d_cost;
        HIPCHK(d_items.upload(items.data(), items.size()), "upload(vote items)");
        HIPCHK(d_cost.alloc((size_t)n_cand), "hipMalloc(cost)");
        hipLaunchKernelGGL(k_vote, dim3((unsigned)items.size()), dim3(64), 0, 0, d_items.p, Tc, BJ, marker_size / 2, d_cost.p);
        HIPCHK(hipGetLastError(), "k_vote");
        HIPCHK((hipError_t)d2h(cost.data(), d_cost.p, sizeof(double) * n_cand, nullptr), "download(cost)");
    }
    if (cost_out && n_cand) memcpy(cost_out, cost.data(), sizeof(double) * n_cand);
    std::vector<int> bidx((size_t)n_sets);
    for (int64_t s = 0; s < n_sets; s++) {
        double mn = std::numeric_limits<double>::max();
        int64_t at = -1;
        for (int64_t i = set_begin[s]; i < set_begin[s + 1]; i++)
            if (cost[i] < mn) { mn = cost[i]; at = i; }
        best[s] = at < 0 ? -1 : at - set_begin[s];
        if (weight) weight[s] = at < 0 ? 0.0 : mn;
        bidx[s] = (int)at;
    }
    if (best_T && n_sets) {
        DevBuf<int> d_idx;
        DevBuf<double> d_out;
        HIPCHK(d_idx.upload(bidx.data(), bidx.size()), "upload(best)");
        HIPCHK(d_out.alloc(12 * (size_t)n_sets), "hipMalloc(best_T)");
        hipLaunchKernelGGL(k_gather12, dim3((unsigned)((12 * n_sets + 255) / 256)), dim3(256), 0, 0, (int)n_sets, d_idx.p, Tc, d_out.p);
        HIPCHK(hipGetLastError(), "k_gather12");
        HIPCHK((hipError_t)d2h(best_T, d_out.p, sizeof(double) * 12 * n_sets, nullptr), "download(best_T)");
    }
    return AAR_OK;
}

static int check_sets(int64_t n_cand, int64_t n_sets, const int64_t *set_begin) {
    if (n_cand < 0 || n_sets < 0 || n_cand >= (1LL << 31) - 64) return set_error(AAR_ERR_INVALID, "vote: bad candidate count");
    if (n_sets > 0 && (!set_begin || set_begin[0] != 0 || set_begin[n_sets] != n_cand))
        return set_error(AAR_ERR_INVALID, "vote: set ranges must tile [0, n)");
    for (int64_t s = 0; s < n_sets; s++)
        if (set_begin[s + 1] < set_begin[s]) return set_error(AAR_ERR_INVALID, "vote: set ranges must ascend");
    return AAR_OK;
}

int initdev_pair_vote(InitDevice *D, int type, int64_t n_cand, const int32_t *a, const int32_t *b, int64_t n_sets,
                      const int64_t *set_begin, double marker_size, int64_t *best, double *weight, double *best_T) {
    if (int rc = check_sets(n_cand, n_sets, set_begin)) return rc;
    if (n_sets == 0) return AAR_OK;
    DevBuf<int> d_a, d_b;
    DevBuf<double> d_T, d_BJ;
    HIPCHK(d_a.upload(a, (size_t)n_cand), "upload(candidates)");
    HIPCHK(d_b.upload(b, (size_t)n_cand), "upload(candidates)");
    HIPCHK(d_T.alloc(12 * (size_t)n_cand), "hipMalloc(T)");
    HIPCHK(d_BJ.alloc(24 * (size_t)n_cand), "hipMalloc(j-side)");
    if (n_cand) {
        hipLaunchKernelGGL(k_pair_cands, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, 0, (long long)n_cand, type, d_a.p,
                           d_b.p, D->poses, marker_size / 2, d_T.p, d_BJ.p);
        HIPCHK(hipGetLastError(), "k_pair_cands");
    }
    return vote_sets(n_cand, d_T.p, d_BJ.p, n_sets, set_begin, marker_size, best, weight, best_T, nullptr);
}

int initdev_object_vote(InitDevice *D, int64_t n_cand, const int32_t *cand_pose, const int32_t *cand_cam,
                        const int32_t *cand_marker, int32_t n_cams, const double *to_root_cam, int32_t n_markers,
                        const double *to_root_marker, int64_t n_sets, const int64_t *set_begin, double marker_size,
                        int64_t *best, double *weight, double *best_T) {
    if (int rc = check_sets(n_cand, n_sets, set_begin)) return rc;
    if (n_sets == 0) return AAR_OK;
    DevBuf<int> d_p, d_c, d_m;
    DevBuf<double> d_Tcr, d_Tmr, d_T, d_BJ;
    HIPCHK(d_p.upload(cand_pose, (size_t)n_cand), "upload(candidates)");
    HIPCHK(d_c.upload(cand_cam, (size_t)n_cand), "upload(candidates)");
    HIPCHK(d_m.upload(cand_marker, (size_t)n_cand), "upload(candidates)");
    HIPCHK(d_Tcr.upload(to_root_cam, 12 * (size_t)n_cams), "upload(to_root_cam)");
    HIPCHK(d_Tmr.upload(to_root_marker, 12 * (size_t)n_markers), "upload(to_root_marker)");
    HIPCHK(d_T.alloc(12 * (size_t)n_cand), "hipMalloc(T)");
    HIPCHK(d_BJ.alloc(24 * (size_t)n_cand), "hipMalloc(j-side)");
    if (n_cand) {
        hipLaunchKernelGGL(k_object_cands, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, 0, (long long)n_cand, d_p.p, d_c.p,
                           d_m.p, D->poses, d_Tcr.p, d_Tmr.p, marker_size / 2, d_T.p, d_BJ.p);
        HIPCHK(hipGetLastError(), "k_object_cands");
    }
    return vote_sets(n_cand, d_T.p, d_BJ.p, n_sets, set_begin, marker_size, best, weight, best_T, nullptr);
}

}  // namespace aar

using namespace aar;

static void top3x4(const double *M16, double *m12, int64_t n) {
    for (int64_t i = 0; i < n; i++) memcpy(m12 + 12 * i, M16 + 16 * i, sizeof(double) * 12);
}

extern "C" int aar_ippe_square(double marker_size, const aar_cam_model *cam, int64_t n, const float *uv, double *T1, double *err1,
                               double *T2, double *err2, int32_t device_id) {
    if (!cam || n < 0 || (n > 0 && (!uv || !T1 || !T2 || !err1 || !err2)) || cam->n_dist < 0 || cam->n_dist > AAR_MAX_DIST)
        return set_error(AAR_ERR_INVALID, "aar_ippe_square: bad argument");
    InitDevice *D = nullptr;
    if (int rc = initdev_create(device_id, &D)) return rc;
    std::vector<int32_t> dc((size_t)n, 0);
    std::vector<float> e1((size_t)n), e2((size_t)n), uvK(8 * (size_t)n);
    int rc = initdev_ippe(D, cam, 1, (float)marker_size, n, uv, dc.data(), e1.data(), e2.data(), uvK.data());
    std::vector<double> P(24 * (size_t)n);
    if (!rc && n) {
        hipError_t e = (hipError_t)d2h(P.data(), D->poses, sizeof(double) * 24 * n, nullptr);
        if (e != hipSuccess) rc = hip_fail("download(poses)", e);
    }
    initdev_destroy(D);
    if (rc) return rc;
    for (int64_t d = 0; d < n; d++) {
        for (int s = 0; s < 2; s++) {
            double *T = (s ? T2 : T1) + 16 * d;
            memcpy(T, &P[(2 * d + s) * 12], sizeof(double) * 12);
            T[12] = T[13] = T[14] = 0; T[15] = 1;
        }
        err1[d] = e1[d]; err2[d] = e2[d];
    }
    return AAR_OK;
}

extern "C" int aar_vote_transforms(double marker_size, int64_t n_sets, const int64_t *set_begin, const double *T,
                                   const double *T1inv, const double *T2inv, int64_t *best, double *weight, double *cost,
                                   int32_t device_id) {
    if (n_sets < 0 || (n_sets > 0 && (!set_begin || !best))) return set_error(AAR_ERR_INVALID, "aar_vote_transforms: bad argument");
    const int64_t n = n_sets ? set_begin[n_sets] : 0;
    if (n > 0 && (!T || !T1inv || !T2inv)) return set_error(AAR_ERR_INVALID, "aar_vote_transforms: null matrices");
    if (int rc = check_sets(n, n_sets, set_begin)) return rc;
    if (int rc = select_device(device_id)) return rc;
    if (n_sets == 0) return AAR_OK;
    std::vector<double> t12(12 * (size_t)n), a12(12 * (size_t)n), b12(12 * (size_t)n);
    top3x4(T, t12.data(), n); top3x4(T1inv, a12.data(), n); top3x4(T2inv, b12.data(), n);
    DevBuf<double> d_T, d_A, d_B, d_BJ;
    HIPCHK(d_T.upload(t12.data(), t12.size()), "upload(T)");
    HIPCHK(d_A.upload(a12.data(), a12.size()), "upload(T1inv)");
    HIPCHK(d_B.upload(b12.data(), b12.size()), "upload(T2inv)");
    HIPCHK(d_BJ.alloc(24 * (size_t)n), "hipMalloc(j-side)");
    if (n) {
        hipLaunchKernelGGL(k_prep_jside, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (long long)n, d_A.p, d_B.p,
                           marker_size / 2, d_BJ.p);
        HIPCHK(hipGetLastError(), "k_prep_jside");
    }
    return vote_sets(n, d_T.p, d_BJ.p, n_sets, set_begin, marker_size, best, weight, nullptr, cost);
}
