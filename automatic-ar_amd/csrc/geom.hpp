// Device-side geometry of one marker observation: pinhole projection of the four marker corners
// (MultiCamMapper::project_marker, libs/multicam_mapper.cpp:608-649), the residual rows
// (eval_curr_solution, :1008-1025) and the closed-form 2x6 SE(3) Jacobian rows that replace the
// reference's central differences (obtain_marker_derivs, :976-994) -- SURVEY.md Appendix A.
//
// Every entity (camera, marker, frame) is a row of the `ent` table: R (9, row-major), t (3), and the SO(3)
// left Jacobian J_l(omega) (9), so that d(R(omega) y)/d omega = -[R y]x J_l(omega).
#pragma once
#include <hip/hip_runtime.h>

namespace aar {

constexpr int ENT_STRIDE = 24;  // doubles per entity row: R[9] t[3] Jl[9] pad[3]

struct Ent {
    double R[9], t[3], Jl[9];
};

__device__ __forceinline__ void load_ent(const double *__restrict__ tab, int idx, Ent &e) {
    const double2 *p = reinterpret_cast<const double2 *>(tab + (size_t)idx * ENT_STRIDE);
    double buf[22];
#pragma unroll
    for (int i = 0; i < 11; i++) {
        double2 v = p[i];
        buf[2 * i] = v.x;
        buf[2 * i + 1] = v.y;
    }
#pragma unroll
    for (int i = 0; i < 9; i++) e.R[i] = buf[i];
#pragma unroll
    for (int i = 0; i < 3; i++) e.t[i] = buf[9 + i];
#pragma unroll
    for (int i = 0; i < 9; i++) e.Jl[i] = buf[12 + i];
}

// R, t only (residual pass)
struct EntRT {
    double R[9], t[3];
};
__device__ __forceinline__ void load_ent_rt(const double *__restrict__ tab, int idx, EntRT &e) {
    const double2 *p = reinterpret_cast<const double2 *>(tab + (size_t)idx * ENT_STRIDE);
    double buf[12];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double2 v = p[i];
        buf[2 * i] = v.x;
        buf[2 * i + 1] = v.y;
    }
#pragma unroll
    for (int i = 0; i < 9; i++) e.R[i] = buf[i];
#pragma unroll
    for (int i = 0; i < 3; i++) e.t[i] = buf[9 + i];
}

// cv::Rodrigues vector -> matrix plus the left Jacobian; one entity row from a 6-vector
__device__ __forceinline__ void make_ent_row(const double *__restrict__ v, double *__restrict__ row) {
    const double wx = v[0], wy = v[1], wz = v[2];
    const double t2 = wx * wx + wy * wy + wz * wz;
    const double th = sqrt(t2);
    double R[9];
    if (th < 2.2204460492503131e-16) {  // DBL_EPSILON, as cv::Rodrigues
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    } else {
        double s, c;
        sincos(th, &s, &c);
        const double c1 = 1.0 - c, ith = 1.0 / th;
        const double x = wx * ith, y = wy * ith, z = wz * ith;
        R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
        R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
        R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
    }
    // J_l = I + A [w]x + B [w]x^2,  A = (1-cos)/th^2,  B = (th - sin)/th^3
    double A, B;
    if (th < 1e-2) {
        A = 0.5 - t2 * (1.0 / 24.0) + t2 * t2 * (1.0 / 720.0);
        B = (1.0 / 6.0) - t2 * (1.0 / 120.0) + t2 * t2 * (1.0 / 5040.0);
    } else {
        double s, c;          // the same sincos as above (the compiler merges the two calls)
        sincos(th, &s, &c);
        const double it2 = 1.0 / t2;
        A = (1.0 - c) * it2;
        B = (th - s) * it2 / th;
    }
    double J[9];
    // [w]x^2 = w w^T - |w|^2 I
    J[0] = 1.0 + B * (wx * wx - t2); J[1] = -A * wz + B * wx * wy;   J[2] = A * wy + B * wx * wz;
    J[3] = A * wz + B * wx * wy;     J[4] = 1.0 + B * (wy * wy - t2); J[5] = -A * wx + B * wy * wz;
    J[6] = -A * wy + B * wx * wz;    J[7] = A * wx + B * wy * wz;     J[8] = 1.0 + B * (wz * wz - t2);
#pragma unroll
    for (int i = 0; i < 9; i++) row[i] = R[i];
    row[9] = v[3]; row[10] = v[4]; row[11] = v[5];
#pragma unroll
    for (int i = 0; i < 9; i++) row[12 + i] = J[i];
    row[21] = 0; row[22] = 0; row[23] = 0;
}

// Row of an INTRINSICS entity (optimize_cam_intrinsics): the pinhole matrix intrinsics_vec2mats rebuilds from
// (fx, cx, fy, cy) -- cv::Mat::eye with those four entries, libs/multicam_mapper.cpp:580-593 -- in the first nine doubles,
// so that the kernels read K from the entity table exactly as they read it from the constant table otherwise.
__device__ __forceinline__ void make_k_row(const double *__restrict__ v, double *__restrict__ row) {
    row[0] = v[0]; row[1] = 0; row[2] = v[1];
    row[3] = 0; row[4] = v[2]; row[5] = v[3];
    row[6] = 0; row[7] = 0; row[8] = 1;
#pragma unroll
    for (int i = 9; i < ENT_STRIDE; i++) row[i] = 0;
}

// corner k of the marker model: (-h,h,0) (h,h,0) (h,-h,0) (-h,-h,0)  (aruco marker.cpp:358-367)
__device__ __forceinline__ double corner_sx(int k) { return (k == 1 || k == 2) ? 1.0 : -1.0; }
__device__ __forceinline__ double corner_sy(int k) { return (k < 2) ? 1.0 : -1.0; }

// Projection of corner k: returns (u,v) and optionally the intermediate vectors the Jacobian needs.
struct CornerGeom {
    double ym[3], yf[3], yc[3];  // R_m X ; R_f (R_m X + t_m) ; s - t_c
    double u, v, iw;
    double pc[3];
};

template <class EC, class EM, class EF>
__device__ __forceinline__ void project_corner(const EC &ec, const EM &em, const EF &ef, const double *__restrict__ K,
                                               double h, int k, CornerGeom &g) {
    const double sx = corner_sx(k) * h, sy = corner_sy(k) * h;
    double q[3], s[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        g.ym[i] = em.R[i * 3] * sx + em.R[i * 3 + 1] * sy;
        q[i] = g.ym[i] + em.t[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        g.yf[i] = ef.R[i * 3] * q[0] + ef.R[i * 3 + 1] * q[1] + ef.R[i * 3 + 2] * q[2];
        s[i] = g.yf[i] + ef.t[i];
        g.yc[i] = s[i] - ec.t[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) g.pc[i] = ec.R[i] * g.yc[0] + ec.R[3 + i] * g.yc[1] + ec.R[6 + i] * g.yc[2];
    const double hx = K[0] * g.pc[0] + K[1] * g.pc[1] + K[2] * g.pc[2];
    const double hy = K[3] * g.pc[0] + K[4] * g.pc[1] + K[5] * g.pc[2];
    const double hw = K[6] * g.pc[0] + K[7] * g.pc[1] + K[8] * g.pc[2];
    g.iw = 1.0 / hw;
    g.u = hx * g.iw;
    g.v = hy * g.iw;
}

// residual of one corner: observed - projected.  f32 mode reproduces the reference's cv::Point2f store and
// float subtraction (libs/multicam_mapper.cpp:644-647,1012-1013).
// huber >= 0: both rows are scaled by w = sqrt(rho(e)/e), e = rx^2 + ry^2, rho(e) = e if e <= delta^2 else 2 delta sqrt(e) - delta^2
// with delta^2 and 2 delta rounded to float (hubberMono / getHubberMonoWeight, libs/multicam_mapper.cpp:11-24,1014-1019).
// The Jacobian is NOT re-weighted, as in the reference (:976-994).
__device__ __forceinline__ void corner_residual(float ou, float ov, double u, double v, int res_f32, float huber, double &rx,
                                                double &ry) {
    if (res_f32) {
        rx = (double)(ou - (float)u);
        ry = (double)(ov - (float)v);
    } else {
        rx = (double)ou - u;
        ry = (double)ov - v;
    }
    if (huber >= 0.f) {
        const double e = rx * rx + ry * ry;
        if (e != 0.0) {
            const float dsq = huber * huber, d2 = 2 * huber;
            const double rho = (e <= (double)dsq) ? e : (double)d2 * sqrt(e) - (double)dsq;
            const double w = sqrt(rho / e);
            rx *= w;
            ry *= w;
        }
    }
}

__device__ __forceinline__ void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// Jacobian rows of (u,v) for corner geometry g: Gc/Gm/Gf are [2][6] (row 0 = u, row 1 = v);
// d r/d theta = -G, so J^T J = G^T G and B = -J^T r = G^T r.
// d(u,v)/d(fx, cx, fy, cy) for the pinhole matrix [fx 0 cx; 0 fy cy; 0 0 1] of an intrinsics entity: u = fx x/w + cx
__device__ __forceinline__ void corner_jacobian_intr(const CornerGeom &g, double Gk[2][4]) {
    Gk[0][0] = g.pc[0] * g.iw; Gk[0][1] = 1.0; Gk[0][2] = 0.0; Gk[0][3] = 0.0;
    Gk[1][0] = 0.0; Gk[1][1] = 0.0; Gk[1][2] = g.pc[1] * g.iw; Gk[1][3] = 1.0;
}

template <bool WANT_C, bool WANT_M, bool WANT_F>
__device__ __forceinline__ void corner_jacobian(const Ent &ec, const Ent &em, const Ent &ef,
                                                const double *__restrict__ K, const CornerGeom &g, double Gc[2][6],
                                                double Gm[2][6], double Gf[2][6]) {
    double A[2][3], B[2][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        A[0][j] = (K[j] - g.u * K[6 + j]) * g.iw;
        A[1][j] = (K[3 + j] - g.v * K[6 + j]) * g.iw;
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) B[r][j] = A[r][0] * ec.R[j * 3] + A[r][1] * ec.R[j * 3 + 1] + A[r][2] * ec.R[j * 3 + 2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (WANT_F) {
            double w[3];
            cross3(g.yf, B[r], w);  // B_r . (c_j x yf) = c_j . (yf x B_r)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Gf[r][j] = w[0] * ef.Jl[j] + w[1] * ef.Jl[3 + j] + w[2] * ef.Jl[6 + j];
                Gf[r][3 + j] = B[r][j];
            }
        }
        if (WANT_C) {
            double w[3];
            cross3(B[r], g.yc, w);  // B_r . (yc x c_j) = c_j . (B_r x yc)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Gc[r][j] = w[0] * ec.Jl[j] + w[1] * ec.Jl[3 + j] + w[2] * ec.Jl[6 + j];
                Gc[r][3 + j] = -B[r][j];
            }
        }
        if (WANT_M) {
            double BR[3], w[3];
#pragma unroll
            for (int j = 0; j < 3; j++) BR[j] = B[r][0] * ef.R[j] + B[r][1] * ef.R[3 + j] + B[r][2] * ef.R[6 + j];
            cross3(g.ym, BR, w);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Gm[r][j] = w[0] * em.Jl[j] + w[1] * em.Jl[3 + j] + w[2] * em.Jl[6 + j];
                Gm[r][3 + j] = BR[j];
            }
        }
    }
}

// The two rows of a corner as wrenches about the frame's origin: w_r = (yf x B_r, B_r), B_r = d(u,v)_r / d(world point) -- the same A, B as
// corner_jacobian.  All three Jacobian blocks of the row are LINEAR images of w_r by matrices that depend on the entities only:
//   G_f = w^T F,    F  = [J_l(f) 0; 0 I]
//   G_c = w^T T_c,  T_c = -[J_l(c) 0; [d]x J_l(c) I],  d = t_c - t_f                     (yc = yf - d)
//   G_m = w^T T_m,  T_m = [R_f J_l(m) 0; R_f [t_m]x J_l(m) R_f]                           (ym = R_f^T yf - t_m)
// so a frame's blocks are  V_f = F^T (sum H) F,  W_cf = T_c^T H_cf F,  W_mf = T_m^T H_mf F  with H = sum w w^T over the rows of a slot:
// 21 accumulators per observation instead of 36 + 36 + 21, and the 6x6 images once per slot instead of once per row.
template <class EC>
__device__ __forceinline__ void corner_wrench(const EC &ec, const double *__restrict__ K, const CornerGeom &g, double w[2][6]) {
    double A[2][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        A[0][j] = (K[j] - g.u * K[6 + j]) * g.iw;
        A[1][j] = (K[3 + j] - g.v * K[6 + j]) * g.iw;
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int j = 0; j < 3; j++) w[r][3 + j] = A[r][0] * ec.R[j * 3] + A[r][1] * ec.R[j * 3 + 1] + A[r][2] * ec.R[j * 3 + 2];
        cross3(g.yf, &w[r][3], &w[r][0]);
    }
}

// 1/d: v_rcp_f64 is good to 2^-24.4 (scripts/probe/rcp_probe.hip); one cubic step 1/d = x (1 + e + e^2 + ...), e = 1 - d x,
// leaves 2^-73 and matches the IEEE quotient on 4 M samples -- one instruction less than two Newton steps
__device__ __forceinline__ double rcp_refined(double d) {
    const double x = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, x, 1.0);
    return fma(x, fma(e, e, e), x);
}

// 6x6 SPD inverse, one thread, registers only: out = a^-1 (row-major, both triangles), returns false on a non-positive
// pivot.  LDL^T instead of Cholesky (no square roots, the six divisions as refined reciprocals), the unit factor inverted in
// place, and only the 21 distinct entries of L^-T D^-1 L^-1 formed: ~170 fp64 instructions instead of ~460 -- this runs on
// ONE lane at the end of every frame's pass A workgroup, i.e. on the critical path of the kernel.
__device__ __forceinline__ bool spd6_inverse(double a[6][6], double out[36]) {
    bool ok = true;
    double dinv[6];
#pragma unroll
    for (int p = 0; p < 6; p++) {
        double d = a[p][p];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        dinv[p] = rcp_refined(d);
        double l[6];
#pragma unroll
        for (int q = p + 1; q < 6; q++) l[q] = a[q][p] * dinv[p];
#pragma unroll
        for (int q = p + 1; q < 6; q++)
#pragma unroll
            for (int r = p + 1; r <= q; r++) a[q][r] = fma(-l[q], a[r][p], a[q][r]);
#pragma unroll
        for (int q = p + 1; q < 6; q++) a[q][p] = l[q];   // unit L, strictly lower
    }
    // M = L^-1 (unit lower): column c by forward substitution, M[i][c] = -(L[i][c] + sum_{c<p<i} L[i][p] M[p][c])
    double m[6][6];
#pragma unroll
    for (int c = 0; c < 6; c++)
#pragma unroll
        for (int i = c + 1; i < 6; i++) {
            double s = a[i][c];
#pragma unroll
            for (int p = c + 1; p < i; p++) s = fma(a[i][p], m[p][c], s);
            m[i][c] = -s;
        }
    // a^-1 = M^T D^-1 M: entry (i, j), i >= j: sum_{p >= i} M[p][i] dinv_p M[p][j]   (M[p][p] = 1)
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = (i == j) ? dinv[i] : dinv[i] * m[i][j];
#pragma unroll
            for (int p = i + 1; p < 6; p++) s = fma(m[p][i] * dinv[p], m[p][j], s);
            out[i * 6 + j] = s;
            out[j * 6 + i] = s;
        }
    return ok;
}

// packed lower-triangular index of a symmetric 6x6: (i>=j) -> i(i+1)/2 + j
__device__ __host__ __forceinline__ constexpr int sym6(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

}  // namespace aar
