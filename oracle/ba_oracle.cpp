/*
 * oracle/ba_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ba_oracle.h).
 *
 * Plain C++ restatement of the reference hot path.  Written for clarity and to mirror the
 * reference's evaluation order, not for speed; it shares no source with automatic-ar_amd/.
 * File:line citations are into /root/reference.
 */
#include "ba_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Mat4 {  // stand-in for a 4x4 CV_64F cv::Mat
    double a[16];
    static Mat4 eye() {
        Mat4 m;
        for (int i = 0; i < 16; i++) m.a[i] = (i % 5 == 0) ? 1.0 : 0.0;
        return m;
    }
    double &operator()(int r, int c) { return a[r * 4 + c]; }
    double operator()(int r, int c) const { return a[r * 4 + c]; }
};

Mat4 mul(const Mat4 &x, const Mat4 &y) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += x(i, k) * y(k, j);
            r(i, j) = s;
        }
    return r;
}

// cv::Mat::inv() default (DECOMP_LU): general 4x4 inverse by elimination with partial pivoting.
// Used at libs/multicam_mapper.cpp:619,621 on rigid transforms.
Mat4 inv_lu(const Mat4 &m) {
    double w[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            w[i][j] = m(i, j);
            w[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (piv != c)
            for (int j = 0; j < 8; j++) std::swap(w[c][j], w[piv][j]);
        double d = 1.0 / w[c][c];
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            double f = w[r][c] * d;
            if (f != 0.0)
                for (int j = c; j < 8; j++) w[r][j] -= f * w[c][j];
        }
        for (int j = c; j < 8; j++) w[c][j] *= d;
    }
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r(i, j) = w[i][4 + j];
    return r;
}

// ---- cv::Rodrigues, vector -> matrix ----
void rodrigues_v2m(const double w[3], double R[9]) {
    double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c;
    double it = 1.0 / theta;
    double x = w[0] * it, y = w[1] * it, z = w[2] * it;
    // R = c*I + (1-c)*n*n^T + s*[n]x
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

double det3(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

void inv3_transposed(const double *m, double *o) {  // o = (m^-1)^T
    double d = det3(m), id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id;
    o[1] = (m[5] * m[6] - m[3] * m[8]) * id;
    o[2] = (m[3] * m[7] - m[4] * m[6]) * id;
    o[3] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[6] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[7] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// ---- cv::Rodrigues, matrix -> vector.  OpenCV first replaces R by U*V^T of its SVD (the nearest
// orthogonal matrix); here that projection is computed by the Newton polar iteration, which
// converges to the same U*V^T. ----
void rodrigues_m2v(const double Rin[9], double w[3]) {
    double R[9];
    std::memcpy(R, Rin, sizeof R);
    for (int it = 0; it < 32; it++) {
        double T[9], d = 0;
        inv3_transposed(R, T);
        for (int i = 0; i < 9; i++) {
            double n = 0.5 * (R[i] + T[i]);
            d = std::max(d, std::fabs(n - R[i]));
            R[i] = n;
        }
        if (d < 1e-16) break;
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            w[0] = w[1] = w[2] = 0;
            return;
        }
        double t;
        t = (R[0] + 1) * 0.5; rx = std::sqrt(std::max(t, 0.0));
        t = (R[4] + 1) * 0.5; ry = std::sqrt(std::max(t, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1) * 0.5; rz = std::sqrt(std::max(t, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && ((R[5] > 0) != (ry * rz > 0)))
            rz = -rz;
        double n = std::sqrt(rx * rx + ry * ry + rz * rz);
        double k = theta / n;
        w[0] = rx * k; w[1] = ry * k; w[2] = rz * k;
        return;
    }
    double k = theta / (2.0 * s);
    w[0] = rx * k; w[1] = ry * k; w[2] = rz * k;
}

// vec2transformation_mat: libs/multicam_mapper.cpp:463-473
Mat4 pose_to_mat(const double *v) {
    Mat4 m = Mat4::eye();
    double R[9];
    rodrigues_v2m(v, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) m(i, j) = R[i * 3 + j];
        m(i, 3) = v[3 + i];
    }
    return m;
}

struct Layout {  // offsets into x_full and into z (libs/multicam_mapper.cpp:445-461)
    int C, M, F, rc, rm;
    int64_t full_cam0, full_mk0, full_fr0, full_len;
    int64_t z_cam0, z_mk0, z_fr0, z_intr0, z_len;  // -1 when the group is not optimised
    int jac_slab;                                  // triplet slots per observation: 8 x 18 (+ 8 x 9 with intrinsics)
    explicit Layout(const orc_problem *p) {
        C = p->num_cams; M = p->num_markers; F = p->num_frames; rc = p->root_cam; rm = p->root_marker;
        full_cam0 = 0; full_mk0 = 6LL * (C - 1); full_fr0 = full_mk0 + 6LL * (M - 1);
        full_len = full_fr0 + 6LL * F;
        int64_t o = 0;
        z_cam0 = z_mk0 = z_fr0 = -1;
        if (p->opt_cams) { z_cam0 = o; o += 6LL * (C - 1); }
        if (p->opt_markers) { z_mk0 = o; o += 6LL * (M - 1); }
        if (p->opt_frames) { z_fr0 = o; o += 6LL * F; }
        z_intr0 = -1;
        if (p->opt_intrinsics) { z_intr0 = o; o += 9LL * C; }   // fill_io_vec_cam_intrinsics: ALL cameras, the root too (:488-498)
        z_len = o;
        jac_slab = 144 + (p->opt_intrinsics ? 72 : 0);
    }
    // position of entity within its (root-skipping) group, -1 for the root
    int cam_slot(int c) const { return c == rc ? -1 : (c < rc ? c : c - 1); }
    int mk_slot(int m) const { return m == rm ? -1 : (m < rm ? m : m - 1); }
};

// eVec2Mats (libs/multicam_mapper.cpp:595-606): all transforms of the problem for a given z
struct Mats {
    std::vector<Mat4> cam, mk, fr;
};

void build_full(const orc_problem *p, const Layout &L, const double *x_full, const double *z, std::vector<double> &x) {
    x.assign(x_full, x_full + L.full_len);
    if (!z) return;
    if (L.z_cam0 >= 0) std::memcpy(&x[L.full_cam0], z + L.z_cam0, sizeof(double) * 6 * (L.C - 1));
    if (L.z_mk0 >= 0) std::memcpy(&x[L.full_mk0], z + L.z_mk0, sizeof(double) * 6 * (L.M - 1));
    if (L.z_fr0 >= 0) std::memcpy(&x[L.full_fr0], z + L.z_fr0, sizeof(double) * 6 * L.F);
}

void build_mats(const Layout &L, const std::vector<double> &x, Mats &ma) {
    ma.cam.assign(L.C, Mat4::eye());
    ma.mk.assign(L.M, Mat4::eye());
    ma.fr.assign(L.F, Mat4::eye());
    for (int c = 0; c < L.C; c++)
        if (c != L.rc) ma.cam[c] = pose_to_mat(&x[L.full_cam0 + 6LL * L.cam_slot(c)]);
    for (int m = 0; m < L.M; m++)
        if (m != L.rm) ma.mk[m] = pose_to_mat(&x[L.full_mk0 + 6LL * L.mk_slot(m)]);
    for (int f = 0; f < L.F; f++) ma.fr[f] = pose_to_mat(&x[L.full_fr0 + 6LL * f]);
}

// camera matrix of camera c: the data set's, or -- with the intrinsics in z -- what intrinsics_vec2mats builds from it
// (libs/multicam_mapper.cpp:580-593: cv::Mat::eye with fx, cx, fy, cy; any skew of the calibration is gone)
void cam_matrix(const orc_problem *p, const Layout &L, const double *z, int c, double K[9]) {
    if (z && L.z_intr0 >= 0) {
        const double *q = z + L.z_intr0 + 9LL * c;
        K[0] = q[0]; K[1] = 0; K[2] = q[1]; K[3] = 0; K[4] = q[2]; K[5] = q[3]; K[6] = 0; K[7] = 0; K[8] = 1;
    } else {
        std::memcpy(K, p->K + 9 * c, sizeof(double) * 9);
    }
}

// project_marker: libs/multicam_mapper.cpp:608-649.  out = 4 corners (x,y) in double, before the
// cv::Point2f store.
void project_marker(const orc_problem *p, const Layout &L, const Mat4 &Tc, const Mat4 &Tf, const Mat4 &Tm, int c,
                    int m, const double *K, double out[8]) {
    Mat4 T = Tf;
    if (c != L.rc) T = mul(inv_lu(Tc), T);
    if (m != L.rm) T = mul(T, Tm);
    double KT[12];  // cam_mat * transform.rowRange(0,3)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * T(k, j);
            KT[i * 4 + j] = s;
        }
    // marker_points_3d_mat (libs/multicam_mapper.cpp:261-270; aruco marker.cpp:358-367)
    float hs = (float)p->marker_size / 2.f;
    double h = (double)hs;
    const double X[4][4] = {{-h, h, 0, 1}, {h, h, 0, 1}, {h, -h, 0, 1}, {-h, -h, 0, 1}};
    for (int k = 0; k < 4; k++) {
        double v[3];
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int j = 0; j < 4; j++) s += KT[i * 4 + j] * X[k][j];
            v[i] = s;
        }
        out[2 * k] = v[0] / v[2];
        out[2 * k + 1] = v[1] / v[2];
    }
}

// hubberMono / getHubberMonoWeight: libs/multicam_mapper.cpp:11-24
double huber_weight(double e, float delta) {
    if (e == 0) return 1;
    float dsq = delta * delta, d2 = 2 * delta;
    double rho = (e <= dsq) ? e : d2 * std::sqrt(e) - dsq;
    return std::sqrt(rho / e);
}

// residual rows of one observation (libs/multicam_mapper.cpp:1008-1025)
void obs_residual(const orc_problem *p, const double proj[8], const float *uv, int res_mode, double r[8]) {
    for (int k = 0; k < 4; k++) {
        double ex, ey;
        if (res_mode == ORC_RES_F32) {
            float px = (float)proj[2 * k], py = (float)proj[2 * k + 1];  // cv::Point2f store
            ex = (double)(uv[2 * k] - px);                               // float subtraction
            ey = (double)(uv[2 * k + 1] - py);
        } else {
            ex = (double)uv[2 * k] - proj[2 * k];
            ey = (double)uv[2 * k + 1] - proj[2 * k + 1];
        }
        if (p->with_huber) {
            double w = huber_weight(ex * ex + ey * ey, p->huber_delta);
            ex *= w; ey *= w;
        }
        r[2 * k] = ex; r[2 * k + 1] = ey;
    }
}

// ---------- analytic Jacobian pieces ("Analytic replacement") ----------
void left_jacobian_so3(const double w[3], double Jl[9]) {
    double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
    double A, B;  // A = (1-cos t)/t^2, B = (t - sin t)/t^3
    if (t < 1e-2) {
        A = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
        B = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        A = (1.0 - std::cos(t)) / t2;
        B = (t - std::sin(t)) / (t2 * t);
    }
    double Wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Wx[i * 3 + k] * Wx[k * 3 + j];
            W2[i * 3 + j] = s;
        }
    for (int i = 0; i < 9; i++) Jl[i] = ((i % 4 == 0) ? 1.0 : 0.0) + A * Wx[i] + B * W2[i];
}

inline void cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

struct Ent {  // R, t, left Jacobian of one SE(3) entity; identity for roots
    double R[9], t[3], Jl[9];
};

void make_ent(const double *v, Ent &e) {
    rodrigues_v2m(v, e.R);
    e.t[0] = v[3]; e.t[1] = v[4]; e.t[2] = v[5];
    left_jacobian_so3(v, e.Jl);
}

void ident_ent(Ent &e) {
    for (int i = 0; i < 9; i++) e.R[i] = e.Jl[i] = (i % 4 == 0) ? 1.0 : 0.0;
    e.t[0] = e.t[1] = e.t[2] = 0;
}

inline void matvec(const double *A, const double *x, double *y) {
    for (int i = 0; i < 3; i++) y[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
}
inline void matTvec(const double *A, const double *x, double *y) {
    for (int i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

// d(u,v)/d(params) for one observation: Gc, Gm, Gf are 8x6 row-major (rows = 2*corner + {x,y});
// also the double-precision projection. dr/dparam = -G.
// Gk (8x4, optional): d(u,v)/d(fx, cx, fy, cy) of the pinhole matrix [fx 0 cx; 0 fy cy; 0 0 1]
void analytic_blocks(const orc_problem *p, const Ent &ec, const Ent &em, const Ent &ef, const double *K, double proj[8],
                     double Gc[48], double Gm[48], double Gf[48], double *Gk = nullptr) {
    float hs = (float)p->marker_size / 2.f;
    double h = (double)hs;
    const double X[4][3] = {{-h, h, 0}, {h, h, 0}, {h, -h, 0}, {-h, -h, 0}};
    for (int k = 0; k < 4; k++) {
        double ym[3], q[3], yf[3], s[3], yc[3], pc[3];
        matvec(em.R, X[k], ym);
        for (int i = 0; i < 3; i++) q[i] = ym[i] + em.t[i];
        matvec(ef.R, q, yf);
        for (int i = 0; i < 3; i++) s[i] = yf[i] + ef.t[i];
        for (int i = 0; i < 3; i++) yc[i] = s[i] - ec.t[i];
        matTvec(ec.R, yc, pc);
        double hx[3];
        matvec(K, pc, hx);
        double iw = 1.0 / hx[2], u = hx[0] * iw, v = hx[1] * iw;
        proj[2 * k] = u; proj[2 * k + 1] = v;
        if (Gk) {   // u = fx x/w + cx, v = fy y/w + cy
            double *g = Gk + 8 * k;
            g[0] = pc[0] * iw; g[1] = 1; g[2] = 0; g[3] = 0;
            g[4] = 0; g[5] = 0; g[6] = pc[1] * iw; g[7] = 1;
        }
        double A[6];  // d(u,v)/dp
        for (int j = 0; j < 3; j++) {
            A[j] = (K[j] - u * K[6 + j]) * iw;
            A[3 + j] = (K[3 + j] - v * K[6 + j]) * iw;
        }
        double Bm[6];  // A * Rc^T
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < 3; j++)
                Bm[r * 3 + j] = A[r * 3] * ec.R[j * 3] + A[r * 3 + 1] * ec.R[j * 3 + 1] + A[r * 3 + 2] * ec.R[j * 3 + 2];
        double BRf[6];  // B * Rf
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < 3; j++)
                BRf[r * 3 + j] = Bm[r * 3] * ef.R[j] + Bm[r * 3 + 1] * ef.R[3 + j] + Bm[r * 3 + 2] * ef.R[6 + j];
        for (int j = 0; j < 3; j++) {
            double cf[3] = {ef.Jl[j], ef.Jl[3 + j], ef.Jl[6 + j]};
            double cm[3] = {em.Jl[j], em.Jl[3 + j], em.Jl[6 + j]};
            double cc[3] = {ec.Jl[j], ec.Jl[3 + j], ec.Jl[6 + j]};
            double vf[3], vm[3], vc[3];
            cross(cf, yf, vf);  // -[yf]x Jl_f e_j
            cross(cm, ym, vm);
            cross(yc, cc, vc);  // +[yc]x Jl_c e_j
            for (int r = 0; r < 2; r++) {
                Gf[(2 * k + r) * 6 + j] = Bm[r * 3] * vf[0] + Bm[r * 3 + 1] * vf[1] + Bm[r * 3 + 2] * vf[2];
                Gm[(2 * k + r) * 6 + j] = BRf[r * 3] * vm[0] + BRf[r * 3 + 1] * vm[1] + BRf[r * 3 + 2] * vm[2];
                Gc[(2 * k + r) * 6 + j] = Bm[r * 3] * vc[0] + Bm[r * 3 + 1] * vc[1] + Bm[r * 3 + 2] * vc[2];
                Gf[(2 * k + r) * 6 + 3 + j] = Bm[r * 3 + j];
                Gm[(2 * k + r) * 6 + 3 + j] = BRf[r * 3 + j];
                Gc[(2 * k + r) * 6 + 3 + j] = -Bm[r * 3 + j];
            }
        }
    }
}

struct Triplets {
    int32_t *rows, *cols;
    double *vals;
    int64_t n;
};

int64_t jacobian_impl(const orc_problem *p, const double *x_full, const double *z, int jac_mode, int32_t *rows,
                      int32_t *cols, double *vals) {
    Layout L(p);
    std::vector<double> x;
    build_full(p, L, x_full, z, x);
    const int64_t N = p->num_obs;
    // every observation owns a fixed slab of 8*18 (+ 8*9) triplet slots so threads never collide
    const int64_t slab = L.jac_slab;
    std::vector<int64_t> cnt(N, 0);

    if (jac_mode == ORC_JAC_ANALYTIC) {
        std::vector<Ent> ec(L.C), em(L.M), ef(L.F);
        for (int c = 0; c < L.C; c++) { if (c == L.rc) ident_ent(ec[c]); else make_ent(&x[L.full_cam0 + 6LL * L.cam_slot(c)], ec[c]); }
        for (int m = 0; m < L.M; m++) { if (m == L.rm) ident_ent(em[m]); else make_ent(&x[L.full_mk0 + 6LL * L.mk_slot(m)], em[m]); }
        for (int f = 0; f < L.F; f++) make_ent(&x[L.full_fr0 + 6LL * f], ef[f]);
#pragma omp parallel for schedule(static)
        for (int64_t o = 0; o < N; o++) {
            int c = p->obs_cam[o], m = p->obs_marker[o], f = p->obs_frame[o];
            double proj[8], Gc[48], Gm[48], Gf[48], Gk[32], Kc[9];
            cam_matrix(p, L, z, c, Kc);
            analytic_blocks(p, ec[c], em[m], ef[f], Kc, proj, Gc, Gm, Gf, L.z_intr0 >= 0 ? Gk : nullptr);
            int64_t base = o * slab, n = 0;
            auto emit = [&](const double *G, int64_t col0) {
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 6; j++) {
                        rows[base + n] = (int32_t)(8 * o + r);
                        cols[base + n] = (int32_t)(col0 + j);
                        vals[base + n] = -G[r * 6 + j];
                        n++;
                    }
            };
            if (L.z_cam0 >= 0 && c != L.rc) emit(Gc, L.z_cam0 + 6LL * L.cam_slot(c));
            if (L.z_mk0 >= 0 && m != L.rm) emit(Gm, L.z_mk0 + 6LL * L.mk_slot(m));
            if (L.z_fr0 >= 0) emit(Gf, L.z_fr0 + 6LL * f);
            if (L.z_intr0 >= 0)   // 9 columns per camera: 4 live, the 5 distortion columns explicit zeros as in the reference
                for (int r = 0; r < 8; r++)
                    for (int j = 0; j < 9; j++) {
                        rows[base + n] = (int32_t)(8 * o + r);
                        cols[base + n] = (int32_t)(L.z_intr0 + 9LL * c + j);
                        vals[base + n] = j < 4 ? -Gk[(r / 2) * 8 + (r % 2) * 4 + j] : 0.0;
                        n++;
                    }
            cnt[o] = n;
        }
    } else {
        // obtain_transformation_derivs / obtain_marker_derivs: libs/multicam_mapper.cpp:803-994.
        // Central differences, rotation perturbed in Rodrigues-vector space (:905-911), translation
        // perturbed directly in the matrix (:913-916).
        const bool f32 = (jac_mode == ORC_JAC_NUMERIC_F32), track = (jac_mode == ORC_JAC_TRACK);
        const double delta = (f32 || track) ? 1e-3 : 1e-6;  // J_delta, libs/multicam_mapper.h:189 ; der_epsilon, libs/sparselevmarq.h:47
        Mats ma;
        build_mats(L, x, ma);
#pragma omp parallel for schedule(static)
        for (int64_t o = 0; o < N; o++) {
            int c = p->obs_cam[o], m = p->obs_marker[o], f = p->obs_frame[o];
            const float *uv = p->obs_uv + 8 * o;
            int64_t base = o * slab, n = 0;
            double Kc[9];
            cam_matrix(p, L, z, c, Kc);
            auto quotient = [&](const double pa[8], const double ps[8], int64_t col) {   // obtain_marker_derivs, :976-994
                double ra[8], rs[8];
                if (track) {  // calcDerivates differentiates the whole error function, Huber weights included
                    obs_residual(p, pa, uv, ORC_RES_F64, ra);
                    obs_residual(p, ps, uv, ORC_RES_F64, rs);
                }
                for (int k = 0; k < 8; k++) {
                    double ea, es;
                    if (track) {
                        ea = ra[k]; es = rs[k];
                    } else if (f32) {
                        ea = (double)(uv[k] - (float)pa[k]);
                        es = (double)(uv[k] - (float)ps[k]);
                    } else {
                        ea = (double)uv[k] - pa[k];
                        es = (double)uv[k] - ps[k];
                    }
                    const double dv = (ea - es) / (2 * delta);
                    if (track && !(std::fabs(dv) > 1e-4)) continue;  // libs/sparselevmarq.h:182,215
                    rows[base + n] = (int32_t)(8 * o + k);
                    cols[base + n] = (int32_t)col;
                    vals[base + n] = dv;
                    n++;
                }
            };
            auto block = [&](int which, const double *pose, int64_t col0) {
                const Mat4 &T0 = which == 0 ? ma.cam[c] : (which == 1 ? ma.mk[m] : ma.fr[f]);
                for (int i = 0; i < 6; i++) {
                    Mat4 Ta = T0, Ts = T0;
                    if (i < 3) {
                        double ra[3] = {pose[0], pose[1], pose[2]}, rs[3] = {pose[0], pose[1], pose[2]};
                        ra[i] += delta; rs[i] -= delta;
                        double Ra[9], Rs[9];
                        rodrigues_v2m(ra, Ra); rodrigues_v2m(rs, Rs);
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) { Ta(a, b) = Ra[a * 3 + b]; Ts(a, b) = Rs[a * 3 + b]; }
                    } else {
                        Ta(i - 3, 3) += delta; Ts(i - 3, 3) -= delta;
                    }
                    double pa[8], ps[8];
                    project_marker(p, L, which == 0 ? Ta : ma.cam[c], which == 2 ? Ta : ma.fr[f], which == 1 ? Ta : ma.mk[m], c, m, Kc, pa);
                    project_marker(p, L, which == 0 ? Ts : ma.cam[c], which == 2 ? Ts : ma.fr[f], which == 1 ? Ts : ma.mk[m], c, m, Kc, ps);
                    quotient(pa, ps, col0 + i);
                }
            };
            if (L.z_cam0 >= 0 && c != L.rc) block(0, &x[L.full_cam0 + 6LL * L.cam_slot(c)], L.z_cam0 + 6LL * L.cam_slot(c));
            if (L.z_mk0 >= 0 && m != L.rm) block(1, &x[L.full_mk0 + 6LL * L.mk_slot(m)], L.z_mk0 + 6LL * L.mk_slot(m));
            if (L.z_fr0 >= 0) block(2, &x[L.full_fr0 + 6LL * f], L.z_fr0 + 6LL * f);
            if (L.z_intr0 >= 0)   // the intrinsics phase of obtain_transformation_derivs, :835-893: fx, cx, fy, cy +- J_delta in the
                for (int i = 0; i < 9; i++) {   // matrix, d0..d4 +- J_delta in a vector project_marker never reads (:630-640)
                    double Ka[9], Ks[9];
                    std::memcpy(Ka, Kc, sizeof Ka); std::memcpy(Ks, Kc, sizeof Ks);
                    const int at[4] = {0, 2, 4, 5};
                    if (i < 4) { Ka[at[i]] += delta; Ks[at[i]] -= delta; }
                    double pa[8], ps[8];
                    project_marker(p, L, ma.cam[c], ma.fr[f], ma.mk[m], c, m, Ka, pa);
                    project_marker(p, L, ma.cam[c], ma.fr[f], ma.mk[m], c, m, Ks, ps);
                    quotient(pa, ps, L.z_intr0 + 9LL * c + i);
                }
            cnt[o] = n;
        }
    }
    // compact the slabs
    int64_t w = 0;
    for (int64_t o = 0; o < N; o++) {
        int64_t base = o * slab;
        if (w != base) {
            std::memmove(rows + w, rows + base, sizeof(int32_t) * cnt[o]);
            std::memmove(cols + w, cols + base, sizeof(int32_t) * cnt[o]);
            std::memmove(vals + w, vals + base, sizeof(double) * cnt[o]);
        }
        w += cnt[o];
    }
    return w;
}

void residual_impl(const orc_problem *p, const double *x_full, const double *z, int res_mode, double *r) {
    Layout L(p);
    std::vector<double> x;
    build_full(p, L, x_full, z, x);
    Mats ma;
    build_mats(L, x, ma);
    // eval_curr_solution is serial in the reference (:996-1028); rows are independent, so the
    // result does not depend on the loop schedule.
    for (int64_t o = 0; o < p->num_obs; o++) {
        int c = p->obs_cam[o], m = p->obs_marker[o], f = p->obs_frame[o];
        double proj[8], Kc[9];
        cam_matrix(p, L, z, c, Kc);
        project_marker(p, L, ma.cam[c], ma.fr[f], ma.mk[m], c, m, Kc, proj);
        obs_residual(p, proj, p->obs_uv + 8 * o, res_mode, r + 8 * o);
    }
}

// ---------------- sparse pieces of SparseLevMarq::step ----------------
struct Csc {  // column-compressed, row indices ascending inside a column
    int n_rows = 0, n_cols = 0;
    std::vector<int64_t> ptr;
    std::vector<int32_t> idx;
    std::vector<double> val;
};

// Eigen setFromTriplets semantics (duplicates summed); used for J (libs/multicam_mapper.cpp:800)
Csc csc_from_triplets(int n_rows, int n_cols, const int32_t *rows, const int32_t *cols, const double *vals, int64_t nnz) {
    Csc A;
    A.n_rows = n_rows; A.n_cols = n_cols;
    std::vector<int64_t> count(n_cols + 1, 0);
    for (int64_t k = 0; k < nnz; k++) count[cols[k] + 1]++;
    for (int j = 0; j < n_cols; j++) count[j + 1] += count[j];
    std::vector<int64_t> pos(count.begin(), count.end() - 1);
    std::vector<int32_t> ri(nnz);
    std::vector<double> rv(nnz);
    for (int64_t k = 0; k < nnz; k++) {
        int64_t q = pos[cols[k]]++;
        ri[q] = rows[k]; rv[q] = vals[k];
    }
    A.ptr.assign(1, 0);
    for (int j = 0; j < n_cols; j++) {
        std::vector<std::pair<int32_t, double>> col;
        for (int64_t q = count[j]; q < count[j + 1]; q++) col.push_back({ri[q], rv[q]});
        std::stable_sort(col.begin(), col.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) { return a.first < b.first; });
        for (size_t q = 0; q < col.size(); q++) {
            if (!A.idx.empty() && (int64_t)A.idx.size() > A.ptr.back() && A.idx.back() == col[q].first)
                A.val.back() += col[q].second;
            else { A.idx.push_back(col[q].first); A.val.push_back(col[q].second); }
        }
        A.ptr.push_back((int64_t)A.idx.size());
    }
    return A;
}

Csc transpose(const Csc &A) {
    Csc T;
    T.n_rows = A.n_cols; T.n_cols = A.n_rows;
    T.ptr.assign(A.n_rows + 1, 0);
    for (size_t k = 0; k < A.idx.size(); k++) T.ptr[A.idx[k] + 1]++;
    for (int i = 0; i < A.n_rows; i++) T.ptr[i + 1] += T.ptr[i];
    T.idx.resize(A.idx.size()); T.val.resize(A.idx.size());
    std::vector<int64_t> pos(T.ptr.begin(), T.ptr.end() - 1);
    for (int j = 0; j < A.n_cols; j++)
        for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; k++) {
            int64_t q = pos[A.idx[k]]++;
            T.idx[q] = j; T.val[q] = A.val[k];
        }
    return T;
}

// SparseLevMarq<T>::mult (libs/sparselevmarq.h:265-325): res = Jt * J, one ordered map per output
// column, columns distributed over OpenMP threads; accumulation order inside a column is ascending
// residual row, as in the reference.
Csc jtj_mult(const Csc &Jt, const Csc &J) {
    const int n = J.n_cols;
    std::vector<std::map<uint32_t, double>> colmaps(n);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; j++) {
        std::map<uint32_t, double> &acc = colmaps[j];
        for (int64_t a = J.ptr[j]; a < J.ptr[j + 1]; a++) {
            double y = J.val[a];
            int32_t k = J.idx[a];
            for (int64_t b = Jt.ptr[k]; b < Jt.ptr[k + 1]; b++) {
                uint32_t i = (uint32_t)Jt.idx[b];
                double xv = Jt.val[b];
                auto it = acc.find(i);
                if (it == acc.end()) acc.insert({i, xv * y});
                else it->second += xv * y;
            }
        }
    }
    Csc R;
    R.n_rows = R.n_cols = n;
    R.ptr.assign(1, 0);
    for (int j = 0; j < n; j++) {
        for (auto &kv : colmaps[j]) { R.idx.push_back((int32_t)kv.first); R.val.push_back(kv.second); }
        R.ptr.push_back((int64_t)R.idx.size());
    }
    return R;
}

// Sparse LDL^T (up-looking, elimination-tree based -- the algorithm of T. Davis' LDL package, which
// Eigen::SimplicialLDLT also implements; libs/sparselevmarq.h:394-400).  Works on the upper triangle
// of the symmetrically permuted matrix.  The reference orders with AMD; any fill-reducing order gives
// the same delta up to rounding, so a fixed "frames first" elimination order is used here.
struct Ldlt {
    int n = 0;
    std::vector<int64_t> Lp;
    std::vector<int32_t> Li, parent, lnz, perm, iperm;
    std::vector<double> Lx, D;
    bool ok = false;

    void factor(const Csc &A /* full symmetric */, const std::vector<int32_t> &order) {
        n = A.n_cols;
        perm = order;
        iperm.assign(n, 0);
        for (int k = 0; k < n; k++) iperm[perm[k]] = k;
        // upper triangle of P A P^T, column-compressed
        std::vector<int64_t> Up(n + 1, 0);
        for (int j = 0; j < n; j++)
            for (int64_t a = A.ptr[j]; a < A.ptr[j + 1]; a++) {
                int qi = iperm[A.idx[a]], qj = iperm[j];
                if (qi <= qj) Up[qj + 1]++;
            }
        for (int j = 0; j < n; j++) Up[j + 1] += Up[j];
        std::vector<int32_t> Ui(Up[n]);
        std::vector<double> Ux(Up[n]);
        std::vector<int64_t> pos(Up.begin(), Up.end() - 1);
        for (int j = 0; j < n; j++)
            for (int64_t a = A.ptr[j]; a < A.ptr[j + 1]; a++) {
                int qi = iperm[A.idx[a]], qj = iperm[j];
                if (qi <= qj) { int64_t q = pos[qj]++; Ui[q] = qi; Ux[q] = A.val[a]; }
            }
        // symbolic
        parent.assign(n, -1); lnz.assign(n, 0);
        std::vector<int32_t> flag(n);
        for (int k = 0; k < n; k++) {
            flag[k] = k;
            for (int64_t q = Up[k]; q < Up[k + 1]; q++) {
                int i = Ui[q];
                if (i < k)
                    for (; flag[i] != k; i = parent[i]) {
                        if (parent[i] == -1) parent[i] = k;
                        lnz[i]++; flag[i] = k;
                    }
            }
        }
        Lp.assign(n + 1, 0);
        for (int k = 0; k < n; k++) Lp[k + 1] = Lp[k] + lnz[k];
        Li.assign(Lp[n], 0); Lx.assign(Lp[n], 0.0); D.assign(n, 0.0);
        // numeric
        std::vector<double> Y(n, 0.0);
        std::vector<int32_t> pattern(n);
        std::fill(lnz.begin(), lnz.end(), 0);
        ok = true;
        for (int k = 0; k < n; k++) {
            int top = n;
            flag[k] = k;
            for (int64_t q = Up[k]; q < Up[k + 1]; q++) {
                int i = Ui[q];
                if (i <= k) {
                    Y[i] += Ux[q];
                    int len = 0;
                    for (; flag[i] != k; i = parent[i]) { pattern[len++] = i; flag[i] = k; }
                    while (len > 0) pattern[--top] = pattern[--len];
                }
            }
            D[k] = Y[k]; Y[k] = 0.0;
            for (; top < n; top++) {
                int i = pattern[top];
                double yi = Y[i];
                Y[i] = 0.0;
                int64_t q2 = Lp[i] + lnz[i];
                for (int64_t q = Lp[i]; q < q2; q++) Y[Li[q]] -= Lx[q] * yi;
                double lki = yi / D[i];
                D[k] -= lki * yi;
                Li[q2] = k; Lx[q2] = lki; lnz[i]++;
            }
            if (D[k] == 0.0) { ok = false; return; }
        }
    }

    void solve(const double *b, double *xout) const {
        std::vector<double> y(n);
        for (int k = 0; k < n; k++) y[k] = b[perm[k]];
        for (int j = 0; j < n; j++)
            for (int64_t q = Lp[j]; q < Lp[j] + lnz[j]; q++) y[Li[q]] -= Lx[q] * y[j];
        for (int j = 0; j < n; j++) y[j] /= D[j];
        for (int j = n - 1; j >= 0; j--)
            for (int64_t q = Lp[j]; q < Lp[j] + lnz[j]; q++) y[j] -= Lx[q] * y[Li[q]];
        for (int k = 0; k < n; k++) xout[perm[k]] = y[k];
    }
};

std::vector<int32_t> frames_first_order(const orc_problem *p) {
    Layout L(p);
    std::vector<int32_t> order;
    order.reserve(L.z_len);
    if (L.z_fr0 >= 0)
        for (int64_t i = 0; i < 6LL * L.F; i++) order.push_back((int32_t)(L.z_fr0 + i));
    if (L.z_cam0 >= 0)
        for (int64_t i = 0; i < 6LL * (L.C - 1); i++) order.push_back((int32_t)(L.z_cam0 + i));
    if (L.z_mk0 >= 0)
        for (int64_t i = 0; i < 6LL * (L.M - 1); i++) order.push_back((int32_t)(L.z_mk0 + i));
    if (L.z_intr0 >= 0)
        for (int64_t i = 0; i < 9LL * L.C; i++) order.push_back((int32_t)(L.z_intr0 + i));
    return order;
}

// add mu to every diagonal entry, inserting missing ones
// (add_missing_diagonal_elements + get_diagonal_elements_refs_and_add, libs/sparselevmarq.h:252-258,328-336)
Csc add_diagonal(const Csc &A, double mu) {
    Csc R;
    R.n_rows = A.n_rows; R.n_cols = A.n_cols;
    R.ptr.assign(1, 0);
    for (int j = 0; j < A.n_cols; j++) {
        bool done = false;
        for (int64_t a = A.ptr[j]; a < A.ptr[j + 1]; a++) {
            if (!done && A.idx[a] > j) { R.idx.push_back(j); R.val.push_back(mu); done = true; }
            R.idx.push_back(A.idx[a]);
            R.val.push_back(A.val[a] + ((A.idx[a] == j) ? mu : 0.0));
            if (A.idx[a] == j) done = true;
        }
        if (!done) { R.idx.push_back(j); R.val.push_back(mu); }
        R.ptr.push_back((int64_t)R.idx.size());
    }
    return R;
}

struct StepSystem {
    Csc JtJ;
    std::vector<double> B;
};

void build_system(const orc_problem *p, const double *x_full, const double *z, int jac_mode, const double *r, StepSystem &S) {
    Layout L(p);
    const int64_t N = p->num_obs;
    std::vector<int32_t> rows(L.jac_slab * N), cols(L.jac_slab * N);
    std::vector<double> vals(L.jac_slab * N);
    int64_t nnz = jacobian_impl(p, x_full, z, jac_mode, rows.data(), cols.data(), vals.data());
    Csc J = csc_from_triplets((int)(8 * N), (int)L.z_len, rows.data(), cols.data(), vals.data(), nnz);
    Csc Jt = transpose(J);                    // libs/sparselevmarq.h:355
    S.JtJ = jtj_mult(Jt, J);                  // :362
    S.B.assign(L.z_len, 0.0);                 // B = -Jt*x64, :367
    for (int j = 0; j < J.n_cols; j++) {
        double s = 0;
        for (int64_t a = J.ptr[j]; a < J.ptr[j + 1]; a++) s += J.val[a] * r[J.idx[a]];
        S.B[j] = -s;
    }
}

}  // namespace

extern "C" {

int64_t orc_full_len(const orc_problem *p) { return Layout(p).full_len; }
int64_t orc_num_vars(const orc_problem *p) { return Layout(p).z_len; }

void orc_extract_z(const orc_problem *p, const double *x_full, double *z) {
    Layout L(p);
    if (L.z_cam0 >= 0) std::memcpy(z + L.z_cam0, x_full + L.full_cam0, sizeof(double) * 6 * (L.C - 1));
    if (L.z_mk0 >= 0) std::memcpy(z + L.z_mk0, x_full + L.full_mk0, sizeof(double) * 6 * (L.M - 1));
    if (L.z_fr0 >= 0) std::memcpy(z + L.z_fr0, x_full + L.full_fr0, sizeof(double) * 6 * L.F);
    if (L.z_intr0 >= 0)   // fill_io_vec_cam_intrinsics, :488-498
        for (int c = 0; c < L.C; c++) {
            double *q = z + L.z_intr0 + 9LL * c;
            const double *K = p->K + 9 * c;
            q[0] = K[0]; q[1] = K[2]; q[2] = K[4]; q[3] = K[5];
            for (int j = 0; j < 5; j++) q[4 + j] = p->dist ? p->dist[5 * c + j] : 0.0;
        }
}

int64_t orc_jac_capacity(const orc_problem *p) { return Layout(p).jac_slab * p->num_obs; }

void orc_get_intrinsics(const orc_problem *p, const double *z, double *K_out, double *dist_out) {   // intrinsics_vec2mats, :580-593
    Layout L(p);
    for (int c = 0; c < L.C; c++) {
        cam_matrix(p, L, z, c, K_out + 9 * c);
        for (int j = 0; j < 5; j++)
            dist_out[5 * c + j] = (z && L.z_intr0 >= 0) ? z[L.z_intr0 + 9LL * c + 4 + j] : (p->dist ? p->dist[5 * c + j] : 0.0);
    }
}

void orc_merge_z(const orc_problem *p, const double *x_full, const double *z, double *x_out) {
    Layout L(p);
    std::vector<double> x;
    build_full(p, L, x_full, z, x);
    std::memcpy(x_out, x.data(), sizeof(double) * L.full_len);
}

void orc_rodrigues_vec2mat(const double w[3], double R[9]) { rodrigues_v2m(w, R); }
void orc_rodrigues_mat2vec(const double R[9], double w[3]) { rodrigues_m2v(R, w); }

void orc_residuals(const orc_problem *p, const double *x_full, const double *z, int res_mode, double *r) {
    residual_impl(p, x_full, z, res_mode, r);
}

int64_t orc_jacobian(const orc_problem *p, const double *x_full, const double *z, int jac_mode, int32_t *rows,
                     int32_t *cols, double *vals) {
    return jacobian_impl(p, x_full, z, jac_mode, rows, cols, vals);
}

void orc_normal_equations_dense(const orc_problem *p, const double *x_full, const double *z, int jac_mode,
                                int res_mode, double *JtJ, double *B) {
    Layout L(p);
    std::vector<double> r(8 * p->num_obs);
    residual_impl(p, x_full, z, res_mode, r.data());
    StepSystem S;
    build_system(p, x_full, z, jac_mode, r.data(), S);
    const int64_t P = L.z_len;
    std::fill(JtJ, JtJ + P * P, 0.0);
    for (int j = 0; j < P; j++)
        for (int64_t a = S.JtJ.ptr[j]; a < S.JtJ.ptr[j + 1]; a++) JtJ[(int64_t)S.JtJ.idx[a] * P + j] = S.JtJ.val[a];
    std::memcpy(B, S.B.data(), sizeof(double) * P);
}

int orc_damped_solve(const orc_problem *p, const double *x_full, const double *z, int jac_mode, int res_mode,
                     double mu, double *delta) {
    std::vector<double> r(8 * p->num_obs);
    residual_impl(p, x_full, z, res_mode, r.data());
    StepSystem S;
    build_system(p, x_full, z, jac_mode, r.data(), S);
    Ldlt chol;
    chol.factor(add_diagonal(S.JtJ, mu), frames_first_order(p));
    if (!chol.ok) return -1;
    chol.solve(S.B.data(), delta);
    return 0;
}

// SparseLevMarq<T>::solve / init / step, libs/sparselevmarq.h:238-249,349-430,440-472
double orc_lm_solve(const orc_problem *p_in, const double *x_full, double *z_inout, const orc_lm_params *prm,
                    int jac_mode, int res_mode, orc_lm_iter *trace, int32_t trace_cap, int32_t *n_iters,
                    int32_t num_threads) {
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
    // with_huber: MultiCamMapper::solve sets hubberDelta = 10 before solver.solve (libs/multicam_mapper.cpp:425) and
    // optCallBack lowers it by 7.5/500 after every step while it is above 2.5 (:412-417)
    orc_problem pw = *p_in;
    if (pw.with_huber && !prm->huber_fixed) pw.huber_delta = 10;
    const orc_problem *p = &pw;
    Layout L(p);
    const int64_t P = L.z_len, rowsN = 8 * p->num_obs;
    std::vector<double> curr_z(z_inout, z_inout + P), x64(rowsN);
    auto sumsq = [&](const std::vector<double> &v) { double s = 0; for (double e : v) s += e * e; return s; };
    // init (:238-249)
    residual_impl(p, x_full, curr_z.data(), res_mode, x64.data());
    double currErr = sumsq(x64), prevErr = currErr;
    double mu = -1, v = 2;  // v is uninitialised in the reference (:133); 2 is the value every accepted step assigns (:411)
    std::vector<int32_t> order = frames_first_order(p);
    int mustExit = 0, iters = 0;
    for (int i = 0; i < prm->max_iters && !mustExit; i++) {
        // ---- step (:349-430) ----
        StepSystem S;
        build_system(p, x_full, curr_z.data(), jac_mode, x64.data(), S);
        if (mu < 0) {  // :369-377
            double maxv = -DBL_MAX;
            for (int j = 0; j < (int)P; j++)
                for (int64_t a = S.JtJ.ptr[j]; a < S.JtJ.ptr[j + 1]; a++)
                    if (S.JtJ.idx[a] == j && S.JtJ.val[a] > maxv) maxv = S.JtJ.val[a];
            mu = maxv * prm->tau;
        }
        double gain = 0, dnorm = 0;
        int ntries = 0;
        bool accepted = false;
        do {
            Ldlt chol;
            chol.factor(add_diagonal(S.JtJ, mu), order);  // :387-394
            std::vector<double> delta(P), est(P);
            chol.solve(S.B.data(), delta.data());         // :400
            for (int64_t k = 0; k < P; k++) est[k] = curr_z[k] + delta[k];
            residual_impl(p, x_full, est.data(), res_mode, x64.data());
            double err = sumsq(x64);
            double Lq = 0, d2 = 0;  // L = 0.5*delta^T(mu*delta - B), :406
            for (int64_t k = 0; k < P; k++) { Lq += delta[k] * (mu * delta[k] - S.B[k]); d2 += delta[k] * delta[k]; }
            Lq *= 0.5;
            dnorm = std::sqrt(d2);
            gain = (err - prevErr) / Lq;
            if (gain > 0 && ((err - prevErr) < 0)) {  // :409-415
                mu = mu * std::max(0.33, 1. - std::pow(2 * gain - 1, 3));
                v = 2.f;
                currErr = err;
                curr_z = est;
                accepted = true;
            } else {
                mu = mu * v;
                v = v * 5;
            }
        } while (gain <= 0 && ntries++ < 5 && !accepted);
        // ---- stop rules (:458-464) ----
        if (currErr < prm->min_error) mustExit = 1;
        if (std::fabs(prevErr - currErr) <= prm->min_step_error_diff ||
            std::fabs((prevErr - currErr) / (double)rowsN) <= prm->min_average_step_error_diff || !accepted)
            mustExit = 2;
        if (currErr > prevErr) mustExit = 3;
        if (trace && iters < trace_cap) {
            trace[iters].err = currErr; trace[iters].mu = mu; trace[iters].gain = gain;
            trace[iters].delta_norm = dnorm; trace[iters].accepted = accepted ? 1 : 0;
            trace[iters].tries = ntries + (accepted ? 1 : 0);
        }
        iters++;
        if (pw.with_huber && !prm->huber_fixed && pw.huber_delta > 2.5) pw.huber_delta = (float)((double)pw.huber_delta - 7.5 / 500);  // optCallBack
        prevErr = currErr;
    }
    std::memcpy(z_inout, curr_z.data(), sizeof(double) * P);
    if (n_iters) *n_iters = iters;
    return currErr;
}

void orc_reproj_stats(const orc_problem *p, const double *x_full, const double *z, double *rmse, double *mean_dist,
                      double *sum_sq) {
    orc_problem q = *p;
    q.with_huber = 0;
    std::vector<double> r(8 * p->num_obs);
    residual_impl(&q, x_full, z, ORC_RES_F64, r.data());
    double s = 0, d = 0;
    for (int64_t k = 0; k < 4 * p->num_obs; k++) {
        double e = r[2 * k] * r[2 * k] + r[2 * k + 1] * r[2 * k + 1];
        s += e;
        d += std::sqrt(e);
    }
    if (rmse) *rmse = std::sqrt(s / (4.0 * p->num_obs));
    if (mean_dist) *mean_dist = d / (4.0 * p->num_obs);
    if (sum_sq) *sum_sq = s;
}

}  // extern "C"

// ---- cv::undistortPoints restatement (see ba_oracle.h) ----
extern "C" void orc_undistort_points(const double K[9], const double *dist, int n_dist, int64_t n, const float *in, float *out) {
    double k[12];
    for (int i = 0; i < 12; i++) k[i] = (i < n_dist) ? dist[i] : 0.0;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1.0 / fx, ify = 1.0 / fy;
    for (int64_t i = 0; i < n; i++) {
        double x = ((double)in[2 * i] - cx) * ifx, y = ((double)in[2 * i + 1] - cy) * ify;
        const double x0 = x, y0 = y;
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = (1.0 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double dx = 2.0 * k[2] * x * y + k[3] * (r2 + 2.0 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dy = k[2] * (r2 + 2.0 * y * y) + 2.0 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        const double xx = K[0] * x + K[1] * y + K[2], yy = K[3] * x + K[4] * y + K[5], ww = 1.0 / (K[6] * x + K[7] * y + K[8]);
        out[2 * i] = (float)(xx * ww);
        out[2 * i + 1] = (float)(yy * ww);
    }
}

extern "C" void orc_distort_points(const double K[9], const double *dist, int n_dist, int64_t n, const double *in, double *out) {
    double k[12];
    for (int i = 0; i < 12; i++) k[i] = (i < n_dist) ? dist[i] : 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double x = (in[2 * i] - K[2]) / K[0], y = (in[2 * i + 1] - K[5]) / K[4];
        const double r2 = x * x + y * y;
        const double cdist = (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1.0 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2);
        const double xd = x * cdist + 2.0 * k[2] * x * y + k[3] * (r2 + 2.0 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double yd = y * cdist + k[2] * (r2 + 2.0 * y * y) + 2.0 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        out[2 * i] = K[0] * xd + K[2];
        out[2 * i + 1] = K[4] * yd + K[5];
    }
}
