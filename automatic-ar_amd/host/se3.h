// Host-side SE(3) helpers: the small-matrix arithmetic the reference does with cv::Mat / cv::Rodrigues
// (libs/multicam_mapper.cpp:463-486).  No OpenCV, no Eigen.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>

namespace aar {

struct Mat3 {
    double m[9];
};

// rigid transform [R|t] (the 3x4 top of the reference's 4x4 CV_64F matrices)
struct Rigid {
    double R[9];
    double t[3];
    static Rigid identity() {
        Rigid r;
        for (int i = 0; i < 9; i++) r.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        r.t[0] = r.t[1] = r.t[2] = 0;
        return r;
    }
};

inline void mat3_mul(const double *a, const double *b, double *c) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
inline void mat3_vec(const double *a, const double *x, double *y) {
    for (int i = 0; i < 3; i++) y[i] = a[i * 3] * x[0] + a[i * 3 + 1] * x[1] + a[i * 3 + 2] * x[2];
}

inline Rigid compose(const Rigid &a, const Rigid &b) {  // a * b
    Rigid r;
    mat3_mul(a.R, b.R, r.R);
    double v[3];
    mat3_vec(a.R, b.t, v);
    for (int i = 0; i < 3; i++) r.t[i] = v[i] + a.t[i];
    return r;
}

inline Rigid inverse(const Rigid &a) {
    Rigid r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.R[i * 3 + j] = a.R[j * 3 + i];
    double v[3];
    mat3_vec(r.R, a.t, v);
    for (int i = 0; i < 3; i++) r.t[i] = -v[i];
    return r;
}

// cv::Rodrigues vector -> matrix: theta < DBL_EPSILON gives I, else c*I + (1-c)*n*n^T + s*[n]x
inline void rodrigues_vec2mat(const double w[3], double R[9]) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (th < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = std::cos(th), s = std::sin(th), c1 = 1. - c, ith = 1. / th;
    const double x = w[0] * ith, y = w[1] * ith, z = w[2] * ith;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

// One-sided Jacobi SVD of a 3x3: returns U*V^T, the orthogonal polar factor cv::Rodrigues substitutes
// for its input before extracting the rotation vector.
inline void nearest_rotation(const double Rin[9], double Q[9]) {
    double A[9], V[9];
    std::memcpy(A, Rin, sizeof A);
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; i++) {
                    alpha += A[i * 3 + p] * A[i * 3 + p];
                    beta += A[i * 3 + q] * A[i * 3 + q];
                    gamma += A[i * 3 + p] * A[i * 3 + q];
                }
                off = std::fmax(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
                if (std::fabs(gamma) < 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; i++) {
                    const double ap = A[i * 3 + p], aq = A[i * 3 + q];
                    A[i * 3 + p] = c * ap - s * aq;
                    A[i * 3 + q] = s * ap + c * aq;
                    const double vp = V[i * 3 + p], vq = V[i * 3 + q];
                    V[i * 3 + p] = c * vp - s * vq;
                    V[i * 3 + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    // A = U*diag(sigma): normalise the columns to get U, then Q = U*V^T
    double U[9];
    for (int j = 0; j < 3; j++) {
        double n = 0;
        for (int i = 0; i < 3; i++) n += A[i * 3 + j] * A[i * 3 + j];
        n = std::sqrt(n);
        for (int i = 0; i < 3; i++) U[i * 3 + j] = A[i * 3 + j] / n;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Q[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
}

// cv::Rodrigues matrix -> vector, including the theta ~ pi branch
inline void rodrigues_mat2vec(const double Rin[9], double w[3]) {
    double R[9];
    nearest_rotation(Rin, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    const double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            w[0] = w[1] = w[2] = 0;
            return;
        }
        double t = (R[0] + 1) * 0.5;
        rx = std::sqrt(std::fmax(t, 0.));
        t = (R[4] + 1) * 0.5;
        ry = std::sqrt(std::fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5;
        rz = std::sqrt(std::fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        const double k = theta / std::sqrt(rx * rx + ry * ry + rz * rz);
        w[0] = rx * k; w[1] = ry * k; w[2] = rz * k;
        return;
    }
    const double k = theta / (2 * s);
    w[0] = rx * k; w[1] = ry * k; w[2] = rz * k;
}

// vec2transformation_mat / transformation_mat2vec: libs/multicam_mapper.cpp:463-486
inline Rigid pose_to_rigid(const double v[6]) {
    Rigid r;
    rodrigues_vec2mat(v, r.R);
    r.t[0] = v[3]; r.t[1] = v[4]; r.t[2] = v[5];
    return r;
}
inline void rigid_to_pose(const Rigid &r, double v[6]) {
    rodrigues_mat2vec(r.R, v);
    v[3] = r.t[0]; v[4] = r.t[1]; v[5] = r.t[2];
}

}  // namespace aar
