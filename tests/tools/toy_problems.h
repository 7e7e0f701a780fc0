// Toy least-squares problems for the host-callback path of aar::SparseLevMarq (tests/tools/host_lm_main.cpp) and, compiled into oracle/_ref/libref_lm.so, for
// the reference's own solver (oracle/ref_harness.cpp: ref_lm_toy) -- TEST INFRASTRUCTURE.  Plain pointers so that both vector types can call them.
//   problem 0: chained Rosenbrock residuals, n unknowns (even start -1.2, odd start 1): r[2i] = 10 (z[i+1] - z[i]^2), r[2i+1] = 1 - z[i]
//   problem 1: one-view pose fit, 6 unknowns (Rodrigues vector, translation): pinhole projections of a 3 x 3 grid of points minus their projections at a fixed pose
#pragma once
#include <cmath>
#include <cstdint>

namespace toy {

inline int num_residuals(int problem, int n) { return problem == 0 ? 2 * (n - 1) : 18; }
inline int num_unknowns(int problem) { return problem == 0 ? 6 : 6; }
inline void start(int problem, double *z) {
    if (problem == 0) { for (int i = 0; i < 6; i++) z[i] = (i % 2) ? 1.0 : -1.2; }
    else { const double s[6] = {0.25, -0.15, 0.1, 0.08, -0.05, 2.3}; for (int i = 0; i < 6; i++) z[i] = s[i]; }
}

inline void rodrigues(const double *w, double R[9]) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (th < 1e-14) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    const double c = std::cos(th), s = std::sin(th), c1 = 1 - c, x = w[0] / th, y = w[1] / th, z = w[2] / th;
    R[0] = c + c1 * x * x; R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y; R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}
inline void project(const double *pose, double gx, double gy, double *u, double *v) {
    double R[9];
    rodrigues(pose, R);
    const double X = 0.3 * gx, Y = 0.3 * gy, Z = 0.05 * gx * gy;
    const double x = R[0] * X + R[1] * Y + R[2] * Z + pose[3], y = R[3] * X + R[4] * Y + R[5] * Z + pose[4], w = R[6] * X + R[7] * Y + R[8] * Z + pose[5];
    *u = 600.0 * x / w + 320.0;
    *v = 600.0 * y / w + 240.0;
}

inline void residuals(int problem, const double *z, int n, double *r) {
    if (problem == 0) {
        for (int i = 0; i + 1 < n; i++) { r[2 * i] = 10.0 * (z[i + 1] - z[i] * z[i]); r[2 * i + 1] = 1.0 - z[i]; }
    } else {
        const double truth[6] = {0.1, 0.2, -0.05, 0.02, 0.03, 2.0};
        int k = 0;
        for (int gy = -1; gy <= 1; gy++)
            for (int gx = -1; gx <= 1; gx++) {
                double u, v, u0, v0;
                project(z, gx, gy, &u, &v);
                project(truth, gx, gy, &u0, &v0);
                r[k++] = u - u0;
                r[k++] = v - v0;
            }
    }
}

// analytic Jacobian of problem 0 as (row, col, value) through a callback
template <class Emit>
inline void jacobian0(const double *z, int n, Emit emit) {
    for (int i = 0; i + 1 < n; i++) {
        emit(2 * i, i, -20.0 * z[i]);
        emit(2 * i, i + 1, 10.0);
        emit(2 * i + 1, i, -1.0);
    }
}

}  // namespace toy
