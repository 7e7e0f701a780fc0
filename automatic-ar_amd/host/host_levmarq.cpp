// aar::detail::HostLevMarq -- the Levenberg-Marquardt loop of ucoslam::SparseLevMarq<double> (libs/sparselevmarq.h) for evaluation functions that live on the
// HOST: what aar::SparseLevMarq<double> runs when the callables it is given are NOT aar::MultiCamMapper's own (those run as HIP kernels, multicam_mapper.cpp).
// SURVEY.md section 8b keeps this class for API compatibility: the reference's solver is a general sparse LM that other callers use with a few hundred unknowns
// (a 2-view pose fit, a calibration refinement); here the normal equations are formed from the caller's sparse Jacobian and solved DENSE (LDL^T without pivoting,
// the step Eigen's SimplicialLDLT gives to rounding).  Semantics restated from the reference, rule for rule:
//   init        libs/sparselevmarq.h:238-249   curr_z, x = f(curr_z), currErr = prevErr = |x|^2, mu = -1
//   step        :349-430   J = f_J(curr_z); B = -J^T x with the x of the LAST evaluation (after a step whose tries all failed that is the failed trial's -- kept);
//               mu_0 = tau * max STORED diagonal entry of J^T J; <= 6 tries of {mu on every diagonal entry, solve, trial error, L = delta^T (mu delta - B) / 2,
//               gain = (err - prevErr) / L}; accept (gain > 0 and err < prevErr): mu *= max(0.33, 1 - (2 gain - 1)^3), v = 2; else mu *= v, v *= 5;
//               the loop goes on only while gain <= 0 (a positive gain with a larger error ends it unaccepted)
//   solve       :440-472   stop function: do { step; callback } while (!stop(curr_z)) -- prevErr then stays the INITIAL error (kept);
//               otherwise <= maxIters steps with the three exits of :458-461, prevErr <- currErr after each
//   derivatives :165-220   central differences, (f(z + e) - f(z - e)) / (2.f * e), entries with |d| <= 1e-4 dropped
// `v` is uninitialised in the reference (:133); v_0 = 2 here, as in the device path.  Nothing under oracle/ is included, linked or called.
#include <algorithm>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>

#include "multicam_mapper.h"

namespace aar {
namespace detail {

namespace {

// J^T J (dense, n x n, both triangles), B = -J^T x, and which diagonal entries the sparse product would STORE (a column of J with an entry)
void normal_equations(const SparseJacobian<double> &J, const std::vector<double> &x, int64_t n, std::vector<double> &H, std::vector<double> &B, std::vector<char> &stored) {
    // entries grouped by row, duplicates added up (Eigen's setFromTriplets does the same)
    std::vector<int64_t> order(J.val.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = (int64_t)k;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return J.row[a] != J.row[b] ? J.row[a] < J.row[b] : (J.col[a] != J.col[b] ? J.col[a] < J.col[b] : a < b); });
    H.assign((size_t)n * n, 0.0);
    B.assign((size_t)n, 0.0);
    stored.assign((size_t)n, 0);
    std::vector<int64_t> cols;
    std::vector<double> vals;
    size_t k = 0;
    while (k < order.size()) {
        const int64_t r = J.row[order[k]];
        cols.clear();
        vals.clear();
        for (; k < order.size() && J.row[order[k]] == r; k++) {
            const int64_t c = J.col[order[k]];
            if (c < 0 || c >= n || r < 0 || r >= (int64_t)x.size()) throw std::out_of_range("SparseLevMarq: Jacobian entry outside rows x cols");
            if (!cols.empty() && cols.back() == c) vals.back() += J.val[order[k]];
            else { cols.push_back(c); vals.push_back(J.val[order[k]]); }
        }
        for (size_t a = 0; a < cols.size(); a++) {
            stored[(size_t)cols[a]] = 1;
            B[(size_t)cols[a]] -= vals[a] * x[(size_t)r];
            for (size_t b = 0; b < cols.size(); b++) H[(size_t)cols[a] * n + cols[b]] += vals[a] * vals[b];
        }
    }
}

// delta = A^-1 b for symmetric A (n x n, row-major, destroyed): LDL^T without pivoting, as Eigen::SimplicialLDLT factors (libs/sparselevmarq.h:394-400)
void ldlt_solve(std::vector<double> &A, const std::vector<double> &b, int64_t n, std::vector<double> &x) {
    std::vector<double> d((size_t)n);
    for (int64_t j = 0; j < n; j++) {
        double dj = A[(size_t)j * n + j];
        for (int64_t k = 0; k < j; k++) dj -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * d[(size_t)k];
        d[(size_t)j] = dj;
        for (int64_t i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int64_t k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * d[(size_t)k];
            A[(size_t)i * n + j] = s / dj;   // unit lower factor below the diagonal
        }
    }
    x = b;
    for (int64_t i = 0; i < n; i++) {
        double s = x[(size_t)i];
        for (int64_t k = 0; k < i; k++) s -= A[(size_t)i * n + k] * x[(size_t)k];
        x[(size_t)i] = s;
    }
    for (int64_t i = 0; i < n; i++) x[(size_t)i] /= d[(size_t)i];
    for (int64_t i = n - 1; i >= 0; i--) {
        double s = x[(size_t)i];
        for (int64_t k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[(size_t)k];
        x[(size_t)i] = s;
    }
}

double sum_sq(const std::vector<double> &x) {
    double s = 0;
    for (double v : x) s += v * v;
    return s;
}

}  // namespace

void HostLevMarq::central_differences(const eVector &z, SparseJacobian<double> &J, const F &f, double der_epsilon) {   // libs/sparselevmarq.h:193-214
    eVector xp, xm;
    J.rows = 0;
    J.cols = (int64_t)z.size();
    J.row.clear(); J.col.clear(); J.val.clear();
    for (size_t i = 0; i < z.size(); i++) {
        eVector zp(z), zm(z);
        zp[i] += der_epsilon;
        zm[i] -= der_epsilon;
        f(zp, xp);
        f(zm, xm);
        if (xp.size() != xm.size()) throw std::runtime_error("SparseLevMarq: the evaluation function changed its output size");
        J.rows = (int64_t)xp.size();
        const double den = 2.f * der_epsilon;   // (sic: a float 2 times a double)
        for (size_t r = 0; r < xp.size(); r++) {
            const double d = (xp[r] - xm[r]) / den;
            if (std::fabs(d) > 1e-4) J.insert((int64_t)r, (int64_t)i) = d;
        }
    }
}

void HostLevMarq::init(const eVector &z, const F &f) {   // :238-249
    curr_z = z;
    f(curr_z, x);
    currErr = prevErr = sum_sq(x);
    mu = -1;
    active = true;
}

bool HostLevMarq::step(const F &f, const FJ &fJ, const Prm &prm) {   // :349-430
    const int64_t n = (int64_t)curr_z.size();
    J.resize((int64_t)x.size(), n);   // (empty: a function that only inserts starts from nothing, as one that calls setFromTriplets does)
    fJ(curr_z, J);
    std::vector<double> H, B;
    std::vector<char> stored;
    normal_equations(J, x, n, H, B, stored);
    if (mu < 0) {   // first time only
        double maxv = std::numeric_limits<double>::lowest();
        for (int64_t k = 0; k < n; k++)
            if (stored[(size_t)k] && H[(size_t)k * n + k] > maxv) maxv = H[(size_t)k * n + k];
        mu = maxv * prm.tau;
    }
    double gain = 0;
    int ntries = 0, tries = 0;
    bool accepted = false;
    std::vector<double> A, delta, est((size_t)n);
    do {
        tries++;
        A = H;
        for (int64_t k = 0; k < n; k++) A[(size_t)k * n + k] += mu;   // mu on every diagonal entry, the missing ones included (:387-392)
        ldlt_solve(A, B, n, delta);
        for (int64_t k = 0; k < n; k++) est[(size_t)k] = curr_z[(size_t)k] + delta[(size_t)k];
        f(est, x);
        const double err = sum_sq(x);
        double L = 0;
        for (int64_t k = 0; k < n; k++) L += delta[(size_t)k] * (mu * delta[(size_t)k] - B[(size_t)k]);
        L *= 0.5;
        gain = (err - prevErr) / L;
        if (gain > 0 && (err - prevErr) < 0) {
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
    last_gain = gain;
    last_tries = tries;
    if (prm.verbose)
        std::cout << std::setprecision(5) << "Curr Error=" << currErr << " AErr(prev-curr)=" << (prevErr - currErr) / x.size() << " gain=" << gain << " dumping factor=" << mu << std::endl;
    return accepted;
}

double HostLevMarq::solve(eVector &z, const F &f, const FJ &fJ, const Prm &prm, const std::function<void(const eVector &)> &step_cb,
                          const std::function<bool(const eVector &)> &stop_fn) {   // :440-472
    prevErr = std::numeric_limits<double>::max();
    init(z, f);
    iterations = 0;
    exit_code = 0;
    if (stop_fn) {
        do {
            step(f, fJ, prm);
            iterations++;
            if (step_cb) step_cb(curr_z);
        } while (!stop_fn(curr_z));
    } else {
        int mustExit = 0;
        for (int i = 0; i < prm.maxIters && !mustExit; i++) {
            if (prm.verbose) std::cerr << "iteration " << i << "/" << prm.maxIters << "  ";
            const bool accepted = step(f, fJ, prm);
            iterations++;
            if (currErr < prm.minError) mustExit = 1;
            if (std::fabs(prevErr - currErr) <= prm.min_step_error_diff || std::fabs((prevErr - currErr) / x.size()) <= prm.min_average_step_error_diff || !accepted) mustExit = 2;
            if (currErr > prevErr) mustExit = 3;
            if (step_cb) step_cb(curr_z);
            prevErr = currErr;
        }
        exit_code = mustExit;
    }
    z = curr_z;
    return currErr;
}

}  // namespace detail
}  // namespace aar
