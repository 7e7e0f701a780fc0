// Test driver (not part of the product): a caller of ucoslam::SparseLevMarq with evaluation functions of ITS OWN -- the toy problems of toy_problems.h -- compiled
// against the mirror of automatic-ar_amd/host/multicam_mapper.h, i.e. the host loop of host_levmarq.cpp (libs/sparselevmarq.h:80-118).
// Prints the per-step trace for tests/test_host_levmarq.py, which holds it against the reference's own solver (oracle/_ref: ref_lm_toy).
//   usage: host_lm_main <problem> <mode> <max_iters> <min_error> <min_step> <min_avg> <tau> <der_eps> <stop_after> <steps>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "../../automatic-ar_amd/host/multicam_mapper.h"
#include "toy_problems.h"

namespace ucoslam { using aar::SparseLevMarq; }
typedef ucoslam::SparseLevMarq<double> Solver;
typedef Solver::eVector eVector;

int main(int argc, char **argv) {
    if (argc < 11) return 2;
    const int problem = atoi(argv[1]), mode = atoi(argv[2]), max_iters = atoi(argv[3]);
    const double min_error = atof(argv[4]), min_step = atof(argv[5]), min_avg = atof(argv[6]), tau = atof(argv[7]), der_eps = atof(argv[8]);
    const int stop_after = atoi(argv[9]), steps = atoi(argv[10]);
    const int n = toy::num_unknowns(problem), m = toy::num_residuals(problem, n);
    Solver solver;
    Solver::Params prms(max_iters, min_error, min_step, min_avg, tau, der_eps);
    prms.min_average_step_error_diff = min_avg;
    prms.verbose = false;
    solver.setParams(prms);
    auto f = [&](const eVector &z, eVector &err) { err.resize(m); toy::residuals(problem, z.data(), n, err.data()); };
    auto fJ = [&](const eVector &z, aar::SparseJacobian<double> &J) {
        std::vector<aar::Triplet<double>> t;
        toy::jacobian0(z.data(), n, [&](int r, int c, double v) { t.push_back(aar::Triplet<double>(r, c, v)); });
        J.resize(m, n);
        J.setFromTriplets(t.begin(), t.end());
    };
    eVector z(n);
    toy::start(problem, z.data());
    const bool analytic = problem == 0 && mode != 1;
    int nt = 0;
    auto record = [&](int acc) { printf("step %d err %.17g mu %.17g acc %d\n", nt, solver.host_state().currErr, solver.host_state().mu, acc); nt++; };
    double err = 0;
    try {
        if (mode == 3) {
            solver.init(z, f);
            for (int k = 0; k < steps; k++) { const bool acc = analytic ? solver.step(f, fJ) : solver.step(f); record(acc ? 1 : 0); }
            err = solver.getCurrentSolution(z);
        } else {
            int calls = 0;
            solver.setStepCallBackFunc([&](const eVector &) { record(-1); });
            if (mode == 2) solver.setStopFunction([&](const eVector &) { return ++calls >= stop_after; });
            err = analytic ? solver.solve(z, f, fJ) : solver.solve(z, f);
        }
    } catch (const std::exception &e) { fprintf(stderr, "exception: %s\n", e.what()); return 1; }
    printf("final_err %.17g exit_code %d iterations %d\n", err, solver.host_state().exit_code, solver.host_state().iterations);
    for (int i = 0; i < n; i++) printf("z %d %.17g\n", i, z[i]);
    return 0;
}
