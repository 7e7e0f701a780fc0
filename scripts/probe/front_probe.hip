// Diagnostic (not part of the product): launch_chol on a random SPD system of n = argv[1] (default 288) unknowns; prints the time of
// the dense solve and the residual of the solution.  AAR_LDL_FRONT=0 selects the per-tile diag / trsm / update chain.
#define AAR_FRONT_TL 1
#include "../../automatic-ar_amd/csrc/solve_kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace aar;
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 288, nT = (n + 95) / 96, n_pad = nT * 96, nfix = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<double> A((size_t)n_pad * n_pad, 0.0), M((size_t)n * n);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto &v : M) v = nd(g);
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k < n; k++) s += M[(size_t)i*n+k]*M[(size_t)j*n+k]; A[(size_t)i*n_pad+j] = s + (i==j ? n : 0); }
    DeviceProblem P; P.n = n; P.n_pad = n_pad; P.nT = nT; P.A = n / 6;
    auto al = [](size_t bytes) { void *p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
    P.blk[0].S = (double*)al(sizeof(double)*n_pad*n_pad); P.blk[0].rhs = (double*)al(8*n_pad); P.blk[0].g0 = (double*)al(8*n_pad);
    P.Dfac = (double*)al(8*nT*96*96); P.Linv16 = (double*)al(8*nT*6*256); P.delta_s = (double*)al(8*n_pad);
    std::vector<int32_t> fx(n_pad / 6 + 1, 0);
    for (int i = 0; i < nfix; i++) fx[(i * 7) % (n / 6)] = 1;      // a few gauge entities
    P.ent_fixed = (int32_t*)al(4*fx.size()); (void)hipMemcpy(P.ent_fixed, fx.data(), 4*fx.size(), hipMemcpyHostToDevice);
    P.flags = (int32_t*)al(16); P.bs_flags = (int32_t*)al(64);
    std::vector<double> b(n_pad, 0.0), r(n_pad, 0.0);
    for (int i = 0; i < n; i++) { b[i] = nd(g); r[i] = nd(g); }
    for (int rep = 0; rep < 4; rep++) {
        (void)hipMemcpy(P.blk[0].S, A.data(), 8*(size_t)n_pad*n_pad, hipMemcpyHostToDevice);
        (void)hipMemcpy(P.blk[0].rhs, r.data(), 8*n_pad, hipMemcpyHostToDevice); (void)hipMemcpy(P.blk[0].g0, b.data(), 8*n_pad, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        if (getenv("AAR_TL_FIRST")) { P.nT = nT; launch_front<2, false>(P, P.blk[0], 0, 0.5, 0); } else launch_chol(P, 0, 0.5, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("n %d rep %d: launch_chol %.1f us (%s)\n", n, rep, ms*1e3, hipGetErrorString(hipGetLastError()));
        if (rep == 3) {   // timeline of the LAST front launch of the chain (the last tile) unless AAR_TL_FIRST stops the chain early
            unsigned long long tl[296];
            (void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_ftl), sizeof tl);
            printf("  kernel: load %llu | loop %llu | epilogue %llu cycles\n", tl[289]-tl[288], tl[290]-tl[289], tl[291]-tl[290]);
            const char *nm[3] = {"row wave 0", "first matrix wave", "matrix wave 15"};
            for (int w = 0; w < 3; w++) {
                printf("  %s per step [phase1 work | wait B1 | phase2 work (strip part) | wait B2]\n   ", nm[w]);
                for (int k = 0; k < 15; k++) { const unsigned long long *t = tl + (w * 16 + k) * 6; printf(" %d:[%llu|%llu|%llu(%llu)|%llu]", k, t[1]-t[0], t[2]-t[1], t[3]-t[2], w ? t[5]-t[2] : 0ull, t[4]-t[3]); }
                printf("\n");
            }
        }
    }
    int32_t fl[4]; (void)hipMemcpy(fl, P.flags, 16, hipMemcpyDeviceToHost);
    // check the solve: (A + 0.5 I) x = b + r on free rows, x = 0 on gauge rows
    std::vector<double> x(n_pad); (void)hipMemcpy(x.data(), P.delta_s, 8*n_pad, hipMemcpyDeviceToHost);
    double worst = 0, xg = 0, xmax = 0;
    for (int i = 0; i < n; i++) {
        xmax = fmax(xmax, fabs(x[i]));
        if (fx[i / 6]) { xg = fmax(xg, fabs(x[i])); continue; }
        double s = 0;
        for (int j = 0; j < n; j++) { if (fx[j / 6]) continue; double a = (j <= i) ? A[(size_t)i*n_pad+j] : A[(size_t)j*n_pad+i]; s += (a + (i==j?0.5:0))*x[j]; }
        worst = fmax(worst, fabs(s - (b[i] + r[i])));
    }
    for (int i = n; i < n_pad; i++) xg = fmax(xg, fabs(x[i]));
    printf("n %d: residual of the solve %.3e, max |x| on gauge / padding rows %.3e, max |x| %.3e, flags %d\n", n, worst, xg, xmax, fl[0] | fl[1] | fl[2] | fl[3]);
    return worst < 1e-9 && xg == 0.0 ? 0 : 1;
}
