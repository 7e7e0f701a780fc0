// Diagnostic (not part of the product): cycle stamps inside the dense LDL^T kernels on a random SPD system.
#define AAR_STAMPS 1
#include "../../automatic-ar_amd/csrc/solve_kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace aar;
int main() {
    const int n = 288, n_pad = 288, nT = 3;
    std::vector<double> A((size_t)n * n), M((size_t)n * n);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto &v : M) v = nd(g);
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k < n; k++) s += M[(size_t)i*n+k]*M[(size_t)j*n+k]; A[(size_t)i*n+j] = s + (i==j ? n : 0); }
    DeviceProblem P; P.n = n; P.n_pad = n_pad; P.nT = nT; P.A = 48;
    auto al = [](size_t bytes) { void *p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
    P.blk[0].S = (double*)al(sizeof(double)*n*n); P.blk[0].rhs = (double*)al(8*n); P.blk[0].g0 = (double*)al(8*n);
    P.Dfac = (double*)al(8*nT*96*96); P.Linv16 = (double*)al(8*nT*6*256); P.delta_s = (double*)al(8*n);
    P.Lp = (double*)al(8*(size_t)nT*n_pad*96); P.zf = (double*)al(8*n_pad);
    P.ent_fixed = (int32_t*)al(4*48); P.flags = (int32_t*)al(16); P.bs_flags = (int32_t*)al(64);
    std::vector<double> b(n, 1.0);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipMemcpy(P.blk[0].S, A.data(), 8*(size_t)n*n, hipMemcpyHostToDevice);
        (void)hipMemset(P.blk[0].rhs, 0, 8*n); (void)hipMemcpy(P.blk[0].g0, b.data(), 8*n, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        launch_chol(P, 0, 0.5, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[64];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
        printf("rep %d: launch_chol %.1f us\n", rep, ms*1e3);
        printf("  diag (last step): load %llu | 16 block steps %llu | store %llu | inverses %llu  (cycles)\n",
               st[1]-st[0], st[8]-st[1], st[9]-st[8], st[10]-st[9]);
#ifdef AAR_TIMELINE
        if (rep == 2) {
            unsigned long long tl[240];
            (void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_tl), sizeof tl);
            const char *nm[3] = {"wave 0", "wave 14 (rows)", "wave 5"};
            for (int w = 0; w < 3; w++) {
                printf("  %s: per step [C work | wait X | rows | wait Y]\n   ", nm[w]);
                for (int k = 0; k < 16; k++) { const unsigned long long *t = tl + (w * 16 + k) * 5; printf(" %d:[%llu|%llu|%llu|%llu]", k, t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3]); }
                printf("\n");
            }
        }
#endif
        printf("  trsm (last launch): load %llu | six blocks %llu\n", st[17]-st[16], st[24]-st[17]);
    }
    // check the solve: x = (A + 0.5 I)^-1 b
    std::vector<double> x(n); (void)hipMemcpy(x.data(), P.delta_s, 8*n, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) { double a = (j <= i) ? A[(size_t)i*n+j] : A[(size_t)j*n+i]; s += (a + (i==j?0.5:0))*x[j]; } worst = fmax(worst, fabs(s - 1.0)); }
    printf("residual of the solve: %.3e\n", worst);
    return 0;
}
