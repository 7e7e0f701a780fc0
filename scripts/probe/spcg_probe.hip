// Diagnostic (not part of the product): cycle stamps inside k_spcg on a random SPD system of the size of config 3's reduced system.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/probe/spcg_probe.hip -o scripts/probe/spcg_probe
#define AAR_STAMPS 1
#include "../../automatic-ar_amd/csrc/spcg_kernels.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <random>
using namespace aar;
namespace aar { volatile int g_last_kernel_id = 0; BacksubArgs backsub_args(const DeviceProblem &, int, int, int) { return BacksubArgs(); } }   // (the probe launches no riders; the real one lives in solve_kernels.hip)
int main(int argc, char **argv) {
    const int nT = argc > 1 ? atoi(argv[1]) : 3;
    const double eta = argc > 2 ? atof(argv[2]) : 1e-6;
    const int n = 96 * nT, n_pad = n, A = n / 6;
    std::vector<double> S((size_t)n * n), M((size_t)n * n);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto &v : M) v = nd(g);
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k < n / 4; k++) s += M[(size_t)i*n+k]*M[(size_t)j*n+k]; S[(size_t)i*n+j] = s + (i==j ? 1.0 : 0); }
    DeviceProblem P; P.n = n; P.n_pad = n_pad; P.nT = nT; P.A = A; P.pcg_eta = eta;
    auto al = [](size_t bytes) { void *p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
    P.blk[0].S = (double*)al(sizeof(double)*n*n); P.blk[0].rhs = (double*)al(8*n); P.blk[0].g0 = (double*)al(8*n);
    P.delta_s = (double*)al(8*n); P.ent_fixed = (int32_t*)al(4*A); P.flags = (int32_t*)al(16);
    P.spcg_ws = (double*)al(8 * spcg_ws_doubles(n_pad)); P.spcg_iters = (int32_t*)al(16);
    spcg_ws_reset(P, 0);
    {   // entity rows {R, t, J_l} for the coarse space: 1/6 of the entities cameras, the rest markers; small rotations, metre-scale translations
        P.C = A / 6; P.M = A - P.C; P.spcg_coarse = argc > 3 ? atoi(argv[3]) : 1; P.spcg_coarse_on = P.spcg_coarse;
        P.spcg_root_c = 0; P.spcg_root_m = 8 < A ? 8 : A - 1; P.spcg_n_free = A - 2;
        std::vector<double> er((size_t)A * 24, 0.0);
        for (int e = 0; e < A; e++) { double *r = &er[(size_t)e * 24]; r[0] = r[4] = r[8] = 1; r[12] = r[16] = r[20] = 1; r[13] = 0.05 * nd(g); r[15] = -r[13]; for (int k = 0; k < 3; k++) r[9 + k] = nd(g); }
        P.ent[0] = (double*)al(8 * er.size()); (void)hipMemcpy(P.ent[0], er.data(), 8 * er.size(), hipMemcpyHostToDevice);
        P.spcg_pre = (double*)al(8 * spcg_pre_doubles(n_pad));
    }
    std::vector<int32_t> fx(A, 0); fx[0] = 1; fx[8 < A ? 8 : A - 1] = 1;   // two gauge entities, as in a bundle problem
    (void)hipMemcpy(P.ent_fixed, fx.data(), 4 * A, hipMemcpyHostToDevice);
    std::vector<double> b(n, 1.0);
    (void)hipMemcpy(P.blk[0].S, S.data(), 8*(size_t)n*n, hipMemcpyHostToDevice);
    (void)hipMemcpy(P.blk[0].g0, b.data(), 8*n, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 4; rep++) {
        int zero[64] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sp_polls), zero, sizeof zero);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        launch_spcg(P, 0, 0.5, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[512]; int polls[64], it[4];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_sp_stamps), sizeof st);
        (void)hipMemcpyFromSymbol(polls, HIP_SYMBOL(g_sp_polls), sizeof polls);
        (void)hipMemcpy(it, P.spcg_iters, 16, hipMemcpyDeviceToHost);
        printf("rep %d: k_spcg_pre + k_spcg %.1f us (events), %d iterations; shader cycles: slab %llu | inverse + prec(r) %llu | publish0 %llu | gather0 %llu | pass0 %llu | total %llu\n", rep, ms*1e3, it[0],
               st[1]-st[0], st[2]-st[1], st[3]-st[2], st[4]-st[3], st[6]-st[4], st[5]-st[0]);
        if (rep == 3) for (int k = 0; k < it[0] && k < 24; k++)
            printf("   it %2d: prec+shares+publish %llu | poll %llu | gather %llu (re-polls %d) | matvec %llu | scalars+updates %llu\n", k, st[9+4*k]-st[8+4*k], st[300+k+1]-st[9+4*k], st[10+4*k]-st[9+4*k], polls[k+1],
                   st[11+4*k]-st[10+4*k], (k + 1 < it[0] ? st[8+4*(k+1)] : st[11+4*k]) - st[11+4*k]);
    }
    { unsigned long long st[512]; (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_sp_stamps), sizeof st); int fl[4]; (void)hipMemcpy(fl, P.flags, 16, hipMemcpyDeviceToHost);
      auto dv = [&](int k) { double d; memcpy(&d, &st[k], 8); return d; };
      printf("flags %d %d %d %d | bb %g d1 %g u %g r %g mv0 %g mv6 %g mv7 %g | gam %g dlt %g m %g w %g\n", fl[0], fl[1], fl[2], fl[3], dv(400), dv(401), dv(402), dv(403), dv(404), dv(405), dv(406), dv(410), dv(411), dv(412), dv(413)); }
    std::vector<double> x(n); (void)hipMemcpy(x.data(), P.delta_s, 8*n, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; i++) { if (fx[i / 6]) continue; double s = 0; for (int j = 0; j < n; j++) { if (fx[j / 6]) continue; double a = (j <= i) ? S[(size_t)i*n+j] : S[(size_t)j*n+i]; s += (a + (i==j?0.5:0))*x[j]; } worst = fmax(worst, fabs(s - 1.0)); }
    printf("residual of the solve (max norm, b = 1): %.3e\n", worst);
    return 0;
}
