// Solver "spcg" (aar_solver_options.solver = AAR_SOLVER_SPCG, and what AAR_SOLVER_AUTO picks for reduced systems that fit):
// the damped reduced system  (S + mu I) x = b  of one LM try (libs/sparselevmarq.h:384-400 after block elimination of the frames)
// solved by block-Jacobi-preconditioned conjugate gradients on the EXPLICIT Schur complement S that k_schur / k_schur_mfma have
// already left in HBM -- instead of the dense LDL^T chain of solve_kernels.hip (48 serial 6x6 pivot steps per 96-wide tile, 70 % of a
// step at 8 cameras / 40 markers).  Inexact LM: stopped at |r| <= eta |b| (eta = 0.1), the LM gain test judges the step as it
// judges an exact one; a solve that does not get there within the iteration cap raises device flag 8 and the host redoes that try
// with the direct chain (S is never modified here).
//
// Mapping.  ONE WAVEFRONT PER SHARED ENTITY (camera / marker), each its own workgroup on its own CU: wavefront e keeps the six rows
// 6e .. 6e+5 of S + mu I in REGISTERS (lane = (row i = lane / 8, column group g = lane % 8): columns {16k + 2g, 16k + 2g + 1},
// n_pad / 8 doubles per lane), the inverse of its own 6x6 diagonal block (the preconditioner), and its six entries of every CG vector.
// There is no workgroup barrier and no shared state inside a CU: a wavefront talks to the others only through the hand-over below.
//
// Pipelined CG (Ghysels & Vanroose 2014, Alg. 3), so that an iteration has ONE hand-over: every wavefront publishes
// m = M^-1 w (its six entries) together with its shares of (r,u), (w,u), (r,r) -- all of which exist BEFORE the matrix-vector
// product --, gathers everybody's, and then computes its six rows of n = A m, the scalars alpha / beta (every wavefront adds the same
// shares in the same order: same bits, same decisions) and the eight vector recurrences, all in registers.
//
// Hand-over without flags or fences: every value travels as an 8-byte agent-scope atomic store into a slot that holds a
// sentinel (a NaN bit pattern no arithmetic produces); the receiver's poll IS its load of the payload -- it re-loads the slots that
// still read as the sentinel.  One buffer per iteration (no slot is ever reused inside a launch), two buffer sets alternating between
// launches: a launch clears, at its start, the slots it owns in the OTHER set (the kernel boundary orders that against the next
// launch).  Against hop_publish / hop_wait (payload, s_waitcnt, flag | poll flag, load payload) this saves a memory round trip on
// each side.
#include "geom.hpp"
#include "kernels.h"

namespace aar {

namespace {

constexpr unsigned long long SPCG_EMPTY = 0xFFF85EEDFFF85EEDull;   // both halves equal: hipMemsetD32Async restores it

__device__ __forceinline__ bool sp_empty(double v) { return (unsigned long long)__double_as_longlong(v) == SPCG_EMPTY; }
__device__ __forceinline__ void sp_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double sp_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {   // lane permutation inside a row of 16 lanes (every lane has a source)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;

__device__ __forceinline__ double rl(double v, int l) {   // lane l's value, wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// sum over the 8 lanes of a row group (result in all 8)
__device__ __forceinline__ double sum8(double v) {
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    return v;
}
// sum over all 64 lanes, the same bits in every lane
__device__ __forceinline__ double sum64(double v) {
    v = sum8(v);
    v += dpp<DPP_MIRROR>(v);
    return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
// sum over the six matrix rows of a wavefront of a value that is the same in the 8 lanes of a row (rows 6, 7 hold zeros)
__device__ __forceinline__ double sum_rows(double v) {
    v += dpp<DPP_ROR8>(v);
    return (rl(v, 0) + rl(v, 16)) + rl(v, 32);
}

}  // namespace

#ifdef AAR_STAMPS   // diagnostic build (scripts/probe/spcg_probe.hip): cycle stamps of the reporting wavefront
__device__ unsigned long long g_sp_stamps[512];
__device__ int g_sp_polls[64];
#define SP_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (e == e0 && lane == 0 && (k) < 512) g_sp_stamps[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SP_STAMP(k) do { } while (0)
#endif

struct SpcgArgs {
    const double *S, *rhs, *g0;       // lower triangle of the Schur complement (no damping), Schur part of the rhs, shared gradient
    const int32_t *ent_fixed;
    int n, n_pad;
    double mu, eta2;
    int max_it;
    double *ws;                       // [2][SPCG_BUFS][stride] hand-over slots: m [n_pad] | shares [3][n_pad / 6]
    long long set_len;                // doubles per set
    int stride, parity;
    double *x_out;                    // [n_pad] delta_s
    int32_t *iters;                   // [0] iterations of this solve, [1] running total, [2] solves, [3] fallbacks requested (flag 8)
    int32_t *flags;
};

template <int NT>
__global__ void __launch_bounds__(64) k_spcg(const SpcgArgs a) {
    constexpr int NPAD = 96 * NT, NK = 6 * NT, NENT = 16 * NT, NM = (NPAD + 63) / 64, NE = (NENT + 63) / 64;
    __shared__ __align__(16) double mv[NPAD];
    __shared__ int32_t fx[NENT];
    const int lane = threadIdx.x, e = blockIdx.x, i = lane >> 3, g = lane & 7;
    for (int q = lane; q < NENT; q += 64) fx[q] = (6 * q >= a.n) ? 1 : a.ent_fixed[q];
    __syncthreads();
    if (fx[e]) {   // gauge / switched-off / padding entity: identity rows, zero right-hand side; nobody waits for this wavefront
        if (lane < 6) a.x_out[6 * e + lane] = 0.0;
        return;
    }
    int e0 = 0;    // the first free entity: its wavefront reports
    while (fx[e0]) e0++;
    double *set = a.ws + (size_t)a.parity * a.set_len, *other = a.ws + (size_t)(1 - a.parity) * a.set_len;
    SP_STAMP(0);
    {   // the slots this wavefront owns in the other set, for the launch after this one (the previous launch dirtied iters[0] + 2 buffers of it)
        const int nprev = min(a.iters[0] + 2, SPCG_BUFS);
        for (int bq = lane; bq < nprev; bq += 64) {
            double *b = other + (size_t)bq * a.stride;
#pragma unroll
            for (int k = 0; k < 6; k++) sp_st(b + 6 * e + k, __longlong_as_double((long long)SPCG_EMPTY));
#pragma unroll
            for (int k = 0; k < 3; k++) sp_st(b + NPAD + k * NENT + e, __longlong_as_double((long long)SPCG_EMPTY));
        }
    }
    // which of the slots this lane gathers belong to free entities (the others read as zero without being polled)
    unsigned mact = 0, eact = 0;
#pragma unroll
    for (int k = 0; k < NM; k++) { const int idx = lane + 64 * k; if (idx < NPAD && !fx[idx / 6]) mact |= 1u << k; }
#pragma unroll
    for (int k = 0; k < NE; k++) { const int en = lane + 64 * k; if (en < NENT && !fx[en]) eact |= 1u << k; }

    // ---- rows 6e .. 6e+5 of S + mu I into registers (S holds its lower triangle: the part right of the diagonal is read transposed) ----
    const bool ra = i < 6;
    const int row = 6 * e + (ra ? i : 0);
    double A2[2 * NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int c0 = 16 * k + 2 * g;
        double v0 = 0.0, v1 = 0.0;
        if (ra && !fx[c0 / 6]) {
            if (c0 + 1 <= row) {
                const double2 t = *reinterpret_cast<const double2 *>(a.S + (size_t)row * a.n_pad + c0);
                v0 = t.x; v1 = t.y;
            } else {
                v0 = (c0 <= row) ? a.S[(size_t)row * a.n_pad + c0] : a.S[(size_t)c0 * a.n_pad + row];
                v1 = a.S[(size_t)(c0 + 1) * a.n_pad + row];
            }
            if (c0 == row) v0 += a.mu;
            if (c0 + 1 == row) v1 += a.mu;
        }
        A2[2 * k] = v0; A2[2 * k + 1] = v1;
    }
    SP_STAMP(1);
    // ---- the preconditioner: inverse of the damped diagonal block, every lane the whole block (same instruction stream), keeps its row ----
    double mi[6];
    {
        double blk[6][6], inv[36];
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = 0; q <= p; q++) {
                const double v = a.S[(size_t)(6 * e + p) * a.n_pad + 6 * e + q] + (p == q ? a.mu : 0.0);
                blk[p][q] = v; blk[q][p] = v;
            }
        if (!spd6_inverse(blk, inv) && lane == 0) atomicOr(a.flags, 2);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < 6; p++) v = (i == p) ? inv[p * 6 + k] : v;
            mi[k] = v;
        }
    }
    auto prec = [&](double w) -> double {   // (M^-1 w)_i from the six entries of w in this wavefront
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) s = fma(mi[k], rl(w, 8 * k), s);
        return s;
    };
    bool dead = false;   // a hand-over timed out: flag 4, leave (wave-uniform)
    // publish: buffer b <- this wavefront's six entries of `val` and its three shares
    auto publish = [&](int b, double val, double s0, double s1, double s2) {
        double *buf = set + (size_t)b * a.stride;
        double *dst = nullptr;
        double v = 0.0;
        if (ra && g == 0) { dst = buf + 6 * e + i; v = val; }
        else if (lane == 1) { dst = buf + NPAD + e; v = s0; }
        else if (lane == 2) { dst = buf + NPAD + NENT + e; v = s1; }
        else if (lane == 3) { dst = buf + NPAD + 2 * NENT + e; v = s2; }
        if (dst) sp_st(dst, v);
    };
    // gather: buffer b -> the whole vector in mv (LDS), the three sums over all wavefronts
    auto gather = [&](int b, double &t0, double &t1, double &t2) {
        const double *buf = set + (size_t)b * a.stride;
        double v[NM], sh[3][NE];
#pragma unroll
        for (int k = 0; k < NM; k++) v[k] = ((mact >> k) & 1) ? sp_ld(buf + lane + 64 * k) : 0.0;
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int k = 0; k < NE; k++) sh[q][k] = ((eact >> k) & 1) ? sp_ld(buf + NPAD + q * NENT + lane + 64 * k) : 0.0;
        long spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NM; k++) ok = ok && !sp_empty(v[k]);
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int k = 0; k < NE; k++) ok = ok && !sp_empty(sh[q][k]);
            if (__ballot(!ok) == 0ull) break;
#ifdef AAR_STAMPS
            if (e == e0 && lane == 0 && b < 64) g_sp_polls[b]++;
#endif
            if (++spins > (1L << 20)) { dead = true; break; }   // ~1 s: a wavefront of the grid is not running (device shared / oversubscribed)
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int k = 0; k < NM; k++) if (sp_empty(v[k])) v[k] = sp_ld(buf + lane + 64 * k);
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int k = 0; k < NE; k++) if (sp_empty(sh[q][k])) sh[q][k] = sp_ld(buf + NPAD + q * NENT + lane + 64 * k);
        }
#pragma unroll
        for (int k = 0; k < NM; k++) if (lane + 64 * k < NPAD) mv[lane + 64 * k] = v[k];
        double p0 = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll
        for (int k = 0; k < NE; k++) { p0 += sh[0][k]; p1 += sh[1][k]; p2 += sh[2][k]; }
        t0 = sum64(p0); t1 = sum64(p1); t2 = sum64(p2);
        __syncthreads();   // (one wavefront: the LDS stores above are visible to its own reads below)
    };
    auto matvec = [&]() -> double {   // this lane's row of A against the vector in mv, summed over the row's 8 lanes
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const double2 t = reinterpret_cast<const double2 *>(mv)[8 * k + g];
            s0 = fma(A2[2 * k], t.x, s0);
            s1 = fma(A2[2 * k + 1], t.y, s1);
        }
        const double s = sum8(s0 + s1);
        __syncthreads();   // mv may be overwritten by the next gather
        return s;
    };

    // ---- x = 0, r = b, u = M^-1 r, w = A u ----
    double x = 0.0, r = ra ? a.rhs[row] + a.g0[row] : 0.0;
    double u = prec(r), w, z = 0.0, q = 0.0, s = 0.0, p = 0.0;
    double bb, d0, d1;
    SP_STAMP(2);
    publish(0, u, sum_rows(r * r), 0.0, 0.0);
    SP_STAMP(3);
    gather(0, bb, d0, d1);
    SP_STAMP(4);
    int it = 0, status = 0;   // status: 1 converged, 2 cap, 3 not positive definite
    if (!dead) {
        w = matvec();
        double g_old = 0.0, a_old = 0.0;
        if (!(bb > 0.0)) status = 1;   // b = 0: x = 0
        while (!status) {
            const double m = prec(w);
            const double ru = r * u, wu = w * u, rr = r * r;
            SP_STAMP(8 + 4 * it);
            publish(it + 1, m, sum_rows(ru), sum_rows(wu), sum_rows(rr));
            SP_STAMP(9 + 4 * it);
            double gam, dlt, rho;
            gather(it + 1, gam, dlt, rho);
            SP_STAMP(10 + 4 * it);
            if (dead) break;
            if (rho <= a.eta2 * bb) { status = 1; break; }
            if (it >= a.max_it) { status = 2; break; }
            const double nn = matvec();
            SP_STAMP(11 + 4 * it);
            const double beta = it ? gam / g_old : 0.0;
            const double den = it ? dlt - beta * gam / a_old : dlt;
            if (!(den > 0.0) || !(gam > 0.0)) { status = 3; break; }   // p^T A p <= 0: the damped system is not positive definite in floating point
            const double alpha = gam / den;
            z = fma(beta, z, nn);
            q = fma(beta, q, m);
            s = fma(beta, s, w);
            p = fma(beta, p, u);
            x = fma(alpha, p, x);
            r = fma(-alpha, s, r);
            u = fma(-alpha, q, u);
            w = fma(-alpha, z, w);
            g_old = gam; a_old = alpha;
            it++;
        }
    }
    SP_STAMP(5);
    if (ra && g == 0) a.x_out[row] = x;
    if (lane == 0) {
        if (dead) atomicOr(a.flags, 4);
        if (status == 3) atomicOr(a.flags, 2);
        if (status == 2) atomicOr(a.flags, 8);
        if (e == e0) {
            a.iters[0] = dead ? SPCG_BUFS : it;   // (a timed-out launch may have dirtied any buffer: the next one clears them all)
            a.iters[1] += it;
            a.iters[2] += 1;
            if (status == 2) a.iters[3] += 1;
        }
    }
}

template <int NT>
static void launch_spcg_nt(const SpcgArgs &a, int n_ent, hipStream_t st) {
    hipLaunchKernelGGL(k_spcg<NT>, dim3(n_ent), dim3(64), 0, st, a);
}

bool spcg_fits(int nT) { return nT >= 1 && nT <= SPCG_MAX_NT; }
size_t spcg_ws_doubles(int n_pad) { return (size_t)2 * SPCG_BUFS * spcg_stride(n_pad); }
void spcg_ws_reset(const DeviceProblem &P, hipStream_t st) {
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.spcg_ws), (int)0xFFF85EED, 2 * spcg_ws_doubles(P.n_pad), st);
}

// delta_s of (S + mu I) delta_s = rhs + g0 by CG on the explicit reduced system of block set `which` (S is left as it is)
void launch_spcg(const DeviceProblem &P, int which, double mu, hipStream_t st) {
    const DeviceProblem::Blocks &b = P.blk[which];
    SpcgArgs a;
    a.S = b.S; a.rhs = b.rhs; a.g0 = b.g0; a.ent_fixed = P.ent_fixed; a.n = P.n; a.n_pad = P.n_pad;
    a.mu = mu; a.eta2 = P.pcg_eta * P.pcg_eta; a.max_it = std::min(P.spcg_max_it, SPCG_MAX_IT);
    a.ws = P.spcg_ws; a.stride = spcg_stride(P.n_pad); a.set_len = (long long)SPCG_BUFS * a.stride; a.parity = P.spcg_parity;
    a.x_out = P.delta_s; a.iters = P.spcg_iters; a.flags = P.flags;
    P.spcg_parity ^= 1;
    const int n_ent = P.n_pad / 6;
    HookScope _h(P, KID_SPCG);
    switch (P.nT) {
#define SPCG_CASE(t) case t: launch_spcg_nt<t>(a, n_ent, st); break;
        SPCG_CASE(1) SPCG_CASE(2) SPCG_CASE(3) SPCG_CASE(4) SPCG_CASE(5) SPCG_CASE(6) SPCG_CASE(7) SPCG_CASE(8)
        SPCG_CASE(9) SPCG_CASE(10) SPCG_CASE(11) SPCG_CASE(12) SPCG_CASE(13) SPCG_CASE(14) SPCG_CASE(15) SPCG_CASE(16)
#undef SPCG_CASE
        default: break;   // (spcg_fits() is checked when the solver is chosen)
    }
}

}  // namespace aar
