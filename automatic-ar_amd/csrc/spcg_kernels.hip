// Solver "spcg" (aar_solver_options.solver = AAR_SOLVER_SPCG, and what AAR_SOLVER_AUTO picks for reduced systems that fit: up to 224 cameras + markers):
// the damped reduced system  (S + mu I) x = b  of one LM try (libs/sparselevmarq.h:384-400 after block elimination of the frames) solved by preconditioned
// conjugate gradients on the EXPLICIT Schur complement S that k_schur / k_schur_mfma have already left in HBM -- instead of the dense LDL^T chain of
// solve_kernels.hip (48 serial 6x6 pivot steps per 96-wide tile).  Inexact LM: an inner solve stops when BOTH r^T M^-1 r <= eta^2 b^T M^-1 b (ONE forcing term,
// default 3e-4: kernels.h SPCG_ETA_DEFAULT -- chosen for the final POSES, which then agree with the direct solver's to ~1e-6) AND r^T M^-1 r <= eps^2 mu (absolute
// tolerance, default 2e-5) hold; the LM gain test judges the step as it judges an exact one.  A solve that does not get there within the iteration cap (64 up to
// four tiles, 128 above) or meets non-positive curvature raises device flag 8 and the host redoes that try with the direct chain (S is never modified here); a solve
// whose predecessor came within 20 % of the cap goes there at once.
//
// Preconditioner.  Block-Jacobi (the 6x6 diagonal blocks) while that converges in a dozen iterations -- the early LM steps; once a solve of the run has needed
// spcg_coarse_from (12) iterations the COARSE SPACE of the groups' rigid motions joins (k_spcg_pre below: two-level additive preconditioner as block-Jacobi on an
// augmented system whose twelve extra unknowns sit in the two root entities' idle slots): 18 instead of 57 iterations at the last LM step of config 3.
//
// Mapping.  ONE WAVEFRONT PER SHARED ENTITY (camera / marker), each its own workgroup on its own CU: wavefront e keeps the six rows
// 6e .. 6e+5 of S + mu I in REGISTERS (lane = (row i = lane / 8, column group g = lane % 8): columns {16k + 2g, 16k + 2g + 1},
// n_pad / 8 doubles per lane), the inverse of its own 6x6 diagonal block (the preconditioner), and its six entries of every CG vector.
// There is no workgroup barrier and no shared state inside a CU: a wavefront talks to the others only through the hand-over below.
//
// Pipelined CG (Ghysels & Vanroose 2014, Alg. 3), so that an iteration has ONE hand-over: every wavefront publishes
// m = M^-1 w (its six entries) together with its shares of (r,u), (w,u) -- both of which exist BEFORE the matrix-vector
// product --, gathers everybody's, and then computes its six rows of n = A m, the scalars alpha / beta (every wavefront adds the same
// shares in the same order: same bits, same decisions) and the eight vector recurrences, all in registers.
//
// Hand-over without flags or fences: every value travels as an 8-byte agent-scope atomic store into a slot that holds a
// sentinel (a NaN bit pattern no arithmetic produces); the receiver's poll IS its load of the payload -- it re-loads the slots that
// still read as the sentinel.  One buffer per iteration (no slot is ever reused inside a launch), two buffer sets alternating between
// launches: a launch clears, at its start, the slots it owns in the OTHER set (the kernel boundary orders that against the next
// launch).  Against hop_publish / hop_wait (payload, s_waitcnt, flag | poll flag, load payload) this saves a memory round trip on
// each side.  Measured (scripts/probe/spcg_probe.hip, cycle stamps): an iteration is ~3 300 shader cycles -- publish 300, the poll's round trip 960 (260 per
// 16-byte load instruction: the number of records matters, not their placement), the sums over the gathered shares 720, the matrix-vector product 750, scalars
// and recurrences 560 -- and every piece of straight-line set-up code costs about one cycle per byte of instructions (the instruction cache is cold at every launch).
#include "geom.hpp"
#include "kernels.h"
#include "backsub.hpp"
#include "wave.hpp"

namespace aar {

namespace {

constexpr unsigned long long SPCG_EMPTY = 0xFFF85EEDFFF85EEDull;   // both halves equal: hipMemsetD32Async restores it

__device__ __forceinline__ bool sp_empty(double v) { return (unsigned long long)__double_as_longlong(v) == SPCG_EMPTY; }
__device__ __forceinline__ void sp_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the same store kept in the XCD's L2 (sc0 instead of sc1): for readers on the SAME XCD only, see SPCG_SPREAD
__device__ __forceinline__ void sp_st_xcd(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ double sp_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ double rl(double v, int l) {   // lane l's value, wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

}  // namespace

#ifdef AAR_STAMPS   // diagnostic build (scripts/probe/spcg_probe.hip): cycle stamps of the reporting wavefront
__device__ unsigned long long g_sp_stamps[512];
__device__ int g_sp_polls[80];
#define SP_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (e == e0 && lane == 0 && (k) < 512) g_sp_stamps[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SP_VAL(k, v) do { if (e == e0 && lane == 0 && (k) < 512) g_sp_stamps[k] = (unsigned long long)__double_as_longlong(v); } while (0)
#else
#define SP_STAMP(k) do { } while (0)
#define SP_VAL(k, v) do { } while (0)
#endif

struct SpcgArgs {
    const double *S, *rhs, *g0;       // lower triangle of the Schur complement (no damping), Schur part of the rhs, shared gradient
    const int32_t *ent_fixed;
    int n, n_pad;
    double mu, eta2;
    double abs2_mu;                   // eps^2 mu: the ABSOLUTE stopping threshold on r^T M^-1 r (along a weakly determined direction M ~ mu, so |A^-1 r|^2 ~ r^T M^-1 r / mu <= eps^2)
    int max_it;
    double *ws;                       // [2][SPCG_BUFS][stride] hand-over records, one of 8 doubles (64 bytes) per entity: m [6] | share of (r,u) | share of (w,u)
    long long set_len;                // doubles per set
    int stride, parity;
    double *x_out;                    // [n_pad] delta_s
    int32_t *iters;                   // [0] iterations of this solve, [1] running total, [2] solves, [3] solves that gave up (flag 8), [4] solves whose wavefronts all shared one XCD
    int32_t *flags;
    int spread;                       // 8: every eighth workgroup works (one XCD under round-robin placement); 1: every workgroup
    // riders: the workgroups of the grid that are NOT CG wavefronts do the frame back-substitution of the try (backsub.hpp) -- they fetch what does not depend
    // on delta_s (the frame's W blocks, g_f, V_f^-1) while the CG runs on the other XCD and wait for ONE flag, raised by the last CG wavefront to leave
    int ride, n_ent_total, n_riders;
    BacksubArgs bs;
    int32_t *done;                    // [0] arrivals (monotonic over launches), [1] flag = done_epoch once all n_ent_total CG workgroups of this launch have left
    int done_epoch;
    int test_drop;                    // test hook (AAR_SPCG_TEST_DROP): the wavefront of this entity leaves without a word, as if it had never been scheduled
    // CO kernels (two-level preconditioner through the AUGMENTED system, see k_spcg_pre): everything a wavefront keeps comes assembled from k_spcg_pre --
    // rows of the augmented matrix [n_pad][n_pad] (damping in, columns of fixed entities zero, the groups' roots turned into the coarse unknowns), right-hand side
    // [n_pad], inverse diagonal blocks [n_ent][36], Z_e [n_ent][36]
    const double *pre_rows, *pre_minv, *pre_z, *pre_azt, *pre_share;
    int root_c, root_m;               // the fixed entity whose slot carries the cameras' / markers' rigid-motion unknowns (-1: that group has none)
    int C, M;
};

typedef unsigned int sp_u32x4 __attribute__((ext_vector_type(4)));

template <int NT, bool CO>
__global__ void __launch_bounds__(64) k_spcg(const SpcgArgs a) {
    // NL: 16-byte pieces of a buffer per lane (a buffer = 16 NT records of 64 bytes = 64 NT pieces); piece c = lane + 64 k belongs to record
    // c / 4 and holds its words 2 (lane % 4), 2 (lane % 4) + 1: lanes with lane % 4 < 3 gather entries of m, lanes with lane % 4 == 3 the two shares
    constexpr int NPAD = 96 * NT, NK = 6 * NT, NENT = 16 * NT, NL = NT, NEB = (NENT + 63) / 64;
    __shared__ __align__(16) double mv[NPAD];
    // Placement.  Workgroups are dealt round-robin over the 8 XCDs (observed, not promised): with a.spread == 8 only every eighth workgroup works, so
    // that all wavefronts of the solve share ONE XCD and its L2.  Whether they really do is checked at run time (every wavefront publishes its
    // XCC id in the first, placement-independent hand-over): if so, the later hand-overs keep their records in that L2 (sc0 stores; the sc1 loads
    // are L2-served) instead of sending every store out to the fabric and every poll after it -- a hand-over then costs an L2 round trip, not a memory one.
    if (blockIdx.x % a.spread || (int)blockIdx.x >= a.n_ent_total * a.spread) {
        if (!a.ride || blockIdx.x % 8 == 0) return;   // (position 0 mod 8 behind the CG range: the CG's XCD -- left alone)
        __shared__ double red[32];
        const int bx = (int)blockIdx.x, cg_range = a.n_ent_total * a.spread;
        const int rid = bx < cg_range ? bx - bx / a.spread - 1 : a.n_ent_total * (a.spread - 1) + (bx - cg_range) - (bx - cg_range + 7) / 8;
        for (int blk = rid; blk <= a.bs.n_frame_blocks; blk += a.n_riders) backsub_body(a.bs, blk, red, a.done + 1, a.done_epoch, a.flags);
        return;
    }
    const int lane = threadIdx.x, e = blockIdx.x / a.spread, i = lane >> 3, g = lane & 7;
    auto leave = [&]() {   // this workgroup's entries of delta_s are on their way (agent-scope stores): drain them, arrive; the last one raises the riders' flag
        if (!a.ride) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int old = __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == a.done_epoch * a.n_ent_total) __hip_atomic_store(a.done + 1, a.done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const int n_free_ent = a.n / 6;   // entities beyond are padding: identity rows
    // (CO: a group's root carries the group's coarse unknowns -- its wavefront works, its record is read like any other)
    auto fixed = [&](int ent) -> bool { return ent >= n_free_ent || (a.ent_fixed[ent < n_free_ent ? ent : 0] != 0 && !(CO && (ent == a.root_c || ent == a.root_m))); };
    double *set = a.ws + (size_t)a.parity * a.set_len, *other = a.ws + (size_t)(1 - a.parity) * a.set_len;
    const int nprev_raw = a.iters[0];
    {   // the record this wavefront owns in the other set's buffers, for the launch after this one (the previous launch dirtied iters[0] + 2 of them): 8 buffers per store.
        // Wavefronts of fixed entities do it too: a root's record is written by the launches that carry the coarse space and must read as idle to the others
        const int nprev = min(nprev_raw + 2, SPCG_BUFS);
        for (int bq = lane >> 3; bq < nprev; bq += 8) sp_st(other + (size_t)bq * a.stride + 8 * e + (lane & 7), __longlong_as_double((long long)SPCG_EMPTY));
    }
    if (fixed(e)) {   // gauge / switched-off / padding entity: identity rows, zero right-hand side; nobody waits for this wavefront
        if (lane < 6) sp_st(a.x_out + 6 * e + lane, 0.0);
        leave();
        return;
    }
    if (e == a.test_drop) return;
    // ---- every load of the set-up is issued before the first is used: one memory latency, not one per stage ----
    int e0 = -1, n_act = 0;    // the first free entity (its wavefront reports), the number of free entities
#pragma unroll
    for (int k = 0; k < NEB; k++) {
        const unsigned long long fr = __ballot(lane + 64 * k < NENT && !fixed(lane + 64 * k));
        if (e0 < 0 && fr) e0 = 64 * k + __builtin_ctzll(fr);
        n_act += __builtin_popcountll(fr);
    }
    const double xcc = (double)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf);   // hwreg(HW_REG_XCC_ID, 0, 4)
    bool same_xcd = false;     // until the first hand-over has told
    unsigned pact = 0;   // bit k: piece lane + 64 k belongs to a free entity (the others read as zero without being polled)
#pragma unroll
    for (int k = 0; k < NL; k++) if (!fixed((lane + 64 * k) >> 2)) pact |= 1u << k;
    // rows 6e .. 6e+5 of S + mu I into registers (S holds its lower triangle: the part right of the diagonal is read transposed)
    const bool ra = i < 6;
    const int row = 6 * e + (ra ? i : 0);
    double A2[2 * NK];
    double blk[6][6];
    double r, mi[6], zo[6];
    const bool pseudo = CO && (e == a.root_c || e == a.root_m);   // this wavefront's six unknowns are its group's rigid motion
    const int my_root = e < a.C ? a.root_c : (e < a.C + a.M ? a.root_m : -1);   // where this entity's group keeps its coarse unknowns
    if (CO) {
        // regular rows: assembled by k_spcg_pre (damping in, columns of fixed entities zero, (A Z)(row, .) in the roots' columns); a root's rows -- the coarse
        // unknowns' -- are rows of (A Z)^T (zero in the roots' own columns: E is kept apart, below)
        const double *rp = pseudo ? a.pre_azt + (size_t)((e == a.root_c ? 0 : 6) + (ra ? i : 0)) * a.n_pad : a.pre_rows + (size_t)row * a.n_pad;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const double2 t = reinterpret_cast<const double2 *>(rp)[8 * k + g];
            A2[2 * k] = ra ? t.x : 0.0; A2[2 * k + 1] = ra ? t.y : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double2 t = reinterpret_cast<const double2 *>(a.pre_minv + 36 * e + 6 * (ra ? i : 0))[k];
            mi[2 * k] = (ra && !pseudo) ? t.x : 0.0; mi[2 * k + 1] = (ra && !pseudo) ? t.y : 0.0;
            const double2 z2 = reinterpret_cast<const double2 *>(a.pre_z + 36 * e + 6 * (ra ? i : 0))[k];
            const bool zon = ra && !pseudo && my_root >= 0;
            zo[2 * k] = zon ? z2.x : 0.0; zo[2 * k + 1] = zon ? z2.y : 0.0;
        }
        r = (ra && !pseudo) ? a.rhs[row] + a.g0[row] : 0.0;
    } else {
        unsigned long long cfx[(NK + 63) / 64] = {};   // bit k: column pair k belongs to a gauge / padding entity
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int c0 = 16 * k + 2 * g, c1 = c0 + 1;
            A2[2 * k] = a.S[(c0 <= row) ? (size_t)row * a.n_pad + c0 : (size_t)c0 * a.n_pad + row];
            A2[2 * k + 1] = a.S[(c1 <= row) ? (size_t)row * a.n_pad + c1 : (size_t)c1 * a.n_pad + row];
            if (fixed(c0 / 6)) cfx[k >> 6] |= 1ull << (k & 63);
        }
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = 0; q <= p; q++) blk[p][q] = a.S[(size_t)(6 * e + p) * a.n_pad + 6 * e + q];
        r = a.rhs[row] + a.g0[row];
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int c0 = 16 * k + 2 * g;
            const bool off = !ra || ((cfx[k >> 6] >> (k & 63)) & 1);
            A2[2 * k] = off ? 0.0 : A2[2 * k] + (c0 == row ? a.mu : 0.0);
            A2[2 * k + 1] = off ? 0.0 : A2[2 * k + 1] + (c0 + 1 == row ? a.mu : 0.0);
        }
        if (!ra) r = 0.0;
    }
    SP_STAMP(0);
    SP_STAMP(1);
    // ---- the preconditioner: inverse of the damped diagonal block, every lane the whole block (same instruction stream), keeps its row (CO: k_spcg_pre did it) ----
    if (!CO) {
        double inv[36];
#pragma unroll
        for (int p = 0; p < 6; p++) {
            blk[p][p] += a.mu;
#pragma unroll
            for (int q = 0; q < p; q++) blk[q][p] = blk[p][q];
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
    // CO, a root's wavefront: row i of E (the coarse unknowns against each other) and entry i of Z^T b are sums of the shares k_spcg_pre's workgroups left, taken in
    // entity order; the preconditioner's block is the inverse of E's diagonal block
    double ec[12];
#pragma unroll
    for (int q = 0; q < 12; q++) ec[q] = 0.0;
    if (CO && pseudo) {
        const int Gq = e == a.root_c ? 0 : 1, elo = Gq ? a.C : 0, ehi = Gq ? a.C + a.M : a.C;
        // lane (class = lane % 8, slots 10 (lane / 8) .. + 9): the entities q = class mod 8 of the group, 40 entities per round with every load in flight at once
        // (a fixed entity's slots are zero from the allocation on); then the eight classes are added: a fixed order
        const int ecl = lane & 7, sg = lane >> 3;
        double acc[10];
#pragma unroll
        for (int t = 0; t < 10; t++) acc[t] = 0.0;
        for (int q0 = elo; q0 < ehi; q0 += 40) {
            double2 v[5][5];
#pragma unroll
            for (int kk = 0; kk < 5; kk++) {
                const int q = q0 + 8 * kk + ecl;
#pragma unroll
                for (int t = 0; t < 5; t++) v[kk][t] = q < ehi ? reinterpret_cast<const double2 *>(a.pre_share + 80 * q + 10 * sg)[t] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int kk = 0; kk < 5; kk++)
#pragma unroll
                for (int t = 0; t < 5; t++) { acc[2 * t] += v[kk][t].x; acc[2 * t + 1] += v[kk][t].y; }
        }
#pragma unroll
        for (int t = 0; t < 10; t++) acc[t] = sum8(acc[t]);
        if (ecl == 0) {
#pragma unroll
            for (int t = 0; t < 5; t++) reinterpret_cast<double2 *>(mv + 10 * sg)[t] = make_double2(acc[2 * t], acc[2 * t + 1]);
        }
        __builtin_amdgcn_wave_barrier();
        double eb[6][6], inv[36];
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = 0; q <= p; q++) { const double v = mv[12 * p + 6 * Gq + q]; eb[p][q] = v; eb[q][p] = v; }
        if (!spd6_inverse(eb, inv) && lane == 0) atomicOr(a.flags, 2);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < 6; p++) v = (i == p) ? inv[p * 6 + k] : v;
            mi[k] = v;
        }
        // (the shares of a group's entities are that group's ROWS of E, all twelve columns: the coupling block too)
#pragma unroll
        for (int b = 0; b < 12; b++) ec[b] = ra ? mv[12 * i + b] : 0.0;
        r = ra ? mv[72 + i] : 0.0;
        __builtin_amdgcn_wave_barrier();
    }
    auto prec = [&](double w) -> double {   // (M^-1 w)_i from the six entries of w in this wavefront
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) s = fma(mi[k], rl(w, 8 * k), s);
        return s;
    };
    bool dead = false;   // a hand-over timed out: flag 4, leave (wave-uniform)
    // publish: buffer b <- this wavefront's record: its six entries of `val` (lanes 8 i -> words 0..5) and its shares of the two dot products
    // (lanes 1, 2 -> words 6, 7): ONE store instruction, 64 contiguous bytes.  s1, s2 are per-row products (the same in the 8 lanes of a row, zero in
    // rows 6 and 7): lane g picks its quantity, the rows are added by a rotation and two row exchanges -- the sum arrives in the lanes that store it
    auto publish = [&](int b, double val, double s1, double s2) {
        double *rec = set + (size_t)b * a.stride + 8 * e;
        double sh = (g & 1) ? s1 : s2;
        sh += dpp<DPP_ROR8>(sh);
        sh = sum_across_rows(sh);
        const bool is_m = ra && g == 0, is_s = lane == 1 || lane == 2;
        if (is_m || is_s) {
            if (same_xcd) sp_st_xcd(rec + (is_m ? i : 5 + lane), is_m ? val : sh);
            else sp_st(rec + (is_m ? i : 5 + lane), is_m ? val : sh);
        }
    };
    // gather: buffer b -> the whole vector in mv (LDS), the two sums over all wavefronts.  The poll IS the load of the payload (16-byte sc1
    // loads: a CU's memory queue is what a hand-over costs, so as few requests as possible; every 8-byte half is validated by itself)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(set, 0, (int)(a.set_len * 8), 0x00020000);
    auto gather = [&](int b, double &t1, double &t2) {
        const int soff = b * a.stride * 8;
        double2 cur[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) cur[k] = ((pact >> k) & 1) ? make_double2(__longlong_as_double((long long)SPCG_EMPTY), __longlong_as_double((long long)SPCG_EMPTY)) : make_double2(0.0, 0.0);
        long spins = 0;
        if (SPCG_PRE > 0) __builtin_amdgcn_s_sleep(SPCG_PRE);   // (the others publish when this wavefront does: a poll issued at once finds nothing and costs a round trip)
        for (;;) {
            sp_u32x4 t[NL];
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (sp_empty(cur[k].x) || sp_empty(cur[k].y)) t[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 16 * (lane + 64 * k), soff, 16 /* sc1 */);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                if (sp_empty(cur[k].x) || sp_empty(cur[k].y)) {
                    const double lo = __hiloint2double((int)t[k][1], (int)t[k][0]), hi = __hiloint2double((int)t[k][3], (int)t[k][2]);
                    if (sp_empty(cur[k].x)) cur[k].x = lo;
                    if (sp_empty(cur[k].y)) cur[k].y = hi;
                    ok = ok && !sp_empty(cur[k].x) && !sp_empty(cur[k].y);
                }
            }
            if (__ballot(!ok) == 0ull) break;
#ifdef AAR_STAMPS
            if (e == e0 && lane == 0 && b < 80) g_sp_polls[b]++;
#endif
            if (++spins > (1L << 19)) { dead = true; break; }   // ~1 s: a wavefront of the grid is not running (device shared / oversubscribed)
            __builtin_amdgcn_s_sleep(1);
        }
        SP_STAMP(300 + b);
        const int w2 = lane & 3;
        double sg = 0.0, sd = 0.0;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            if (w2 < 3) *reinterpret_cast<double2 *>(mv + 6 * ((lane + 64 * k) >> 2) + 2 * w2) = cur[k];
            else { sg += cur[k].x; sd += cur[k].y; }
        }
        // lanes = 3 mod 4 hold (sum of shares of (r,u), of (w,u)) of their records: the second moves one lane down, then both are added over the quads
        // (rotations by 4 and 8 keep the position in the quad) and over the rows
        const double sd_dn = dpp<0xF4>(sd);   // quad_perm [0,1,3,3]: lane 2 of a quad reads lane 3 (outside the selection below: a DPP source lane must be active)
        double sh = w2 == 3 ? sg : (w2 == 2 ? sd_dn : 0.0);
        sh += dpp<DPP_ROR4>(sh);
        sh += dpp<DPP_ROR8>(sh);
        sh = sum_across_rows(sh);
        t1 = rl(sh, 3); t2 = rl(sh, 2);
        __builtin_amdgcn_wave_barrier();   // (one wavefront, LDS operations complete in order: the reads below see the stores above)
    };
    auto matvec = [&]() -> double {   // this lane's row of A against the vector in mv, summed over the row's 8 lanes
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const double2 t = reinterpret_cast<const double2 *>(mv)[8 * k + g];
            s0 = fma(A2[2 * k], t.x, s0);
            s1 = fma(A2[2 * k + 1], t.y, s1);
        }
        if (CO && pseudo && g == 0) {   // the coarse unknowns against each other: E times the roots' entries of the vector
#pragma unroll
            for (int b = 0; b < 6; b++) {
                s0 = fma(ec[b], a.root_c >= 0 ? mv[6 * max(a.root_c, 0) + b] : 0.0, s0);
                s1 = fma(ec[6 + b], a.root_m >= 0 ? mv[6 * max(a.root_m, 0) + b] : 0.0, s1);
            }
        }
        const double s = sum8(s0 + s1);
        __builtin_amdgcn_wave_barrier();   // (mv is overwritten by the next gather: program order is enough)
        return s;
    };

    // ---- x = 0, r = b, u = M^-1 r, w = A u.  Stopping rule in the preconditioner's norm: r^T M^-1 r <= eta^2 b^T M^-1 b -- it is the (r,u) the
    //      recurrences need anyway (a third reduction for r^T r would be a ninth word in every record), and along real LM runs it stops
    //      earlier AND ends closer to the exact run than the Euclidean rule (profiles/r04_spcg_experiment.txt) ----
    double x = 0.0;
    double u = prec(r), w = 0.0, z = 0.0, q = 0.0, s = 0.0, p = 0.0;
    double bb, d1;
    SP_STAMP(2);
    publish(0, u, r * u, i == 0 ? xcc + 4096.0 * xcc * xcc : 0.0);   // (second share of this buffer: where this wavefront runs; sum of id and of id^2, exact)
    SP_STAMP(3);
    gather(0, bb, d1);
    SP_STAMP(4);
    {
        const double s2 = floor(d1 * (1.0 / 4096.0)), s1 = d1 - 4096.0 * s2;
        same_xcd = a.spread > 1 && (double)n_act * s2 == s1 * s1;          // all XCC ids equal <=> n sum(id^2) == (sum id)^2
    }
    SP_VAL(400, bb); SP_VAL(401, d1); SP_VAL(402, u); SP_VAL(403, r); SP_VAL(404, mv[0]); SP_VAL(405, mv[6]); SP_VAL(406, mv[7]);
    int it = 0, status = 0;   // status: 1 converged, 2 cap, 3 non-positive curvature
    // CO: every wavefront carries its group's coarse unknowns along (row i: entry i) -- the same recurrences on the root's entries of the gathered vectors with the
    // same alpha, beta -- so that x = x' + Z_e c needs no hand-over of its own at the end
    double uc = 0.0, qc = 0.0, pc = 0.0, xc = 0.0;
    const int rslot = (CO && my_root >= 0 && ra) ? 6 * my_root + i : 0;
    const bool track = CO && my_root >= 0 && ra && !pseudo;
    if (track) uc = mv[rslot];
    if (!dead) {
        w = matvec();
        SP_STAMP(6);
        double inv_g = 0.0, inv_a = 0.0;   // 1 / gamma and 1 / alpha of the previous iteration
        if (!(bb > 0.0)) status = 1;   // b = 0: x = 0
        while (!status) {
            const double m = prec(w);
            SP_STAMP(8 + 4 * it);
            publish(it + 1, m, r * u, w * u);
            SP_STAMP(9 + 4 * it);
            double gam, dlt;
            gather(it + 1, gam, dlt);
            SP_STAMP(10 + 4 * it);
            if (it == 0) { SP_VAL(410, gam); SP_VAL(411, dlt); SP_VAL(412, m); SP_VAL(413, w); }
            if (dead) break;
            if (gam <= a.eta2 * bb && gam <= a.abs2_mu) { status = 1; break; }
            if (it >= a.max_it) { status = 2; break; }
            const double mc = track ? mv[rslot] : 0.0;
            const double nn = matvec();
            SP_STAMP(11 + 4 * it);
            const double beta = gam * inv_g;                   // (0 in the first iteration)
            const double den = dlt - beta * gam * inv_a;       // p^T A p
            if (!(den > 0.0) || !(gam > 0.0)) {
                // p^T A p <= 0.  Either the recurrences have reached their floor (a forcing term below what pipelined CG can attain, |r| / |b| ~ 1e-8 ..
                // 1e-12: x is as good as it gets), or they have broken down on the way there (long runs on ill-conditioned systems), or the damped system
                // really is not positive definite in floating point.  CG does not try to tell the last two apart: it gives up (flag 8) and the direct
                // chain, whose pivots do tell, redoes the try
                status = (gam <= 1e-16 * bb) ? 1 : 3;
                break;
            }
            const double inv_den = rcp_refined(den);
            const double alpha = gam * inv_den;
            inv_a = den * (inv_g = rcp_refined(gam));
            z = fma(beta, z, nn);
            q = fma(beta, q, m);
            s = fma(beta, s, w);
            p = fma(beta, p, u);
            x = fma(alpha, p, x);
            r = fma(-alpha, s, r);
            u = fma(-alpha, q, u);
            w = fma(-alpha, z, w);
            if (CO) { qc = fma(beta, qc, mc); pc = fma(beta, pc, uc); xc = fma(alpha, pc, xc); uc = fma(-alpha, qc, uc); }
            it++;
        }
    }
    SP_STAMP(5);
    if (CO) {   // x = x' + Z_e c; the root's own entries of delta_s are zero
#pragma unroll
        for (int q = 0; q < 6; q++) x = fma(zo[q], rl(xc, 8 * q), x);
        if (pseudo) x = 0.0;
    }
    if (ra && g == 0) sp_st(a.x_out + row, x);
    leave();
    if (lane == 0) {
        if (dead) atomicOr(a.flags, 4);
        if (status == 2 || status == 3) atomicOr(a.flags, 8);
        if (e == e0) {
            a.iters[0] = dead ? SPCG_BUFS : it;   // (a timed-out launch may have dirtied any buffer: the next one clears them all)
            a.iters[1] += it;
            a.iters[2] += 1;
            if (status >= 2) a.iters[3] += 1;
            if (same_xcd) a.iters[4] += 1;
        }
    }
}

// ---- k_spcg_pre: the two-level preconditioner of k_spcg<.., true>, as an AUGMENTED system; one workgroup per shared entity, launched right before k_spcg ----
//
// The <= 8 eigenvalues of the block-Jacobi-preconditioned reduced system that fall below 0.1 at late LM steps (the rest of the spectrum sits in [0.1, 1.8]) belong
// to the near-gauge modes of T = T_c^-1 T_f T_m: ALL non-root cameras moved by one rigid motion G (T_c <- G T_c: every frame absorbs it, T_f <- G T_f, only the root
// camera's observations resist) and ALL non-root markers moved by one (T_m <- G T_m, T_f <- T_f G^-1).  In the Rodrigues parameters entity j's rows of those modes
// are Z_j = [[J_l(w_j)^-1, 0], [-[t_j]x, I]] (columns: rotation, translation of G), in the six columns of its group (scripts/experiments/spcg_coarse.py: they span the
// weak eigenvectors to 0.99 in the preconditioner's norm; with them the CG needs 12 instead of 21 iterations per LM step at the pose-grade forcing term, 18 instead
// of 57 at the last step).  The additive two-level preconditioner D^-1 + Z blockdiag(E_cc, E_mm)^-1 Z^T, E = Z^T A Z, is block-Jacobi on the augmented system
//      [ A      A Z ] [x']   [ b     ]
//      [ Z^T A  E   ] [c ] = [ Z^T b ],      x = x' + Z c        (positive semi-definite, consistent: CG does not mind)
// and the augmented system needs NO new machinery in k_spcg: the twelve coarse unknowns take the places of the two ROOT entities -- whose rows and columns of the
// reduced system are idle (identity) and whose hand-over records nobody reads.  k_spcg then runs exactly as before on 6 (C + M) unknowns, two more of its wavefronts
// working; the coarse coefficients travel in the hand-over that exists, at no cost per iteration.  This kernel assembles what the wavefronts keep: the rows of the
// augmented matrix with the damping in and the columns of fixed entities zero, the inverse diagonal blocks, the right-hand side and Z_j; the workgroup that finishes
// last adds up E and Z^T b and writes the roots' rows.
struct SpcgPreArgs {
    const double *S, *rhs, *g0;
    const double *ent;                // entity rows {R, t, J_l} of the pose S was built at
    const int32_t *ent_fixed;
    int n, n_pad, n_ent, C, M, root_c, root_m;
    double mu;
    double *rows, *minv, *z;          // what k_spcg's regular wavefronts read: their rows of the augmented matrix [n_pad][n_pad], inverse diagonal blocks [n_ent][36], Z_j [n_ent][36]
    double *azt;                      // [12][n_pad] (A Z)^T
    double *share;                    // [n_ent][80]: Z_j^T (A Z)_j (6 x 12) | Z_j^T b_j (6) | idle
    int32_t *flags;
};

__device__ __forceinline__ void coarse_block(const double *__restrict__ row, bool on, double *__restrict__ Z) {   // Z_j from {t, J_l} = row[9..20]
    const double *t = row + 9, *J = row + 12;
    const double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    const double det = J[0] * c00 + J[1] * c01 + J[2] * c02;
    const double id = on ? 1.0 / det : 0.0, o1 = on ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 36; q++) Z[q] = 0.0;
    Z[0] = c00 * id; Z[1] = (J[2] * J[7] - J[1] * J[8]) * id; Z[2] = (J[1] * J[5] - J[2] * J[4]) * id;
    Z[6] = c01 * id; Z[7] = (J[0] * J[8] - J[2] * J[6]) * id; Z[8] = (J[2] * J[3] - J[0] * J[5]) * id;
    Z[12] = c02 * id; Z[13] = (J[1] * J[6] - J[0] * J[7]) * id; Z[14] = (J[0] * J[4] - J[1] * J[3]) * id;
    Z[19] = o1 * t[2]; Z[20] = -o1 * t[1];
    Z[24] = -o1 * t[2]; Z[26] = o1 * t[0];
    Z[30] = o1 * t[1]; Z[31] = -o1 * t[0];
    Z[21] = o1; Z[28] = o1; Z[35] = o1;
}

__global__ void __launch_bounds__(256) k_spcg_pre(const SpcgPreArgs a) {
    extern __shared__ __align__(16) double sh[];
    const int tid = threadIdx.x, j = blockIdx.x;
    const int n_free_ent = a.n / 6, npose = a.C + a.M;
    if (j >= n_free_ent || a.ent_fixed[j]) return;
    double *zl = sh;                                  // [n_ent][36]
    double *al = zl + 36 * a.n_ent;                   // [6][n_pad]: rows 6j .. 6j+5 of A = S + mu I, the columns of fixed / padding entities zero
    double *az = al + 6 * a.n_pad;                    // [3][72] partial sums, then [72] (A Z)_j
    double *bj = az + 216;                            // [6] b_j
    int32_t *fx = reinterpret_cast<int32_t *>(bj + 8);   // [n_ent] 1: fixed / padding entity
    for (int q = tid; q < a.n_ent; q += 256) fx[q] = (q >= n_free_ent || a.ent_fixed[q < n_free_ent ? q : 0] != 0) ? 1 : 0;
    double er[12];   // {t, J_l} of entity tid (the loads travel with those of S below)
    {
        const bool pose = tid < npose && tid < a.n_ent;
#pragma unroll
        for (int q = 0; q < 12; q++) er[q] = pose ? a.ent[(size_t)tid * ENT_STRIDE + 9 + q] : 0.0;
    }
    // rows of S: every load in flight before the first is used (the transposed half is one line per element)
    {
        const int total = 6 * a.n_pad;
#pragma unroll 1
        for (int q0 = tid; q0 < total; q0 += 8 * 256) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + 256 * k;
                if (q < total) {
                    const int i = q / a.n_pad, col = q - i * a.n_pad, row = 6 * j + i;
                    v[k] = a.S[(col <= row) ? (size_t)row * a.n_pad + col : (size_t)col * a.n_pad + row];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + 256 * k;
                if (q < total) {
                    const int i = q / a.n_pad, col = q - i * a.n_pad, row = 6 * j + i;
                    al[q] = v[k] + (col == row ? a.mu : 0.0);
                }
            }
        }
    }
    for (int q = tid; q < a.n_ent; q += 256) {
        double Z[36], row[21];
        const bool pose = q < npose;
#pragma unroll
        for (int k = 0; k < 12; k++) row[9 + k] = q == tid ? er[k] : (pose ? a.ent[(size_t)q * ENT_STRIDE + 9 + k] : 0.0);
        coarse_block(row, pose && !(q >= n_free_ent || a.ent_fixed[q < n_free_ent ? q : 0] != 0), Z);
#pragma unroll
        for (int k = 0; k < 18; k++) reinterpret_cast<double2 *>(zl + 36 * q)[k] = make_double2(Z[2 * k], Z[2 * k + 1]);
    }
    if (tid < 6) bj[tid] = a.rhs[6 * j + tid] + a.g0[6 * j + tid];
    __syncthreads();
    for (int q = tid; q < 6 * a.n_pad; q += 256) { const int col = q % a.n_pad; if (fx[col / 6]) al[q] = 0.0; }   // the columns of fixed / padding entities
    __syncthreads();
    const bool pose = j < npose;
    if (tid >= 224) {   // half a wavefront's worth of idle lanes; one of them inverts the diagonal block while the others form (A Z)_j
        if (tid == 224) {
            double blk[6][6], inv[36];
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int q = 0; q < 6; q++) blk[p][q] = al[p * a.n_pad + 6 * j + q];
            if (!spd6_inverse(blk, inv)) atomicOr(a.flags, 2);
#pragma unroll
            for (int q = 0; q < 36; q++) a.minv[36 * j + q] = inv[q];
        }
    } else if (tid < 216) {   // (A Z)(i, c) over a third of the group's entities each
        const int o = tid % 72, part = tid / 72, i = o / 12, c = o % 12;
        const int elo = c < 6 ? 0 : a.C, ehi = c < 6 ? a.C : npose, per = (ehi - elo + 2) / 3;
        const int lo = 6 * min(elo + part * per, ehi), hi = 6 * min(elo + (part + 1) * per, ehi);
        double acc = 0.0;
#pragma unroll 6
        for (int col = lo; col < hi; col++) acc = fma(al[i * a.n_pad + col], zl[6 * col + (c % 6)], acc);
        az[72 * part + o] = acc;
    }
    __syncthreads();
    if (tid < 72) {
        const int i = tid / 12, c = tid % 12;
        const double v = (az[tid] + az[72 + tid]) + az[144 + tid];
        az[tid] = v;
        a.azt[(size_t)c * a.n_pad + 6 * j + i] = v;
        const int root = c < 6 ? a.root_c : a.root_m;       // the coarse unknowns' columns of this entity's rows: where the root's (zero) columns were
        if (root >= 0) al[i * a.n_pad + 6 * root + (c % 6)] = v;
    }
    __syncthreads();
    for (int q = tid; q < 3 * a.n_pad; q += 256) reinterpret_cast<double2 *>(a.rows + (size_t)6 * j * a.n_pad)[q] = reinterpret_cast<const double2 *>(al)[q];
    const double *Zj = zl + 36 * j;
    if (tid < 36) a.z[36 * j + tid] = Zj[tid];
    if (pose && tid >= 64 && tid < 64 + 78) {   // shares of E (rows: this entity's group) and of Z^T b
        const int o = tid - 64;
        double v = 0.0;
        if (o < 72) {
            const int c = o / 12, b = o % 12;
#pragma unroll
            for (int i = 0; i < 6; i++) v = fma(Zj[6 * i + c], az[12 * i + b], v);
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) v = fma(Zj[6 * i + (o - 72)], bj[i], v);
        }
        a.share[80 * j + o] = v;
    }
}

size_t spcg_pre_lds_bytes(int n_pad) { return sizeof(double) * ((size_t)36 * (n_pad / 6) + (size_t)6 * n_pad + 224) + sizeof(int32_t) * (n_pad / 6 + 2); }
// doubles of k_spcg_pre's workspace: rows of the augmented matrix | minv | Z | (A Z)^T | shares
size_t spcg_pre_doubles(int n_pad) { const size_t ne = n_pad / 6; return (size_t)n_pad * n_pad + 36 * ne + 36 * ne + (size_t)12 * n_pad + 80 * ne + 8; }

static void launch_spcg_pre(const DeviceProblem &P, int which, double mu, hipStream_t st, SpcgArgs &k) {
    const size_t ne = P.n_pad / 6;
    SpcgPreArgs a;
    a.S = P.blk[which].S; a.rhs = P.blk[which].rhs; a.g0 = P.blk[which].g0; a.ent = P.ent[which]; a.ent_fixed = P.ent_fixed;
    a.n = P.n; a.n_pad = P.n_pad; a.n_ent = (int)ne; a.C = P.C; a.M = P.M; a.mu = mu;
    a.root_c = P.spcg_root_c; a.root_m = P.spcg_root_m;
    double *w = P.spcg_pre;
    a.rows = w; w += (size_t)P.n_pad * P.n_pad;
    a.minv = w; w += 36 * ne;
    a.z = w; w += 36 * ne;
    a.azt = w; w += (size_t)12 * P.n_pad;
    a.share = w; w += 80 * ne;
    a.flags = P.flags;
    k.pre_rows = a.rows; k.pre_minv = a.minv; k.pre_z = a.z; k.pre_azt = a.azt; k.pre_share = a.share; k.root_c = a.root_c; k.root_m = a.root_m;
    static size_t granted = 0;
    const size_t lds = spcg_pre_lds_bytes(P.n_pad);
    allow_dynamic_lds(reinterpret_cast<const void *>(k_spcg_pre), lds, granted);
    HookScope _h(P, KID_SPCG_PRE);
    hipLaunchKernelGGL(k_spcg_pre, dim3((unsigned)ne), dim3(256), lds, st, a);
}

bool spcg_fits(int nT) { return nT >= 1 && nT <= SPCG_MAX_NT; }
// wavefronts (= workgroups) of k_spcg<nT> a CU can hold at once, as the runtime's occupancy calculator sees this kernel's registers and LDS (0: not known)
int spcg_resident_per_cu(int nT, bool coarse) {
    int nb = 0;
    hipError_t rc = hipErrorInvalidValue;
    switch (nT) {
#define SPCG_CASE(t) case t: { int nc = 1 << 30; if (coarse) rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, k_spcg<t, true>, 64, 0); \
                               if (!coarse || rc == hipSuccess) rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spcg<t, false>, 64, 0); nb = std::min(nb, nc); } break;
        SPCG_CASE(1) SPCG_CASE(2) SPCG_CASE(3) SPCG_CASE(4) SPCG_CASE(5) SPCG_CASE(6) SPCG_CASE(7) SPCG_CASE(8)
        SPCG_CASE(9) SPCG_CASE(10) SPCG_CASE(11) SPCG_CASE(12) SPCG_CASE(13) SPCG_CASE(14)
#undef SPCG_CASE
        default: break;
    }
    if (rc != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb;
}
size_t spcg_ws_doubles(int n_pad) { return (size_t)2 * SPCG_BUFS * spcg_stride(n_pad); }
void spcg_ws_reset(const DeviceProblem &P, hipStream_t st) {
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.spcg_ws), (int)0xFFF85EED, 2 * spcg_ws_doubles(P.n_pad), st);
    (void)hipMemsetAsync(P.spcg_done, 0, 2 * sizeof(int32_t), st);   // the riders' arrival counter and flag start over
    P.spcg_epoch = 0;
}

// delta_s of (S + mu I) delta_s = rhs + g0 by CG on the explicit reduced system of block set `which` (S is left as it is).
// trial >= 0: the frame back-substitution z[trial] = z[which] + delta rides in the same launch (true is returned if it did: the caller then skips launch_backsub)
bool launch_spcg(const DeviceProblem &P, int which, double mu, hipStream_t st, int trial) {
    const DeviceProblem::Blocks &b = P.blk[which];
    SpcgArgs a;
    a.S = b.S; a.rhs = b.rhs; a.g0 = b.g0; a.ent_fixed = P.ent_fixed; a.n = P.n; a.n_pad = P.n_pad;
    a.mu = mu; a.eta2 = P.pcg_eta_now * P.pcg_eta_now; a.abs2_mu = P.pcg_abs_tol * P.pcg_abs_tol * mu; a.max_it = std::min(P.spcg_max_it, SPCG_MAX_IT);
    a.ws = P.spcg_ws; a.stride = spcg_stride(P.n_pad); a.set_len = (long long)SPCG_BUFS * a.stride; a.parity = P.spcg_parity;
    a.x_out = P.delta_s; a.iters = P.spcg_iters; a.flags = P.flags; a.test_drop = P.spcg_test_drop;
    a.pre_rows = a.pre_minv = a.pre_z = a.pre_azt = a.pre_share = nullptr; a.root_c = a.root_m = -1; a.C = P.C; a.M = P.M;
    const bool coarse = spcg_coarse_now(P) && P.spcg_coarse_on;   // this solve carries the coarse space: k_spcg_pre assembles the augmented system first
    if (coarse) launch_spcg_pre(P, which, mu, st, a);
    a.spread = P.spcg_spread;
    P.spcg_parity ^= 1;
    const int n_ent = P.n_pad / 6;
    a.n_ent_total = n_ent;
    // riders: only where the CG leaves seven XCDs idle anyway, and not when somebody wants the back-substitution's own time
    a.ride = (trial >= 0 && P.tune.spcg_backsub_rides && a.spread >= 8 && !P.hook.pre && P.F > 0) ? 1 : 0;
    a.bs = backsub_args(P, which, trial >= 0 ? trial : which, 1);
    a.done = P.spcg_done; a.done_epoch = 0; a.n_riders = 0;
    int grid = n_ent * a.spread;
    if (a.ride) {
        a.done_epoch = ++P.spcg_epoch;
        const int base = n_ent * (a.spread - 1), want = std::min(a.bs.n_frame_blocks + 1, 4096);
        int extra = 0;
        if (base < want) extra = ((want - base) * 8 + 6) / 7;            // (every eighth block behind the CG range sits on the CG's XCD and is left alone)
        grid += extra;
        a.n_riders = base + extra - (extra + 7) / 8;
    }
    HookScope _h(P, KID_SPCG);
    switch (P.nT) {
#define SPCG_CASE(t) case t: if (coarse) hipLaunchKernelGGL((k_spcg<t, true>), dim3(grid), dim3(64), 0, st, a); \
                             else hipLaunchKernelGGL((k_spcg<t, false>), dim3(grid), dim3(64), 0, st, a); break;
        SPCG_CASE(1) SPCG_CASE(2) SPCG_CASE(3) SPCG_CASE(4) SPCG_CASE(5) SPCG_CASE(6) SPCG_CASE(7) SPCG_CASE(8)
        SPCG_CASE(9) SPCG_CASE(10) SPCG_CASE(11) SPCG_CASE(12) SPCG_CASE(13) SPCG_CASE(14)
#undef SPCG_CASE
        default: break;   // (spcg_fits() is checked when the solver is chosen)
    }
    return a.ride != 0;
}

}  // namespace aar
