// Solver "pcg" (aar_solver_options.solver = AAR_SOLVER_PCG; what AAR_SOLVER_AUTO picks for many entities per frame x many frames -- config 5 -- and where the
// explicit reduced system does not fit k_spcg): the damped reduced camera / marker system of one LM try,
//     S x = b,   S = U + mu I - sum_f W_f (V_f + mu I)^-1 W_f^T          (libs/sparselevmarq.h:384-400 after block elimination of the frames)
// solved by block-Jacobi-preconditioned conjugate gradients THROUGH the frame blocks -- S is never formed, nothing is factored:
//     y = S p  =  (U + mu I) p - sum_f W_f t_f,    t_f = (V_f + mu I)^-1 (W_f^T p).
// Inexact LM: an inner solve stops when BOTH |r| <= eta |b| (one forcing term, default 5e-3: kernels.h PCG_ETA_DEFAULT -- chosen for the final POSES, which then
// agree with the direct solver's to ~4e-6; a loose-then-tight sequence is an option, not the default) AND r^T M^-1 r <= eps^2 mu (absolute tolerance, default 5e-5)
// hold; the LM gain test judges the inexact step as it judges an exact one.  It is the one formulation of this solve in which nothing O(n^3) is left: the frame sum
// shards by frame and an iteration exchanges 8 n bytes (k_pcgd_* below: the same solver with the frames sharded over ranks).
//
// k_pcgf (the default form; k_pcg = two passes over W and two hand-overs per iteration, kept for deterministic mode): ONE persistent launch per solve, one workgroup
// per CU over a contiguous range of frames; every workgroup keeps its own copy of the CG vectors in LDS and updates them redundantly, so an iteration is ONE pass
// over its frames' W blocks (c = W^T p, t = V^-1 c, y -= W t with the blocks in registers across both uses) + one atomic flush of its y into two partial vectors +
// ONE grid-wide hand-over through a two-level barrier tree (grid_hop_tree: 16 first-level counters, one second-level counter, 16 flags -- no address sees more than
// 16 agent-scope operations; one counter for 256 workgroups cost 16 us per hop).  Where the forcing term is >= PCG_W32_MIN_ETA the W blocks are STORED in fp32 (pass A
// writes that copy instead of the fp64 blocks; every product and sum stays fp64; see pcgf_operator).  The convergence test is taken by every workgroup from the same
// numbers in the same order: all leave in the same iteration.
#include "geom.hpp"
#include "kernels.h"
#include "wave.hpp"

namespace aar {

struct PcgArgs {
    // blocks of the current point (pass A / pass B output; S holds U: the Schur complement kernels do not run in this mode)
    const double *U, *g0, *W, *Vinv, *hf;
    float *Wf = nullptr;                         // k_pcgf: pass A's fp32 copy of W (kernels.h, Blocks::Wf) for the set-up and the operator's passes; nullptr: fp64 W
    const int32_t *fslot_start, *fslot_ent;      // frame -> its W blocks / their entities
    const int32_t *it_ent, *it_begin, *it_end;   // work items of the entity-side passes: entity, range of its incidences in pair_rec
    const int32_t *ent_item_start;               // [A + 1] entity -> its items (1 .. PCG_MAX_ITEMS each)
    int n_items;
    const int4 *pair_rec;                        // {frame, W block, first W block of the frame, 0}
    const int32_t *ent_fixed;
    const int32_t *up_start, *up_ent;            // entity -> the other entities whose block of U can be non-zero (k_pcgf's operator; the others walk all of U)
    int A, F, n_pad;
    double mu, eta2;                             // damping; eta^2
    double abs2;                                 // eps^2 mu: the ABSOLUTE stopping threshold on r^T M^-1 r, as in k_spcg (the bound |A^-1 r| <= |r| / mu on r^T r is 1000x too careful: measured)
    int max_it;
    // work space
    double *part;                                // [n_items][28] the items' shares (27 values in the set-up, 6 per iteration)
    double *t;                                   // [6 F]
    int32_t *hop = nullptr;                      // k_pcgf: [2][PCG_HOP_WORDS] barrier words by launch parity (grid_hop_tree)
    int32_t *counter;                            // [2]  grid barrier counters: this launch uses counter[parity] (zero at entry) and clears the other
    int parity;
    double *x_out;                               // [6 A (.. n_pad)] delta_s
    int32_t *iters_out;                          // [2] iterations of this solve, running total
    int32_t *flags;
    // k_pcgf, coarse space (the rigid-motion modes of the two groups, as in spcg_kernels.hip: k_spcg_pre -- here as the plain additive two-level preconditioner
    // M^-1 = blockdiag(S_ee)^-1 + Z blockdiag(E_cc, E_mm)^-1 Z^T, the vectors being replicated in every workgroup anyway): entity rows of the pose, group sizes,
    // the accumulator of E = Z^T S Z [144] | [144] = 1 once it holds a complete sum, and from how many iterations of the previous solve on it joins (< 0: never).
    // e_refresh = 0: the sums of an EARLIER solve of the LM run are used as they are (the launcher then leaves them alone; a preconditioner only has to be
    // symmetric positive definite and fixed during a solve -- Z of today with E of some steps ago is both); without a complete sum the kernel forms it anyway
    const double *ent = nullptr;
    int C = 0, M = 0, coarse_from = -1, e_refresh = 1;
    double *eg = nullptr;
};

#ifdef AAR_PCG_STAMPS   // diagnostic build (scripts/dev/pcg_stamps.sh): s_memtime stamps of three workgroups of k_pcgf, per CG iteration
__device__ unsigned long long g_pcg_st[3][32][16];
__device__ __forceinline__ void pcg_stamp(int wg, int G, int it, int slot) {
    const int sel = wg == 0 ? 0 : (wg == G / 2 ? 1 : (wg == G - 1 ? 2 : -1));
    if (sel >= 0 && it < 32 && (threadIdx.x & 63) == 0) g_pcg_st[sel][it][slot] = __builtin_amdgcn_s_memtime();
}
#define PCG_STAMP(it, slot) do { __builtin_amdgcn_sched_barrier(0); pcg_stamp(wg, G, (it), (slot)); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PCG_STAMP(it, slot) do { } while (0)
#endif
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every workgroup of the (co-resident) grid arrives; what it stored with st_agent before is visible to all afterwards.
// false: the grid is not whole -- this workgroup waited 2^26 polls for the others, or somebody else already has (device flag 4) -- and every
// workgroup LEAVES the kernel at its next hop instead of spinning through the remaining ones (up to two per CG iteration): the launch ends within
// one time-out, the host reports AAR_ERR_NUMERIC ("chain timed out").  Co-residency itself is not assumed blindly: the grid is clamped to what the
// occupancy query admits (pcg_max_grid).
// The arrivals form a two-level tree, so that no address sees more than PCG_NY agent-scope operations per hop (256 increments of ONE address serialise at
// the memory side, ~60 ns each: the 15 us a hop used to cost at config 5; 256 pollers of one address are no better): workgroup wg arrives at counter wg % PCG_NY;
// the last arrival of a group arrives at the second level; the last arrival there raises every group's own flag, which is all a group's workgroups poll.
// cnt: [PCG_NY] first level | [1] second level | [PCG_NY] flags, one per 64 bytes; all monotonic over the hops of a launch (zero at entry).
__device__ __forceinline__ bool grid_hop_tree(int32_t *cnt, int &round, int G, int32_t *flags, int wg) {
    __shared__ int hop_dead_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    round++;
    if (threadIdx.x == 0) {
        const int g = wg % PCG_NY, n_groups = G < PCG_NY ? G : PCG_NY, members = (G - g + PCG_NY - 1) / PCG_NY;
        int32_t *l1 = cnt + g * 16, *l2 = cnt + PCG_NY * 16, *go = cnt + (PCG_NY + 1) * 16;
        if (__hip_atomic_fetch_add(l1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == round * members)
            if (__hip_atomic_fetch_add(l2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == round * n_groups)
                for (int k = 0; k < n_groups; k++) __hip_atomic_store(go + k * 16, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        int dead = 0;
        while (__hip_atomic_load(go + g * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 0x3ff) == 0 && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4)) { dead = 1; break; }
            if (spins > (1L << 26)) { atomicOr(flags, 4); dead = 1; break; }
        }
        hop_dead_s = dead;
    }
    __syncthreads();
    return hop_dead_s == 0;
}

// sum of NV per-thread values over the 256 threads, fixed order; result in out[0..NV) on every thread.  lds: 4 * NV doubles
constexpr int PCG_THREADS = 256;
constexpr int PCG_NW = PCG_THREADS / 64;
constexpr int PCGF32_THREADS = 512;   // k_pcgf<true>

template <int NV, int TH = PCG_THREADS>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off);
    }
    __syncthreads();   // (lds may still be read from the previous call)
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) lds[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TH / 64; w++) t += lds[w * NV + i];   // (the same order in every thread: the same bits)
        v[i] = t;
    }
}


// Work items of the two entity-side passes: a range of at most `chunk` (entity, frame) incidences of ONE entity, at most PCG_MAX_ITEMS per
// entity.  A workgroup per entity leaves the pass waiting for the cameras (a camera is seen in every frame: 5000 W blocks = 1.4 MB through
// one CU at the ~13 B/cycle a CU fetches -- 67 us of a 107 us iteration at config 5, while a marker's workgroup is done in half that);
// with balanced items every CU moves the same bytes, and the items of an entity are added up -- in item order -- by whoever needs the value.
constexpr int PCG_MAX_ITEMS = 8;
// LDS (dynamic): x [n] | r [n] | p [n] | Mi [6 n] (the preconditioner: built once, used every iteration) | red [4 * 27]
__global__ void __launch_bounds__(PCG_THREADS) k_pcg(const PcgArgs a) {
    extern __shared__ __align__(16) double lds[];
    const int n = 6 * a.A, G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *x = lds, *r = lds + n, *p = r + n, *Mi = p + n, *red = Mi + 6 * n;
    int32_t *counter = a.hop + a.parity * PCG_HOP_WORDS;   // (grid_hop_tree)
    int round = 0;
    if (wg == 0 && tid < 2 * PCG_NY + 1) __hip_atomic_store(a.hop + (1 - a.parity) * PCG_HOP_WORDS + tid * 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch

    // ---- set-up, first half: every item's share of sum_f W (V+mu)^-1 W^T (lower triangle, 21) and of sum_f W h_f (6) ----
    for (int it = wg; it < a.n_items; it += G) {
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
        for (int pi = a.it_begin[it] + tid; pi < a.it_end[it]; pi += PCG_THREADS) {
            const int4 rec = a.pair_rec[pi];
            const double *Wb = a.W + (size_t)rec.y * 36, *Vi = a.Vinv + (size_t)rec.x * 36, *h = a.hf + (size_t)rec.x * 6;
            double w[36], yv[36];
#pragma unroll
            for (int q = 0; q < 36; q++) w[q] = Wb[q];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) s = fma(w[i * 6 + k], Vi[k * 6 + j], s);
                    yv[i * 6 + j] = s;
                }
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) s = fma(yv[i * 6 + k], w[j * 6 + k], s);
                    acc[i * (i + 1) / 2 + j] += s;
                }
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) s = fma(w[i * 6 + k], h[k], s);
                acc[21 + i] += s;
            }
        }
        block_sum<27>(acc, red);
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 27; i++) st_agent(a.part + (size_t)it * 28 + i, acc[i]);
        }
    }
    if (!grid_hop_tree(counter, round, G, a.flags, wg)) return;
    // ---- set-up, second half, redundantly in every workgroup (one thread per entity): the items' shares in item order, the diagonal
    //      block of S inverted straight into LDS, the right-hand side; x = 0, r = b ----
    for (int e = tid; e < a.A; e += PCG_THREADS) {
        double out[36], be[6];
        if (a.ent_fixed[e]) {
#pragma unroll
            for (int i = 0; i < 36; i++) out[i] = (i % 7 == 0) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) be[i] = 0.0;
        } else {
            double acc[27];
#pragma unroll
            for (int i = 0; i < 27; i++) acc[i] = 0.0;
            const int i0 = a.ent_item_start[e], i1 = a.ent_item_start[e + 1];
            for (int it = i0; it < i1; it++)
#pragma unroll
                for (int i = 0; i < 27; i++) acc[i] += ld_agent(a.part + (size_t)it * 28 + i);
            double m[6][6];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    const double v = a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j] + (i == j ? a.mu : 0.0) - acc[i * (i + 1) / 2 + j];
                    m[i][j] = v; m[j][i] = v;
                }
            if (!spd6_inverse(m, out) && wg == 0) atomicOr(a.flags, 2);
#pragma unroll
            for (int i = 0; i < 6; i++) be[i] = a.g0[6 * e + i] - acc[21 + i];
        }
#pragma unroll
        for (int i = 0; i < 36; i++) Mi[e * 36 + i] = out[i];
#pragma unroll
        for (int i = 0; i < 6; i++) { r[6 * e + i] = be[i]; x[6 * e + i] = 0.0; }
    }
    __syncthreads();
    // (everybody has read the items' set-up shares before any iteration overwrites the buffer: the first entity pass is behind the next hand-over)

    // ---- z = Minv r, p = z ----
    double rz = 0.0, bb = 0.0;
    {
        double s[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += PCG_THREADS) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            p[i] = z;
            s[0] += r[i] * z;
            s[1] += r[i] * r[i];
        }
        block_sum<2>(s, red);
        rz = s[0]; bb = s[1];
    }
    __syncthreads();

    int it_cg = 0;
    double rr = bb;
    while (it_cg < a.max_it && (rr > a.eta2 * bb || rz > a.abs2) && bb > 0.0) {
        // ---- frame pass: t_f = (V_f + mu I)^-1 (W_f^T p), one wavefront per frame ----
        for (int f = wg * (PCG_THREADS / 64) + wave; f < a.F; f += G * (PCG_THREADS / 64)) {
            const int s0 = a.fslot_start[f], s1 = a.fslot_start[f + 1];
            double c[6] = {0, 0, 0, 0, 0, 0};
            for (int s = s0 + lane; s < s1; s += 64) {
                const int e = a.fslot_ent[s];
                const double2 *wb = reinterpret_cast<const double2 *>(a.W + (size_t)s * 36);
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double pe = p[6 * e + i];
                    const double2 w0 = wb[3 * i], w1 = wb[3 * i + 1], w2 = wb[3 * i + 2];
                    c[0] = fma(w0.x, pe, c[0]); c[1] = fma(w0.y, pe, c[1]); c[2] = fma(w1.x, pe, c[2]);
                    c[3] = fma(w1.y, pe, c[3]); c[4] = fma(w2.x, pe, c[4]); c[5] = fma(w2.y, pe, c[5]);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int i = 0; i < 6; i++) c[i] += __shfl_xor(c[i], off);
            if (lane < 6) {
                double tv = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) tv = fma(a.Vinv[(size_t)f * 36 + lane * 6 + k], c[k], tv);
                st_agent(a.t + (size_t)f * 6 + lane, tv);
            }
        }
        if (!grid_hop_tree(counter, round, G, a.flags, wg)) return;
        // ---- entity pass, by item: its share of -sum_f W_ef t_f; the entity's first item also carries (U p)_e ----
        for (int it = wg; it < a.n_items; it += G) {
            const int e = a.it_ent[it];
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (int pi = a.it_begin[it] + tid; pi < a.it_end[it]; pi += PCG_THREADS) {
                const int4 rec = a.pair_rec[pi];
                const double2 *wb = reinterpret_cast<const double2 *>(a.W + (size_t)rec.y * 36);
                double tv[6];
#pragma unroll
                for (int k = 0; k < 6; k++) tv[k] = ld_agent(a.t + (size_t)rec.x * 6 + k);
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double2 w0 = wb[3 * i], w1 = wb[3 * i + 1], w2 = wb[3 * i + 2];
                    acc[i] -= w0.x * tv[0] + w0.y * tv[1] + w1.x * tv[2] + w1.y * tv[3] + w2.x * tv[4] + w2.y * tv[5];
                }
            }
            if (it == a.ent_item_start[e] && !a.ent_fixed[e]) {   // row e of the symmetric U (lower triangle stored)
                for (int bq = tid; bq < a.A; bq += PCG_THREADS) {
                    if (a.ent_fixed[bq]) continue;                  // (its p is zero)
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            const double u = bq < e ? a.U[(size_t)(6 * e + i) * a.n_pad + 6 * bq + j]
                                                    : (bq > e ? a.U[(size_t)(6 * bq + j) * a.n_pad + 6 * e + i]
                                                              : (j <= i ? a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j] : a.U[(size_t)(6 * e + j) * a.n_pad + 6 * e + i]));
                            acc[i] = fma(u, p[6 * bq + j], acc[i]);
                        }
                }
            }
            block_sum<6>(acc, red);
            if (tid < 6) st_agent(a.part + (size_t)it * 28 + tid, acc[tid]);
        }
        if (!grid_hop_tree(counter, round, G, a.flags, wg)) return;
        // ---- vector updates, redundantly in every workgroup (same numbers, same order: same decisions) ----
        // y = S p: the items' shares of every entry, ALL requested before the first is used (an agent-scope load is ~1 us: one round
        // trip, not one per item), added in item order; + mu p; identity rows for gauge entities
        double yl[24];
        double pAp = 0.0;
        {
            double s[1] = {0.0};
            int ny = 0;
            for (int i = tid; i < n; i += PCG_THREADS, ny++) {
                const int e = i / 6, row = i - 6 * e;
                const int i0 = a.ent_item_start[e], cnt = a.ent_item_start[e + 1] - i0;
                double sh[PCG_MAX_ITEMS];
#pragma unroll
                for (int k = 0; k < PCG_MAX_ITEMS; k++) sh[k] = k < cnt ? ld_agent(a.part + (size_t)(i0 + k) * 28 + row) : 0.0;
                double yv = a.mu * p[i];
#pragma unroll
                for (int k = 0; k < PCG_MAX_ITEMS; k++) yv += sh[k];
                if (a.ent_fixed[e]) yv = p[i];
                if (ny < 24) yl[ny] = yv;
                s[0] = fma(p[i], yv, s[0]);
            }
            block_sum<1>(s, red);
            pAp = s[0];
        }
        const double alpha = rz / pAp;
        {
            int ny = 0;
            for (int i = tid; i < n; i += PCG_THREADS, ny++) {
                x[i] = fma(alpha, p[i], x[i]);
                r[i] = fma(-alpha, yl[ny < 24 ? ny : 23], r[i]);
            }
        }
        __syncthreads();
        double s2[2] = {0.0, 0.0};
        double zloc[24];   // this thread's entries of z (n <= 24 * 256)
        int nz = 0;
        for (int i = tid; i < n; i += PCG_THREADS, nz++) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            if (nz < 24) zloc[nz] = z;
            s2[0] += r[i] * z;
            s2[1] += r[i] * r[i];
        }
        block_sum<2>(s2, red);
        const double beta = s2[0] / rz;
        rz = s2[0];
        rr = s2[1];
        nz = 0;
        for (int i = tid; i < n; i += PCG_THREADS, nz++) p[i] = fma(beta, p[i], zloc[nz < 24 ? nz : 23]);
        __syncthreads();
        it_cg++;
    }
    if (wg == 0) {
        for (int i = tid; i < n; i += PCG_THREADS) a.x_out[i] = x[i];
        for (int i = n + tid; i < a.n_pad; i += PCG_THREADS) a.x_out[i] = 0.0;
        if (tid == 0) { a.iters_out[0] = it_cg; a.iters_out[1] += it_cg; a.iters_out[2] += 1; }   // (stopped by max_it: still an inexact step, the LM gain test judges it)
    }
}

// ------------------------------------------------------------------------------------------------
// k_pcgf: the same solver with the operator applied in ONE pass over the W blocks and ONE grid-wide hand-over per iteration
// (k_pcg above: two of each).  The wavefront that owns frame f keeps the frame's W blocks in registers across both uses:
//     c = sum_s W_s^T p_e(s),   t = (V_f + mu I)^-1 c,   y_e(s) -= W_s t
// and the contributions to y are SCATTERED instead of gathered by entity: first into the workgroup's own copy of y in LDS (ds_add_f64: a workgroup
// walks ~F / G frames, the cameras are in every one of them), then -- once per iteration -- into one global vector with fp64 atomics (n per workgroup).
// (U p)_e rides in the same LDS vector (entities dealt over the workgroups).  After the hand-over every workgroup reads the n sums past the L2 and
// updates its own copy of x, r, p redundantly, as before.  Three y buffers rotate: iteration i adds into buffer i % 3, reads it after the hand-over and
// clears buffer (i + 1) % 3, whose last readers (iteration i - 2) passed a hand-over ago.  The set-up (diagonal blocks of S, right-hand side) takes the
// same route: per slot in LDS, one atomic flush, one hand-over.
// The sums come in whatever order the atomics arrive: the deterministic mode keeps k_pcg (item shares added in item order).
// LDS (dynamic): x [n] | r [n] | p [n] | Mi [6 n] (the set-up's accumulators [A][27] live there first) | yacc [n] | red [4 * 27]
// yg: [3][n_pad] zero at entry; sg: [A][28] zero at entry (the launcher clears both)
// ------------------------------------------------------------------------------------------------
// every slot's share of W (V+mu)^-1 W^T (lower triangle, 21) and of W h_f (6) for the frames dealt to this workgroup, into sacc [A][27] (LDS, zeroed here)
// ---- coarse space of k_pcgf ----
// zd: per entity Jinv (9, row-major) | t (3): Z_e = [[J_l^-1, 0], [-[t]x, I]] (rows: the entity's six parameters; columns: rotation, translation of the group's rigid
// motion); an entity without modes (fixed, or not a pose) has an all-zero record -- the diagonal of a J_l^-1 is never all zero (its trace is at least 1)
__device__ __forceinline__ bool pcgf_on(const double *__restrict__ zd_e) { return zd_e[0] != 0.0 || zd_e[4] != 0.0 || zd_e[8] != 0.0; }
__device__ __forceinline__ void pcgf_zrow(const double *__restrict__ zd_e, int row, double (&zr)[6]) {
    if (row < 3) {
        zr[0] = zd_e[3 * row]; zr[1] = zd_e[3 * row + 1]; zr[2] = zd_e[3 * row + 2]; zr[3] = 0.0; zr[4] = 0.0; zr[5] = 0.0;
    } else {
        const double t0 = zd_e[9], t1 = zd_e[10], t2 = zd_e[11];
        zr[0] = row == 3 ? 0.0 : (row == 4 ? -t2 : t1);
        zr[1] = row == 3 ? t2 : (row == 4 ? 0.0 : -t0);
        zr[2] = row == 3 ? -t1 : (row == 4 ? t0 : 0.0);
        const double on = pcgf_on(zd_e) ? 1.0 : 0.0;
        zr[3] = row == 3 ? on : 0.0; zr[4] = row == 4 ? on : 0.0; zr[5] = row == 5 ? on : 0.0;
    }
}
__device__ __forceinline__ void pcgf_zfull(const double *__restrict__ zd_e, double (&Z)[36]) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double zr[6];
        pcgf_zrow(zd_e, i, zr);
#pragma unroll
        for (int c = 0; c < 6; c++) Z[6 * i + c] = zr[c];
    }
}
__device__ __forceinline__ bool pcgf_modes(const PcgArgs &a, int e) { return e < a.C + a.M && a.ent_fixed[e] == 0; }
// the table from the entity rows {R, t, J_l} (one thread per entity; the caller synchronises)
template <int TH>
__device__ __forceinline__ void pcgf_build_zd(const PcgArgs &a, double *__restrict__ zd) {
    for (int e = threadIdx.x; e < a.A; e += TH) {
        double z12[12];
#pragma unroll
        for (int q = 0; q < 12; q++) z12[q] = 0.0;
        if (pcgf_modes(a, e)) {
            const double *row = a.ent + (size_t)e * ENT_STRIDE;
            const double *J = row + 12;
            const double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
            const double id = 1.0 / (J[0] * c00 + J[1] * c01 + J[2] * c02);
            z12[0] = c00 * id; z12[1] = (J[2] * J[7] - J[1] * J[8]) * id; z12[2] = (J[1] * J[5] - J[2] * J[4]) * id;
            z12[3] = c01 * id; z12[4] = (J[0] * J[8] - J[2] * J[6]) * id; z12[5] = (J[2] * J[3] - J[0] * J[5]) * id;
            z12[6] = c02 * id; z12[7] = (J[1] * J[6] - J[0] * J[7]) * id; z12[8] = (J[0] * J[4] - J[1] * J[3]) * id;
            z12[9] = row[9]; z12[10] = row[10]; z12[11] = row[11];
        }
#pragma unroll
        for (int q = 0; q < 12; q++) zd[12 * e + q] = z12[q];
    }
}
// c = blockdiag(E_cc, E_mm)^-1 Z^T rv into cs (every thread): wavefront ca < 6 forms entries ca of Z_c^T rv and Z_m^T rv (its lanes stride over the entities: column ca
// of Z_e against the entity's six entries of rv), ONE sum over the wavefront each; twelve threads apply the inverse blocks; everybody reads c back.  cvec: 24 doubles
template <int TH>
__device__ __forceinline__ void pcgf_coarse_coef(const PcgArgs &a, const double *__restrict__ zd, const double *__restrict__ einv, double *__restrict__ cvec,
                                                 const double *__restrict__ rv, double (&cs)[12]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __syncthreads();   // (cvec may still be read from the previous call; rv is complete)
    for (int ca = wave; ca < 6; ca += TH / 64) {   // (wave-uniform)
        double sc = 0.0, sm = 0.0;
        for (int e = lane; e < a.C + a.M; e += 64) {
            const double *z = zd + 12 * e;
            double v;
            if (ca < 3) {   // column ca of [[Jinv], [-[t]x]]
                const double t0 = z[9], t1 = z[10], t2 = z[11];
                const double n0 = ca == 0 ? 0.0 : (ca == 1 ? t2 : -t1), n1 = ca == 0 ? -t2 : (ca == 1 ? 0.0 : t0), n2 = ca == 0 ? t1 : (ca == 1 ? -t0 : 0.0);
                v = z[ca] * rv[6 * e] + z[3 + ca] * rv[6 * e + 1] + z[6 + ca] * rv[6 * e + 2] + (n0 * rv[6 * e + 3] + n1 * rv[6 * e + 4] + n2 * rv[6 * e + 5]);
            } else {
                v = pcgf_on(z) ? rv[6 * e + ca] : 0.0;
            }
            if (e < a.C) sc += v; else sm += v;
        }
        sc = wave_sum_dpp(sc); sm = wave_sum_dpp(sm);
        if (lane == 0) { cvec[ca] = sc; cvec[6 + ca] = sm; }
    }
    __syncthreads();
    if (tid < 12) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) v = fma(einv[36 * (tid / 6) + 6 * (tid % 6) + k], cvec[6 * (tid / 6) + k], v);
        cvec[12 + tid] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 12; q++) cs[q] = cvec[12 + q];
}
__device__ __forceinline__ double pcgf_coarse_add(const PcgArgs &a, const double *__restrict__ zd, int e, int row, const double (&cs)[12]) {   // (Z c)_i
    double zr[6];
    pcgf_zrow(zd + 12 * e, row, zr);
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 6; q++) v = fma(zr[q], e < a.C ? cs[q] : cs[6 + q], v);
    return v;
}
// the two inverse blocks of E (lower triangles of the sums at Eg, 12 x 12 row-major) by threads 0 and 1 into einv [72]; a group without modes, or a block that is not
// positive definite in floating point: no modes (zeros)
// Gl != nullptr: the sums at Eg were formed WITHOUT the damping's Z^T (mu I) Z, which is added here as mu * Gl (Gl = Z^T Z of today's Z, 12 x 12 row-major in LDS)
__device__ __forceinline__ void pcgf_invert_E(const double *Eg, double *__restrict__ einv, bool agent_loads, const double *__restrict__ Gl = nullptr, double mu = 0.0) {
    const int tid = threadIdx.x;
    if (tid >= 2) return;
    double eb[6][6], inv[36];
    bool have = false;
#pragma unroll
    for (int pp = 0; pp < 6; pp++)
#pragma unroll
        for (int q = 0; q <= pp; q++) {
            const double *src = Eg + 12 * (6 * tid + pp) + 6 * tid + q;
            double v = agent_loads ? ld_agent(src) : *src;
            if (Gl && v != 0.0) v = fma(mu, Gl[12 * (6 * tid + pp) + 6 * tid + q], v);
            eb[pp][q] = v; eb[q][pp] = v;
            have = have || v != 0.0;
        }
    if (!have) {
#pragma unroll
        for (int pp = 0; pp < 6; pp++) eb[pp][pp] = 1.0;
    }
    if (!spd6_inverse(eb, inv)) have = false;
#pragma unroll
    for (int q = 0; q < 36; q++) einv[36 * tid + q] = have ? inv[q] : 0.0;
}

// E = Z^T S Z, this workgroup's share into Ews [144] (LDS, zeroed here; row-major 12 x 12: cameras' modes 0..5, markers' 6..11):
//   frames:  - sum_f P_f^T (V_f + mu I)^-1 P_f,   P_f = W_f^T Z = sum over the frame's slots of W_ef^T Z_e  (6 x 12: one more use of the blocks the set-up reads),
//   U:       + Z^T (U + mu I) Z over the stored blocks of the entities dealt to this workgroup.
// wl: per-wavefront scratch [TH / 64][144]
// add_mu: the damping's Z^T (mu I) Z is added here (one rank; with ranks every rank holds a PARTIAL U: only rank 0 adds it)
template <int TH>
__device__ __forceinline__ void pcgf_setup_coarse(const PcgArgs &a, const double *__restrict__ zd, double *__restrict__ Ews, double *__restrict__ wl, int wg, int G, bool add_mu = true) {
    constexpr int NW = TH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 144; i += TH) Ews[i] = 0.0;
    __syncthreads();
    double *Pl = wl + 144 * wave, *Ql = Pl + 72;
    double eacc[3] = {0.0, 0.0, 0.0};
    const int f_lo = (int)((long long)wg * a.F / G), f_hi = (int)((long long)(wg + 1) * a.F / G);
    for (int f = f_lo + wave; f < f_hi; f += NW) {
        const int s0 = a.fslot_start[f], s1 = a.fslot_start[f + 1];
        const double *Vi = a.Vinv + (size_t)f * 36;
        // ONE pass over the frame's slots (entity-ascending: the cameras' come first and, being at most 64, all fall into the first round).  A lane forms the 36
        // entries of T = W_ef^T Z_e of its slot one by one: the cameras' are summed over the wavefront at once (first round only), the markers' accumulate in
        // 36 registers over the rounds and are summed at the end -- never two sets of accumulators alive
        double Tm[36];
#pragma unroll
        for (int q = 0; q < 36; q++) Tm[q] = 0.0;
        if (s0 == s1 && lane < 36) Pl[lane] = 0.0;   // (a frame without slots: no first round to write the cameras' sums)
        for (int s = s0 + lane, rnd = 0; s - lane < s1; s += 64, rnd++) {
            const bool in = s < s1;
            const int e = in ? a.fslot_ent[s] : 0;
            const double *z = zd + 12 * e;
            const bool on = in && pcgf_on(z), cam = on && e < a.C, mk = on && e >= a.C;
            double w[36];
#pragma unroll
            for (int q = 0; q < 36; q++) w[q] = 0.0;
            if (on) {
                if (a.Wf) {
                    const float4 *q4 = reinterpret_cast<const float4 *>(a.Wf + (size_t)s0 * 36) + (s - s0);
                    const int kf = s1 - s0;
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        const float4 v = q4[(size_t)q * kf];
                        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
                    }
                } else {
                    const double *Wb = a.W + (size_t)s * 36;
#pragma unroll
                    for (int q = 0; q < 36; q++) w[q] = Wb[q];
                }
            }
            const double t0 = z[9], t1 = z[10], t2 = z[11];
#pragma unroll
            for (int k = 0; k < 6; k++) {   // T(k, a) = sum_i W(i, k) Z_e(i, a)
                const double w0 = w[k], w1 = w[6 + k], w2 = w[12 + k], w3 = w[18 + k], w4 = w[24 + k], w5 = w[30 + k];
                double tq[6];
                tq[0] = w0 * z[0] + w1 * z[3] + w2 * z[6] + (w5 * t1 - w4 * t2);
                tq[1] = w0 * z[1] + w1 * z[4] + w2 * z[7] + (w3 * t2 - w5 * t0);
                tq[2] = w0 * z[2] + w1 * z[5] + w2 * z[8] + (w4 * t0 - w3 * t1);
                tq[3] = w3; tq[4] = w4; tq[5] = w5;
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    if (rnd == 0) {   // (wave-uniform)
                        const double v = wave_sum_dpp(cam ? tq[c] : 0.0);
                        if (lane == 0) Pl[6 * k + c] = v;
                    }
                    Tm[6 * k + c] += mk ? tq[c] : 0.0;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 36; q++) {
            const double v = wave_sum_dpp(Tm[q]);
            if (lane == 0) Pl[36 + q] = v;
        }
        __builtin_amdgcn_wave_barrier();
        for (int o = lane; o < 72; o += 64) {   // Q(k, b12) = sum_j Vinv(k, j) P(j, b12)
            const int grp = o / 36, k = (o % 36) / 6, b = o % 6;
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) v = fma(Vi[6 * k + j], Pl[36 * grp + 6 * j + b], v);
            Ql[o] = v;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int o = lane + 64 * u;
            if (o < 144) {
                const int a12 = o / 12, b12 = o % 12;
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) v = fma(Pl[36 * (a12 / 6) + 6 * k + a12 % 6], Ql[36 * (b12 / 6) + 6 * k + b12 % 6], v);
                eacc[u] -= v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int u = 0; u < 3; u++) { const int o = lane + 64 * u; if (o < 144 && eacc[u] != 0.0) atomicAdd(Ews + o, eacc[u]); }
    // Z^T (U + mu I) Z: the block rows dealt to this workgroup (as in pcgf_operator), one thread per stored block
    for (int e = wg; e < a.A; e += G) {
        if (!pcgf_on(zd + 12 * e)) continue;   // (uniform per workgroup)
        const int n0 = a.up_start[e], n1 = a.up_start[e + 1], ge = e >= a.C ? 6 : 0;
        double Ze[36];
        pcgf_zfull(zd + 12 * e, Ze);
        for (int q = n0 - 1 + tid; q < n1; q += TH) {
            const int b = q < n0 ? e : a.up_ent[q];
            if (b > e || (q >= n0 && b == e) || !pcgf_on(zd + 12 * b)) continue;
            double Zb[36], X[36];
            pcgf_zfull(zd + 12 * b, Zb);
#pragma unroll
            for (int i = 0; i < 6; i++) {   // X = U_eb Z_b  (the diagonal block: lower triangle stored, mu on its diagonal)
                double u[6];
#pragma unroll
                for (int j = 0; j < 6; j++)
                    u[j] = b == e ? ((j <= i ? a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j] : a.U[(size_t)(6 * e + j) * a.n_pad + 6 * e + i]) + ((i == j && add_mu) ? a.mu : 0.0))
                                  : a.U[(size_t)(6 * e + i) * a.n_pad + 6 * b + j];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v = 0.0;
#pragma unroll
                    for (int j = 0; j < 6; j++) v = fma(u[j], Zb[6 * j + c], v);
                    X[6 * i + c] = v;
                }
            }
            const int gb = b >= a.C ? 6 : 0;
#pragma unroll
            for (int aa = 0; aa < 6; aa++)
#pragma unroll
                for (int c = 0; c < 6; c++) {   // R = Z_e^T X
                    double v = 0.0;
#pragma unroll
                    for (int i = 0; i < 6; i++) v = fma(Ze[6 * i + aa], X[6 * i + c], v);
                    atomicAdd(Ews + 12 * (ge + aa) + gb + c, v);
                    if (b != e) atomicAdd(Ews + 12 * (gb + c) + ge + aa, v);
                }
        }
    }
    __syncthreads();
}

template <int TH = PCG_THREADS>
__device__ __forceinline__ void pcgf_setup_slots(const PcgArgs &a, double *__restrict__ sacc, int wg, int G) {
    constexpr int NW = TH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 27 * a.A; i += TH) sacc[i] = 0.0;
    __syncthreads();
    const int f_lo = (int)((long long)wg * a.F / G), f_hi = (int)((long long)(wg + 1) * a.F / G);   // the workgroup's frames (see pcgf_operator)
    for (int f = f_lo + wave; f < f_hi; f += NW) {
        const int s0 = a.fslot_start[f], s1 = a.fslot_start[f + 1];
        const double *Vi = a.Vinv + (size_t)f * 36, *h = a.hf + (size_t)f * 6;
        for (int s = s0 + lane; s < s1; s += 64) {
            const int e = a.fslot_ent[s];
            if (a.ent_fixed[e]) continue;
            double w[36], yv[36];
            if (a.Wf) {   // pass A's fp32 copy (kernels.h, Blocks::Wf): half the bytes, consecutive lanes read consecutive pieces
                const float4 *q4 = reinterpret_cast<const float4 *>(a.Wf + (size_t)s0 * 36) + (s - s0);
                const int kf = s1 - s0;
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    const float4 v = q4[(size_t)q * kf];
                    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
                }
            } else {
                const double *Wb = a.W + (size_t)s * 36;
#pragma unroll
                for (int q = 0; q < 36; q++) w[q] = Wb[q];
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) t = fma(w[i * 6 + k], Vi[k * 6 + j], t);
                    yv[i * 6 + j] = t;
                }
            double *dst = sacc + 27 * e;
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) t = fma(yv[i * 6 + k], w[j * 6 + k], t);
                    atomicAdd(dst + i * (i + 1) / 2 + j, t);
                }
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) t = fma(w[i * 6 + k], h[k], t);
                atomicAdd(dst + 21 + i, t);
            }
        }
    }
    __syncthreads();
}

// this workgroup's share of  y = U p - sum_f W_f (V_f + mu I)^-1 W_f^T p  (no mu p term) into yacc [n] (LDS, zeroed here): ONE pass over the W blocks of
// its frames -- c = W^T p, t = Vinv c, y -= W t, the first two rounds of a frame's slot list staying in registers across both uses -- and (U p)_e for the
// entities dealt to it.  p: this workgroup's copy of the search direction (LDS)
// W32: the frame pass reads the fp32 copy of W (a.Wf; half the bytes of the pass, which is HBM-bound at config 5) and widens on use.  Where that copy exists
// NOTHING reads fp64 blocks any more: pass A writes Wf INSTEAD of W (a.W is then stale -- never written -- unless aar_eval_normal_equations asked for it through
// want_w64), and the set-up pass (right-hand side, preconditioner) and the back-substitution read Wf too; a rounding of 6e-8 is far below the forcing terms
// this storage is allowed at (PCG_W32_MIN_ETA) -- final poses unchanged (scripts/experiments/pcg_w_float.py; profiles/r05_attempts.txt section 5).  The kernels
// that read a.W unconditionally (k_pcg, k_pcgd_setup / k_pcgd_iter, the Schur kernels) must therefore never run on a block set that has Wf: launch_pcg and
// launch_pcgd_* refuse (flag 2 -> AAR_ERR_NUMERIC) instead of reading stale blocks
// a W block in registers: 18 double2 (W32: 9 float4), row-major, widened on use
template <bool W32>
struct PcgBlk { typename std::conditional<W32, float4, double2>::type v[W32 ? 9 : 18]; };
template <bool W32>
__device__ __forceinline__ void pcgf_load_blk(const PcgArgs &a, PcgBlk<W32> &b, int s0, int s1, int s) {
    if constexpr (W32) {
        const float4 *q = reinterpret_cast<const float4 *>(a.Wf + (size_t)s0 * 36) + (s - s0);
        const int kf = s1 - s0;
#pragma unroll
        for (int u = 0; u < 9; u++) b.v[u] = q[(size_t)u * kf];
    } else {
        const double2 *q = reinterpret_cast<const double2 *>(a.W + (size_t)s * 36);
#pragma unroll
        for (int u = 0; u < 18; u++) b.v[u] = q[u];
    }
}
// RESIDENT BLOCKS (fp32 storage only).  A workgroup's frames are the same in every CG iteration, and so is their deal to its wavefronts, so blocks can stay in registers
// for the whole damped solve instead of being streamed once per iteration: RES = 1: the first round of a wavefront's first frame (9 float4 = 36 registers per lane:
// a fifth of a wavefront's 4.9 rounds at config 5); 2: + its second round; 3: + the first round of the second frame (61 %).  They are loaded after the set-up passes
// (pcgf_load_resident), so the kernel's peak is the larger of the two phases, not their sum -- and still only RES = 1 fits: the CG loop holds ~220 registers of the
// 256 a wavefront has at two per SIMD (hoisted addresses, the two streamed blocks of the generic frames), 2 / 3 spill 63 / 162 registers and LOSE
// (profiles/r06_attempts.txt section 5; config 5: k_pcg 309 -> 296 us with one block, 336 / 387 with two / three; the fp64-block kernel, one wavefront per SIMD, holds
// TWO blocks of 18 double2: 455 -> 438 us; at ONE wavefront per SIMD -- THX = 256, 512 registers
// per lane -- 4 / 6 / 7 blocks fit without spills, 114 - 231 of them AGPRs, and the kernel takes 360 / 350 / 346 us: what the frame pass gains the set-up passes and the
// vector updates lose with half the wavefronts).  Instantiated: <true, 0 / 1>, <false, 0 / 2>.
// eres: the blocks' entities (-1: no such slot).
template <bool W32, int TH, int RES>
__device__ __forceinline__ void pcgf_load_resident(const PcgArgs &a, int wg, int G, PcgBlk<W32> (&res)[RES > 0 ? RES : 1], int (&eres)[RES > 0 ? RES : 1]) {
    if constexpr (RES > 0) {
        constexpr int NW = TH / 64;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int f_lo = (int)((long long)wg * a.F / G), f_hi = (int)((long long)(wg + 1) * a.F / G);
#pragma unroll
        for (int q = 0; q < RES; q++) {
            const int f = f_lo + wave + (q / 2) * NW, off = (q & 1) * 64;   // block q: round q % 2 of the wavefront's frame q / 2
            int e = -1;
            PcgBlk<W32> b;
#pragma unroll
            for (int u = 0; u < (W32 ? 9 : 18); u++) b.v[u] = typename std::conditional<W32, float4, double2>::type{};
            if (f < f_hi) {
                const int s0 = a.fslot_start[f], s1 = a.fslot_start[f + 1];
                if (s0 + lane + off < s1) { e = a.fslot_ent[s0 + lane + off]; pcgf_load_blk<W32>(a, b, s0, s1, s0 + lane + off); }
            }
            res[q] = b;
            eres[q] = e;
        }
    }
}

template <bool W32, int TH = PCG_THREADS, int RES = 0>
__device__ __forceinline__ void pcgf_operator(const PcgArgs &a, const double *__restrict__ p, double *__restrict__ yacc, double *__restrict__ red, int wg, int G, int st_it,
                                              const PcgBlk<W32> (&res)[RES > 0 ? RES : 1], const int (&eres)[RES > 0 ? RES : 1]) {
    constexpr int NW = TH / 64;
    const int n = 6 * a.A, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < n; i += TH) yacc[i] = 0.0;
    __syncthreads();
    if (wave == 0) PCG_STAMP(st_it, 1);
    // (U p) for the block rows dealt to this workgroup, FIRST (the wavefronts without blocks go straight to their frames): one thread per stored block
    // U_eb, b < e, of the blocks that exist (U is block-sparse: entities that share an observation -- camera x marker; at config 5, 16 of a marker's 216),
    // read row-wise in 16-byte pieces, used twice -- y_e += U_eb p_b and y_b += U_eb^T p_e -- and added to the workgroup's y in LDS like the frames'
    // contributions: no reduction, no barrier.  (p of a gauge entity is zero and its y is overridden later.)
    for (int e = wg; e < a.A; e += G) {
        if (a.ent_fixed[e]) continue;   // (uniform per workgroup)
        const int n0 = a.up_start[e], n1 = a.up_start[e + 1];
        for (int q = n0 - 1 + tid; q < n1; q += TH) {
            if (q < n0) {   // the diagonal block (lower triangle stored)
                double acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        const double u = j <= i ? a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j] : a.U[(size_t)(6 * e + j) * a.n_pad + 6 * e + i];
                        acc[i] = fma(u, p[6 * e + j], acc[i]);
                    }
#pragma unroll
                for (int i = 0; i < 6; i++) atomicAdd(yacc + 6 * e + i, acc[i]);
                continue;
            }
            const int b = a.up_ent[q];
            if (b >= e || a.ent_fixed[b]) continue;
            double2 u[6][3];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double2 *row = reinterpret_cast<const double2 *>(a.U + (size_t)(6 * e + i) * a.n_pad + 6 * b);
                u[i][0] = row[0]; u[i][1] = row[1]; u[i][2] = row[2];
            }
            double pb[6], pe[6], yb[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 6; k++) { pb[k] = p[6 * b + k]; pe[k] = p[6 * e + k]; }
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double ye = u[i][0].x * pb[0] + u[i][0].y * pb[1] + u[i][1].x * pb[2] + u[i][1].y * pb[3] + u[i][2].x * pb[4] + u[i][2].y * pb[5];
                atomicAdd(yacc + 6 * e + i, ye);
                yb[0] = fma(u[i][0].x, pe[i], yb[0]); yb[1] = fma(u[i][0].y, pe[i], yb[1]); yb[2] = fma(u[i][1].x, pe[i], yb[2]);
                yb[3] = fma(u[i][1].y, pe[i], yb[3]); yb[4] = fma(u[i][2].x, pe[i], yb[4]); yb[5] = fma(u[i][2].y, pe[i], yb[5]);
            }
#pragma unroll
            for (int j = 0; j < 6; j++) atomicAdd(yacc + 6 * b + j, yb[j]);
        }
    }
    if (wave == 0) PCG_STAMP(st_it, 2);
    // Frames: a CONTIGUOUS range per workgroup (F / G of them, to one), dealt round-robin to its wavefronts -- every workgroup takes the same number of rounds
    // (dealt wave-major over the whole grid, 5000 frames over 2048 wavefronts left 113 workgroups with three rounds and 143 with two)
    const int f_lo = (int)((long long)wg * a.F / G), f_hi = (int)((long long)(wg + 1) * a.F / G);
    typedef PcgBlk<W32> Blk;
    // (second: the block's second use in a frame, behind the wavefront sums.  Its widened values are made again from the fp32 registers -- an empty asm keeps the
    //  compiler from holding all 72 doubles of the first use alive across the sums, which is what filled the register file)
    auto row = [&](const Blk &b, int i, double (&r)[6], bool second = false) {   // row i of the block as doubles
        if constexpr (W32) {
            const float *fp = reinterpret_cast<const float *>(b.v);
#pragma unroll
            for (int j = 0; j < 6; j++) {
                float f = fp[6 * i + j];
                if (second) asm volatile("" : "+v"(f));
                r[j] = (double)f;
            }
        } else {
            r[0] = b.v[3 * i].x; r[1] = b.v[3 * i].y; r[2] = b.v[3 * i + 1].x; r[3] = b.v[3 * i + 1].y; r[4] = b.v[3 * i + 2].x; r[5] = b.v[3 * i + 2].y;
        }
    };
    // one frame; R0 / R1: which resident block holds its first / second round (-1: fetched here)
    auto frame = [&](int f, auto R0, auto R1) {
        constexpr int r0 = decltype(R0)::value, r1 = decltype(R1)::value;
        const int s0 = a.fslot_start[f], s1 = a.fslot_start[f + 1];
        const double *Vi = a.Vinv + (size_t)f * 36;
        Blk w0l, w1l;
        int e0 = -1, e1 = -1;
        double c[6] = {0, 0, 0, 0, 0, 0};
        auto gather_c = [&](const Blk &b, int e) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double pe = p[6 * e + i];
                double r[6];
                row(b, i, r);
#pragma unroll
                for (int j = 0; j < 6; j++) c[j] = fma(r[j], pe, c[j]);
            }
        };
        if constexpr (r0 >= 0) e0 = eres[r0 >= 0 ? r0 : 0];
        else if (s0 + lane < s1) { e0 = a.fslot_ent[s0 + lane]; pcgf_load_blk<W32>(a, w0l, s0, s1, s0 + lane); }
        if constexpr (r1 >= 0) e1 = eres[r1 >= 0 ? r1 : 0];
        else if (s0 + lane + 64 < s1) { e1 = a.fslot_ent[s0 + lane + 64]; pcgf_load_blk<W32>(a, w1l, s0, s1, s0 + lane + 64); }
        auto with0 = [&](auto fn) { if constexpr (r0 >= 0) fn(res[r0 >= 0 ? r0 : 0]); else fn(w0l); };
        auto with1 = [&](auto fn) { if constexpr (r1 >= 0) fn(res[r1 >= 0 ? r1 : 0]); else fn(w1l); };
        if (e0 >= 0) with0([&](const Blk &b) { gather_c(b, e0); });
        if (e1 >= 0) with1([&](const Blk &b) { gather_c(b, e1); });
        for (int s = s0 + lane + 128; s < s1; s += 64) {
            const int e = a.fslot_ent[s];
            Blk wt;
            pcgf_load_blk<W32>(a, wt, s0, s1, s);
            gather_c(wt, e);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) c[i] = wave_sum_dpp(c[i]);   // (DPP + row swaps: no LDS round trips)
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double tv = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) tv = fma(Vi[k * 6 + j], c[j], tv);
            t[k] = tv;
        }
        auto scatter = [&](const Blk &b, int e) {
            if (a.ent_fixed[e]) return;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double r[6];
                row(b, i, r, true);
                const double v = r[0] * t[0] + r[1] * t[1] + r[2] * t[2] + r[3] * t[3] + r[4] * t[4] + r[5] * t[5];
                atomicAdd(yacc + 6 * e + i, -v);
            }
        };
        if (e0 >= 0) with0([&](const Blk &b) { scatter(b, e0); });
        if (e1 >= 0) with1([&](const Blk &b) { scatter(b, e1); });
        for (int s = s0 + lane + 128; s < s1; s += 64) {
            const int e = a.fslot_ent[s];
            Blk wt;
            pcgf_load_blk<W32>(a, wt, s0, s1, s);
            scatter(wt, e);
        }
    };
    typedef std::integral_constant<int, -1> none_t;
    int f = f_lo + wave;
    auto resident_frame = [&](auto K) {   // the wavefront's frame K: rounds 2 K and 2 K + 1 of the resident blocks, as far as they go
        constexpr int k = decltype(K)::value;
        if constexpr (2 * k < RES) {
            if (f < f_hi) { frame(f, std::integral_constant<int, 2 * k>{}, std::integral_constant<int, (2 * k + 1 < RES) ? 2 * k + 1 : -1>{}); f += NW; }
        }
    };
    resident_frame(std::integral_constant<int, 0>{}); resident_frame(std::integral_constant<int, 1>{});
    resident_frame(std::integral_constant<int, 2>{}); resident_frame(std::integral_constant<int, 3>{});
    for (; f < f_hi; f += NW) frame(f, none_t{}, none_t{});
    PCG_STAMP(st_it, 3 + (wave < 8 ? wave : 7));
    __syncthreads();
    if (wave == 0) PCG_STAMP(st_it, 11);
}

// (the fp32 operator needs 217 registers: two wavefronts per SIMD fit, and the frame pass is latency-bound -- 512 threads per workgroup there)
template <bool W32, int RES = 0, int THX = 0>
__global__ void __launch_bounds__(THX ? THX : (W32 ? PCGF32_THREADS : PCG_THREADS)) k_pcgf(const PcgArgs a, double *__restrict__ yg, double *__restrict__ sg) {
    static_assert(RES >= 0 && RES <= 8, "resident blocks");
    constexpr int TH = THX ? THX : (W32 ? PCGF32_THREADS : PCG_THREADS);
    extern __shared__ __align__(16) double lds[];
    const int n = 6 * a.A, G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *x = lds, *r = lds + n, *p = r + n, *Mi = p + n, *yacc = Mi + 6 * n, *red = yacc + n;
    // coarse space (pcg_lds_bytes): zd [12 A] | Ews [144] | einv [72] | per-wavefront scratch [TH / 64][144]
    double *zd = red + (PCGF32_THREADS / 64) * 27 + 8, *Ews = zd + 12 * a.A, *einv = Ews + 144, *wl = einv + 72;
    // it joins when the previous solve needed many iterations (every workgroup reads the same count, before anybody can overwrite it: that happens behind the last hop)
    const bool co = a.coarse_from >= 0 && a.ent != nullptr && a.iters_out[0] >= a.coarse_from;
    int32_t *counter = a.hop + a.parity * PCG_HOP_WORDS;   // (grid_hop_tree)
    int round = 0;
    if (wg == 0 && tid < 2 * PCG_NY + 1) __hip_atomic_store(a.hop + (1 - a.parity) * PCG_HOP_WORDS + tid * 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    if (co) {
        pcgf_build_zd<TH>(a, zd);
        __syncthreads();
    }

    // ---- set-up, first half: every slot's share of the diagonal blocks and of the right-hand side, by entity in LDS, then ONE atomic flush ----
    double *sacc = Mi;
    if (wave == 0) PCG_STAMP(31, 0);
    pcgf_setup_slots<TH>(a, sacc, wg, G);
    if (wave == 0) PCG_STAMP(31, 1);
    for (int i = tid; i < 27 * a.A; i += TH) {
        const double v = sacc[i];
        if (v != 0.0) atomicAdd(sg + (size_t)(i / 27) * 28 + (i % 27), v);   // (once per solve: not spread over partial tables -- every workgroup would read PCG_NY x 27 A values back: measured +78 us)
    }
    const bool e_new = co && (a.e_refresh != 0 || ld_agent(a.eg + 144) == 0.0);   // (uniform over the grid: the mark is only ever written behind the hop below)
    if (e_new) {   // this workgroup's share of the coarse operator E = Z^T S Z, flushed with the rest
        pcgf_setup_coarse<TH>(a, zd, Ews, wl, wg, G, false);
   // (without the damping's part: that is added at today's mu below, so that kept sums do not carry an old one)
        for (int i = tid; i < 144; i += TH) { const double v = Ews[i]; if (v != 0.0) atomicAdd(a.eg + i, v); }
    }
    if (wave == 0) PCG_STAMP(31, 2);
    if (!grid_hop_tree(counter, round, G, a.flags, wg)) return;
    if (wave == 0) PCG_STAMP(31, 3);
    if (co) {   // E = (kept or new) Z^T S Z + mu Z^T Z, the second term from today's Z (every workgroup for itself: a few hundred LDS additions)
        __syncthreads();
        for (int i = tid; i < 144; i += TH) Ews[i] = 0.0;
        __syncthreads();
        for (int e = tid; e < a.A; e += TH) {
            if (!pcgf_on(zd + 12 * e)) continue;
            double Ze[36];
            pcgf_zfull(zd + 12 * e, Ze);
            const int ge = e >= a.C ? 6 : 0;
#pragma unroll
            for (int aa = 0; aa < 6; aa++)
#pragma unroll
                for (int c = 0; c <= aa; c++) {
                    double v = 0.0;
#pragma unroll
                    for (int i = 0; i < 6; i++) v = fma(Ze[6 * i + aa], Ze[6 * i + c], v);
                    atomicAdd(Ews + 12 * (ge + aa) + ge + c, v);
                }
        }
        __syncthreads();
        pcgf_invert_E(a.eg, einv, true, Ews, a.mu);
    }
    if (e_new && wg == 0 && tid == 0) __hip_atomic_store(a.eg + 144, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // z = M^-1 r: block-Jacobi plus the coarse part Z c, c = blockdiag(E_cc, E_mm)^-1 Z^T r (pcgf_coarse_coef / pcgf_coarse_add)
    double *cvec = wl;   // (the per-wavefront scratch of the set-up is free by now)
    auto coarse_coef = [&](const double *__restrict__ rv, double (&cs)[12]) { pcgf_coarse_coef<TH>(a, zd, einv, cvec, rv, cs); };
    auto coarse_add = [&](int e, int row, const double (&cs)[12]) -> double { return pcgf_coarse_add(a, zd, e, row, cs); };
    // ---- set-up, second half, redundantly in every workgroup (one thread per entity): the diagonal block of S inverted straight into LDS,
    //      the right-hand side; x = 0, r = b.  (Every thread reads its sums into registers before the barrier below lets Mi overwrite sacc.) ----
    // (every thread reads the sums it needs past the L2, not from sacc: Mi may take sacc's place at once -- but only when all threads have flushed it)
    for (int e = tid; e < a.A; e += TH) {
        double out[36], be[6];
        if (a.ent_fixed[e]) {
#pragma unroll
            for (int i = 0; i < 36; i++) out[i] = (i % 7 == 0) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) be[i] = 0.0;
        } else {
            double acc[27];
#pragma unroll
            for (int i = 0; i < 27; i++) acc[i] = ld_agent(sg + (size_t)e * 28 + i);
            double m[6][6];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    const double v = a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j] + (i == j ? a.mu : 0.0) - acc[i * (i + 1) / 2 + j];
                    m[i][j] = v; m[j][i] = v;
                }
            if (!spd6_inverse(m, out) && wg == 0) atomicOr(a.flags, 2);
#pragma unroll
            for (int i = 0; i < 6; i++) be[i] = a.g0[6 * e + i] - acc[21 + i];
        }
#pragma unroll
        for (int i = 0; i < 36; i++) Mi[e * 36 + i] = out[i];
#pragma unroll
        for (int i = 0; i < 6; i++) { r[6 * e + i] = be[i]; x[6 * e + i] = 0.0; }
    }
    __syncthreads();

    if (wave == 0) PCG_STAMP(31, 4);
    // ---- z = Minv r, p = z ----
    double rz = 0.0, bb = 0.0;
    double cs[12];
    if (co) coarse_coef(r, cs);
    {
        double sv[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += TH) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            if (co) z += coarse_add(e, row, cs);
            p[i] = z;
            sv[0] += r[i] * z;
            sv[1] += r[i] * r[i];
        }
        block_sum<2, TH>(sv, red);
        rz = sv[0]; bb = sv[1];
    }
    __syncthreads();

    if (wave == 0) PCG_STAMP(31, 5);
    int it_cg = 0;
    double rr = bb;
    PcgBlk<W32> res[RES > 0 ? RES : 1];
    int eres[RES > 0 ? RES : 1] = {};
    pcgf_load_resident<W32, TH, RES>(a, wg, G, res, eres);   // (after the set-up: its registers are free)
    while (it_cg < a.max_it && (rr > a.eta2 * bb || rz > a.abs2) && bb > 0.0) {
        // y of this iteration: PCG_NYV partial vectors (workgroup wg adds into vector wg % PCG_NYV), summed by every reader.  Measured at config 5, k_pcg us per
        // solve: 1 vector 458, 2: 436, 4: 449, 8: 452, 16: 474 -- the flush's atomics are not what a hop waited for (that was the barrier's one counter)
        double *ygc = yg + (size_t)(it_cg % 3) * PCG_NYV * a.n_pad, *ygn = yg + (size_t)((it_cg + 1) % 3) * PCG_NYV * a.n_pad;
        for (int i = wg * TH + tid; i < PCG_NYV * a.n_pad; i += G * TH) st_agent(ygn + i, 0.0);
        if (wave == 0) PCG_STAMP(it_cg, 0);
        pcgf_operator<W32, TH, RES>(a, p, yacc, red, wg, G, it_cg, res, eres);
        for (int i = tid; i < n; i += TH) {
            const double v = yacc[i];
            if (v != 0.0) atomicAdd(ygc + (size_t)(wg % PCG_NYV) * a.n_pad + i, v);
        }
        if (wave == 0) PCG_STAMP(it_cg, 12);
        if (!grid_hop_tree(counter, round, G, a.flags, wg)) return;
        if (wave == 0) PCG_STAMP(it_cg, 13);
        // ---- vector updates, redundantly in every workgroup (same numbers, same order: same decisions) ----
        double yl[24];
        double pAp = 0.0;
        {
            double sv[1] = {0.0};
            int ny = 0;
            for (int i = tid; i < n; i += TH, ny++) {
                double ys[PCG_NYV];
#pragma unroll
                for (int k = 0; k < PCG_NYV; k++) ys[k] = ld_agent(ygc + (size_t)k * a.n_pad + i);   // (all in flight together; added in a fixed order)
                double ysum = 0.0;
#pragma unroll
                for (int k = 0; k < PCG_NYV; k++) ysum += ys[k];
                double yv = fma(a.mu, p[i], ysum);
                if (a.ent_fixed[i / 6]) yv = p[i];
                if (ny < 24) yl[ny] = yv;
                sv[0] = fma(p[i], yv, sv[0]);
            }
            block_sum<1, TH>(sv, red);
            pAp = sv[0];
        }
        const double alpha = rz / pAp;
        {
            int ny = 0;
            for (int i = tid; i < n; i += TH, ny++) {
                x[i] = fma(alpha, p[i], x[i]);
                r[i] = fma(-alpha, yl[ny < 24 ? ny : 23], r[i]);
            }
        }
        __syncthreads();
        if (co) coarse_coef(r, cs);
        double s2[2] = {0.0, 0.0};
        double zloc[24];
        int nz = 0;
        for (int i = tid; i < n; i += TH, nz++) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            if (co) z += coarse_add(e, row, cs);
            if (nz < 24) zloc[nz] = z;
            s2[0] += r[i] * z;
            s2[1] += r[i] * r[i];
        }
        block_sum<2, TH>(s2, red);
        const double beta = s2[0] / rz;
        rz = s2[0];
        rr = s2[1];
        nz = 0;
        for (int i = tid; i < n; i += TH, nz++) p[i] = fma(beta, p[i], zloc[nz < 24 ? nz : 23]);
        __syncthreads();
        if (wave == 0) PCG_STAMP(it_cg, 14);
        it_cg++;
    }
    if (wg == 0) {
        for (int i = tid; i < n; i += TH) a.x_out[i] = x[i];
        for (int i = n + tid; i < a.n_pad; i += TH) a.x_out[i] = 0.0;
        if (tid == 0) { a.iters_out[0] = it_cg; a.iters_out[1] += it_cg; a.iters_out[2] += 1; }
    }
}

// ------------------------------------------------------------------------------------------------
// The same solver with the frames sharded over ranks (one process per GPU).  Nothing of it is replicated except the n-vector
// updates: every rank holds its own frames' W blocks and its own PARTIAL U, g0 (pass B over its observations); the operator is a
// sum over ranks,  y = mu p + sum_ranks [ U_rank p - sum_{f in rank} W_f t_f ],  so an iteration costs ONE all-reduce of 8 n bytes,
// queued by the host between two launches (a collective cannot be issued from inside a kernel): the persistent launch of k_pcg is
// cut at that point.  Launch k (k_pcgd_iter) = [vector updates from the reduced y of iteration k-1, redundantly in every workgroup;
// converged -> write delta_s and leave] -> frame pass -> hand-over -> entity pass -> hand-over -> this rank's partial y.  The CG
// vectors live in global memory between launches (workgroup 0 writes them back after the second hand-over, when everybody has
// read the old ones).  The set-up (k_pcgd_setup) leaves the rank's partial diagonal blocks and right-hand side for one more
// all-reduce (28 A doubles); launch 0 inverts them.  Every rank computes the same numbers from the same reduced data: all ranks
// stop in the same launch.
struct PcgDistArgs {
    PcgArgs a;
    double *setup_local;      // [A][28] this rank's share: lower triangle of (U_ee - sum_f W (V+mu)^-1 W^T) (21), g0_e - sum_f W h_f (6)
    double *minv;             // [A][36]
    double *state;            // x [n] | r [n] | p [n] | scal [8]: rz, bb, rr, iterations, done
    double *y;                // [n] this rank's partial S p without the mu p term; all-reduced in place between two launches
    int k, last;              // launch number (0: builds the preconditioner and r = b); last: budget exhausted, write delta_s and leave
    double *host;             // mapped host record {done, iterations, -, sequence}; written when publish_seq != 0
    unsigned long long publish_seq;
    // coarse space (fused kernels): this rank's share of E = Z^T S Z rides behind setup_local ([A][28] | [144]: ONE all-reduce); behind minv: E^-1 blocks [72] | the
    // table zd [12 A] (launch 0 writes them for the later launches).  add_mu: this rank adds the damping's Z^T (mu I) Z (rank 0 only: every rank's U is partial)
    int add_mu = 1;
};

__device__ __forceinline__ void pcgd_items_setup(const PcgArgs &a, double *red, int wg, int G, int tid) {
    for (int it = wg; it < a.n_items; it += G) {
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
        for (int pi = a.it_begin[it] + tid; pi < a.it_end[it]; pi += PCG_THREADS) {
            const int4 rec = a.pair_rec[pi];
            const double *Wb = a.W + (size_t)rec.y * 36, *Vi = a.Vinv + (size_t)rec.x * 36, *h = a.hf + (size_t)rec.x * 6;
            double w[36], yv[36];
#pragma unroll
            for (int q = 0; q < 36; q++) w[q] = Wb[q];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) s = fma(w[i * 6 + k], Vi[k * 6 + j], s);
                    yv[i * 6 + j] = s;
                }
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) s = fma(yv[i * 6 + k], w[j * 6 + k], s);
                    acc[i * (i + 1) / 2 + j] += s;
                }
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) s = fma(w[i * 6 + k], h[k], s);
                acc[21 + i] += s;
            }
        }
        block_sum<27>(acc, red);
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 27; i++) st_agent(a.part + (size_t)it * 28 + i, acc[i]);
        }
    }
}

__global__ void __launch_bounds__(PCG_THREADS) k_pcgd_setup(const PcgDistArgs d) {
    __shared__ double red[PCG_NW * 27];
    const PcgArgs &a = d.a;
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    int32_t *counter = a.hop + a.parity * PCG_HOP_WORDS;   // (grid_hop_tree)
    int round = 0;
    if (wg == 0 && tid < 2 * PCG_NY + 1) __hip_atomic_store(a.hop + (1 - a.parity) * PCG_HOP_WORDS + tid * 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pcgd_items_setup(a, red, wg, G, tid);
    if (!grid_hop_tree(counter, round, G, a.flags, wg)) return;
    for (int e = wg * PCG_THREADS + tid; e < a.A; e += G * PCG_THREADS) {
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
        if (!a.ent_fixed[e]) {
            for (int it = a.ent_item_start[e]; it < a.ent_item_start[e + 1]; it++)
#pragma unroll
                for (int i = 0; i < 27; i++) acc[i] += ld_agent(a.part + (size_t)it * 28 + i);
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = 0; j <= i; j++) acc[i * (i + 1) / 2 + j] = a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j] - acc[i * (i + 1) / 2 + j];
                acc[21 + i] = a.g0[6 * e + i] - acc[21 + i];
            }
        }
#pragma unroll
        for (int i = 0; i < 27; i++) d.setup_local[(size_t)e * 28 + i] = acc[i];
        d.setup_local[(size_t)e * 28 + 27] = 0.0;
    }
}

__global__ void __launch_bounds__(PCG_THREADS) k_pcgd_iter(const PcgDistArgs d) {
    extern __shared__ __align__(16) double lds[];
    const PcgArgs &a = d.a;
    const int n = 6 * a.A, G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *x = lds, *r = lds + n, *p = r + n, *Mi = p + n, *red = Mi + 6 * n;
    double *gx = d.state, *gr = gx + n, *gp = gr + n, *gs = gp + n;
    int32_t *counter = a.hop + a.parity * PCG_HOP_WORDS;   // (grid_hop_tree)
    int round = 0;
    if (wg == 0 && tid < 2 * PCG_NY + 1) __hip_atomic_store(a.hop + (1 - a.parity) * PCG_HOP_WORDS + tid * 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto publish = [&](double done_v, double it_v) {
        if (d.publish_seq && wg == 0 && tid == 0) {
            d.host[0] = done_v; d.host[1] = it_v;
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(d.host) + 3, d.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    double rz, bb, rr, itc;
    if (d.k == 0) {   // the preconditioner and r = b from the all-reduced set-up, x = 0, p = z = Minv r
        for (int e = tid; e < a.A; e += PCG_THREADS) {
            double out[36], be[6];
            if (a.ent_fixed[e]) {
#pragma unroll
                for (int i = 0; i < 36; i++) out[i] = (i % 7 == 0) ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < 6; i++) be[i] = 0.0;
            } else {
                double m[6][6];
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        const double v = d.setup_local[(size_t)e * 28 + i * (i + 1) / 2 + j] + (i == j ? a.mu : 0.0);
                        m[i][j] = v; m[j][i] = v;
                    }
                if (!spd6_inverse(m, out) && wg == 0) atomicOr(a.flags, 2);
#pragma unroll
                for (int i = 0; i < 6; i++) be[i] = d.setup_local[(size_t)e * 28 + 21 + i];
            }
#pragma unroll
            for (int i = 0; i < 36; i++) Mi[e * 36 + i] = out[i];
#pragma unroll
            for (int i = 0; i < 6; i++) { r[6 * e + i] = be[i]; x[6 * e + i] = 0.0; }
        }
        __syncthreads();
        double s[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += PCG_THREADS) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            p[i] = z;
            s[0] += r[i] * z;
            s[1] += r[i] * r[i];
        }
        block_sum<2>(s, red);
        rz = s[0]; bb = s[1]; rr = bb; itc = 0.0;
        __syncthreads();
    } else {
        if (gs[4] != 0.0) {   // converged in an earlier launch (the host had queued this one already)
            publish(1.0, gs[3]);
            return;
        }
        for (int i = tid; i < 6 * n; i += PCG_THREADS) Mi[i] = d.minv[i];
        for (int i = tid; i < n; i += PCG_THREADS) { x[i] = gx[i]; r[i] = gr[i]; p[i] = gp[i]; }
        rz = gs[0]; bb = gs[1]; itc = gs[3];
        __syncthreads();
        // y = S p of the iteration before: the rank sum + mu p (identity rows for gauge entities)
        double yl[24];
        double s1[1] = {0.0};
        {
            int ny = 0;
            for (int i = tid; i < n; i += PCG_THREADS, ny++) {
                const double yv = a.ent_fixed[i / 6] ? p[i] : d.y[i] + a.mu * p[i];
                if (ny < 24) yl[ny] = yv;
                s1[0] = fma(p[i], yv, s1[0]);
            }
        }
        block_sum<1>(s1, red);
        const double alpha = rz / s1[0];
        {
            int ny = 0;
            for (int i = tid; i < n; i += PCG_THREADS, ny++) {
                x[i] = fma(alpha, p[i], x[i]);
                r[i] = fma(-alpha, yl[ny < 24 ? ny : 23], r[i]);
            }
        }
        __syncthreads();
        double s2[2] = {0.0, 0.0};
        double zloc[24];
        int nz = 0;
        for (int i = tid; i < n; i += PCG_THREADS, nz++) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            if (nz < 24) zloc[nz] = z;
            s2[0] += r[i] * z;
            s2[1] += r[i] * r[i];
        }
        block_sum<2>(s2, red);
        const double beta = s2[0] / rz;
        rz = s2[0];
        rr = s2[1];
        nz = 0;
        for (int i = tid; i < n; i += PCG_THREADS, nz++) p[i] = fma(beta, p[i], zloc[nz < 24 ? nz : 23]);
        itc += 1.0;
        __syncthreads();
    }
    const bool done = !(itc < (double)a.max_it && (rr > a.eta2 * bb || rz > a.abs2) && bb > 0.0);
    if (done || d.last) {
        if (wg == 0) {
            for (int i = tid; i < n; i += PCG_THREADS) a.x_out[i] = x[i];
            for (int i = n + tid; i < a.n_pad; i += PCG_THREADS) a.x_out[i] = 0.0;
            if (tid == 0) { gs[3] = itc; gs[4] = 1.0; a.iters_out[0] = (int)itc; a.iters_out[1] += (int)itc; a.iters_out[2] += 1; }
        }
        publish(1.0, itc);
        return;
    }
    // ---- frame pass over this rank's frames ----
    for (int f = wg * (PCG_THREADS / 64) + wave; f < a.F; f += G * (PCG_THREADS / 64)) {
        const int s0 = a.fslot_start[f], s1 = a.fslot_start[f + 1];
        double c[6] = {0, 0, 0, 0, 0, 0};
        for (int s = s0 + lane; s < s1; s += 64) {
            const int e = a.fslot_ent[s];
            const double2 *wb = reinterpret_cast<const double2 *>(a.W + (size_t)s * 36);
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double pe = p[6 * e + i];
                const double2 w0 = wb[3 * i], w1 = wb[3 * i + 1], w2 = wb[3 * i + 2];
                c[0] = fma(w0.x, pe, c[0]); c[1] = fma(w0.y, pe, c[1]); c[2] = fma(w1.x, pe, c[2]);
                c[3] = fma(w1.y, pe, c[3]); c[4] = fma(w2.x, pe, c[4]); c[5] = fma(w2.y, pe, c[5]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int i = 0; i < 6; i++) c[i] += __shfl_xor(c[i], off);
        if (lane < 6) {
            double tv = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) tv = fma(a.Vinv[(size_t)f * 36 + lane * 6 + k], c[k], tv);
            st_agent(a.t + (size_t)f * 6 + lane, tv);
        }
    }
    if (!grid_hop_tree(counter, round, G, a.flags, wg)) { publish(1.0, itc); return; }   // (the host stops queueing; device flag 4 says why)
    // ---- entity pass over this rank's incidences, by item; the entity's first item also carries (U_rank p)_e ----
    for (int it = wg; it < a.n_items; it += G) {
        const int e = a.it_ent[it];
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int pi = a.it_begin[it] + tid; pi < a.it_end[it]; pi += PCG_THREADS) {
            const int4 rec = a.pair_rec[pi];
            const double2 *wb = reinterpret_cast<const double2 *>(a.W + (size_t)rec.y * 36);
            double tv[6];
#pragma unroll
            for (int k = 0; k < 6; k++) tv[k] = ld_agent(a.t + (size_t)rec.x * 6 + k);
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double2 w0 = wb[3 * i], w1 = wb[3 * i + 1], w2 = wb[3 * i + 2];
                acc[i] -= w0.x * tv[0] + w0.y * tv[1] + w1.x * tv[2] + w1.y * tv[3] + w2.x * tv[4] + w2.y * tv[5];
            }
        }
        if (it == a.ent_item_start[e] && !a.ent_fixed[e]) {
            for (int bq = tid; bq < a.A; bq += PCG_THREADS) {
                if (a.ent_fixed[bq]) continue;
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        const double u = bq < e ? a.U[(size_t)(6 * e + i) * a.n_pad + 6 * bq + j]
                                                : (bq > e ? a.U[(size_t)(6 * bq + j) * a.n_pad + 6 * e + i]
                                                          : (j <= i ? a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j] : a.U[(size_t)(6 * e + j) * a.n_pad + 6 * e + i]));
                        acc[i] = fma(u, p[6 * bq + j], acc[i]);
                    }
            }
        }
        block_sum<6>(acc, red);
        if (tid < 6) st_agent(a.part + (size_t)it * 28 + tid, acc[tid]);
    }
    if (!grid_hop_tree(counter, round, G, a.flags, wg)) { publish(1.0, itc); return; }
    // ---- this rank's partial y (the items' shares in item order), the CG state back to memory ----
    for (int i = wg * PCG_THREADS + tid; i < n; i += G * PCG_THREADS) {
        const int e = i / 6, row = i - 6 * e;
        const int i0 = a.ent_item_start[e], cnt = a.ent_item_start[e + 1] - i0;
        double sh[PCG_MAX_ITEMS];
#pragma unroll
        for (int k = 0; k < PCG_MAX_ITEMS; k++) sh[k] = k < cnt ? ld_agent(a.part + (size_t)(i0 + k) * 28 + row) : 0.0;
        double yv = 0.0;
#pragma unroll
        for (int k = 0; k < PCG_MAX_ITEMS; k++) yv += sh[k];
        d.y[i] = a.ent_fixed[e] ? 0.0 : yv;
    }
    if (wg == 0) {
        for (int i = tid; i < n; i += PCG_THREADS) { gx[i] = x[i]; gr[i] = r[i]; gp[i] = p[i]; }
        if (d.k == 0) for (int i = tid; i < 6 * n; i += PCG_THREADS) d.minv[i] = Mi[i];
        if (tid == 0) { gs[0] = rz; gs[1] = bb; gs[2] = rr; gs[3] = itc; gs[4] = 0.0; }
    }
    publish(0.0, itc);
}

// ------------------------------------------------------------------------------------------------
// The sharded solver with the one-pass operator of k_pcgf.  Between two launches there is a kernel boundary AND an all-reduce, so NOTHING
// inside a launch has to wait for another workgroup: the set-up scatters this rank's sums straight into its share (atomics, cleared by the
// launcher), launch k scatters this rank's partial y into buffer k % 3, reads the reduced one of launch k - 1 and clears the one of launch k + 1;
// the CG state alternates between two buffers (a late workgroup still reads the old one while workgroup 0 writes the new one).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PCG_THREADS) k_pcgd_setup_f(const PcgDistArgs d) {
    extern __shared__ __align__(16) double lds[];
    const PcgArgs &a = d.a;
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    pcgf_setup_slots(a, lds, wg, G);
    for (int i = tid; i < 27 * a.A; i += PCG_THREADS) {
        const double v = lds[i];
        if (v != 0.0) atomicAdd(d.setup_local + (size_t)(i / 27) * 28 + (i % 27), -v);
    }
    if (a.coarse_from >= 0 && a.ent != nullptr && a.iters_out[0] >= a.coarse_from) {   // this rank's share of the coarse operator (pcgf_setup_coarse), into the same all-reduce
        double *zd = lds + 27 * a.A, *Ews = zd + 12 * a.A, *wl = Ews + 144;
        pcgf_build_zd<PCG_THREADS>(a, zd);
        __syncthreads();
        pcgf_setup_coarse<PCG_THREADS>(a, zd, Ews, wl, wg, G, d.add_mu != 0);
        for (int i = tid; i < 144; i += PCG_THREADS) { const double v = Ews[i]; if (v != 0.0) atomicAdd(d.setup_local + (size_t)28 * a.A + i, v); }
    }
    for (int e = wg * PCG_THREADS + tid; e < a.A; e += G * PCG_THREADS) {   // this rank's U_ee and g0_e (partial sums over its observations)
        if (a.ent_fixed[e]) continue;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) atomicAdd(d.setup_local + (size_t)e * 28 + i * (i + 1) / 2 + j, a.U[(size_t)(6 * e + i) * a.n_pad + 6 * e + j]);
            atomicAdd(d.setup_local + (size_t)e * 28 + 21 + i, a.g0[6 * e + i]);
        }
    }
}

struct PcgDistBufs { const double *state_rd; double *state_wr; const double *y_rd; double *y_wr, *y_zero; };

template <bool W32>   // (fp32 W blocks: kernels.h, Blocks::Wf)
__global__ void __launch_bounds__(PCG_THREADS) k_pcgd_iter_f(const PcgDistArgs d, const PcgDistBufs bf) {
    extern __shared__ __align__(16) double lds[];
    const PcgArgs &a = d.a;
    const int n = 6 * a.A, G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    double *x = lds, *r = lds + n, *p = r + n, *Mi = p + n, *yacc = Mi + 6 * n, *red = yacc + n;
    double *zd = red + (PCGF32_THREADS / 64) * 27 + 8, *einv = zd + 12 * a.A + 144, *cvec = einv + 72;   // (the layout of k_pcgf: pcg_lds_bytes)
    const bool co = a.coarse_from >= 0 && a.ent != nullptr && a.iters_out[0] >= a.coarse_from;   // (the same on every rank: a count the ranks agree on; a solve's launches all see the previous solve's)
    double *g_einv = d.minv + (size_t)36 * a.A, *g_zd = g_einv + 72;
    double cs[12];
    const double *gx = bf.state_rd, *gr = gx + n, *gp = gr + n, *gs = gp + n;
    double *hx = bf.state_wr, *hr = hx + n, *hp = hr + n, *hs = hp + n;
    auto publish = [&](double done_v, double it_v) {
        if (d.publish_seq && wg == 0 && tid == 0) {
            d.host[0] = done_v; d.host[1] = it_v;
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(d.host) + 3, d.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    double rz, bb, rr, itc;
    if (d.k == 0) {   // the preconditioner and r = b from the all-reduced set-up, x = 0, p = z = Minv r
        for (int e = tid; e < a.A; e += PCG_THREADS) {
            double out[36], be[6];
            if (a.ent_fixed[e]) {
#pragma unroll
                for (int i = 0; i < 36; i++) out[i] = (i % 7 == 0) ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < 6; i++) be[i] = 0.0;
            } else {
                double m[6][6];
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        const double v = d.setup_local[(size_t)e * 28 + i * (i + 1) / 2 + j] + (i == j ? a.mu : 0.0);
                        m[i][j] = v; m[j][i] = v;
                    }
                if (!spd6_inverse(m, out) && wg == 0) atomicOr(a.flags, 2);
#pragma unroll
                for (int i = 0; i < 6; i++) be[i] = d.setup_local[(size_t)e * 28 + 21 + i];
            }
#pragma unroll
            for (int i = 0; i < 36; i++) Mi[e * 36 + i] = out[i];
#pragma unroll
            for (int i = 0; i < 6; i++) { r[6 * e + i] = be[i]; x[6 * e + i] = 0.0; }
        }
        if (co) {   // the table and the inverse blocks of the all-reduced E; kept for the later launches of this solve
            pcgf_build_zd<PCG_THREADS>(a, zd);
            pcgf_invert_E(d.setup_local + (size_t)28 * a.A, einv, false);
        }
        __syncthreads();
        if (co && wg == 0) {
            for (int i = tid; i < 12 * a.A; i += PCG_THREADS) g_zd[i] = zd[i];
            if (tid < 72) g_einv[tid] = einv[tid];
        }
        if (co) pcgf_coarse_coef<PCG_THREADS>(a, zd, einv, cvec, r, cs);
        double sv[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += PCG_THREADS) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            if (co) z += pcgf_coarse_add(a, zd, e, row, cs);
            p[i] = z;
            sv[0] += r[i] * z;
            sv[1] += r[i] * r[i];
        }
        block_sum<2>(sv, red);
        rz = sv[0]; bb = sv[1]; rr = bb; itc = 0.0;
        __syncthreads();
    } else {
        if (gs[4] != 0.0) {   // converged in an earlier launch (the host had queued this one already): hand the verdict on to the next one
            if (wg == 0 && tid == 0) { hs[3] = gs[3]; hs[4] = 1.0; }
            publish(1.0, gs[3]);
            return;
        }
        for (int i = tid; i < 6 * n; i += PCG_THREADS) Mi[i] = d.minv[i];
        if (co) {
            for (int i = tid; i < 12 * a.A; i += PCG_THREADS) zd[i] = g_zd[i];
            if (tid < 72) einv[tid] = g_einv[tid];
        }
        for (int i = tid; i < n; i += PCG_THREADS) { x[i] = gx[i]; r[i] = gr[i]; p[i] = gp[i]; }
        rz = gs[0]; bb = gs[1]; itc = gs[3];
        __syncthreads();
        double yl[24];
        double s1[1] = {0.0};
        {
            int ny = 0;
            for (int i = tid; i < n; i += PCG_THREADS, ny++) {
                const double yv = a.ent_fixed[i / 6] ? p[i] : bf.y_rd[i] + a.mu * p[i];
                if (ny < 24) yl[ny] = yv;
                s1[0] = fma(p[i], yv, s1[0]);
            }
        }
        block_sum<1>(s1, red);
        const double alpha = rz / s1[0];
        {
            int ny = 0;
            for (int i = tid; i < n; i += PCG_THREADS, ny++) {
                x[i] = fma(alpha, p[i], x[i]);
                r[i] = fma(-alpha, yl[ny < 24 ? ny : 23], r[i]);
            }
        }
        __syncthreads();
        if (co) pcgf_coarse_coef<PCG_THREADS>(a, zd, einv, cvec, r, cs);
        double s2[2] = {0.0, 0.0};
        double zloc[24];
        int nz = 0;
        for (int i = tid; i < n; i += PCG_THREADS, nz++) {
            const int e = i / 6, row = i - 6 * e;
            double z = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) z = fma(Mi[e * 36 + row * 6 + k], r[6 * e + k], z);
            if (co) z += pcgf_coarse_add(a, zd, e, row, cs);
            if (nz < 24) zloc[nz] = z;
            s2[0] += r[i] * z;
            s2[1] += r[i] * r[i];
        }
        block_sum<2>(s2, red);
        const double beta = s2[0] / rz;
        rz = s2[0];
        rr = s2[1];
        nz = 0;
        for (int i = tid; i < n; i += PCG_THREADS, nz++) p[i] = fma(beta, p[i], zloc[nz < 24 ? nz : 23]);
        itc += 1.0;
        __syncthreads();
    }
    const bool done = !(itc < (double)a.max_it && (rr > a.eta2 * bb || rz > a.abs2) && bb > 0.0);
    if (done || d.last) {
        if (wg == 0) {
            for (int i = tid; i < n; i += PCG_THREADS) a.x_out[i] = x[i];
            for (int i = n + tid; i < a.n_pad; i += PCG_THREADS) a.x_out[i] = 0.0;
            if (tid == 0) { hs[3] = itc; hs[4] = 1.0; a.iters_out[0] = (int)itc; a.iters_out[1] += (int)itc; a.iters_out[2] += 1; }
        }
        publish(1.0, itc);
        return;
    }
    for (int i = wg * PCG_THREADS + tid; i < n; i += G * PCG_THREADS) bf.y_zero[i] = 0.0;   // the buffer of the NEXT launch (last read two launches ago)
    { const PcgBlk<W32> nores[1] = {}; const int noe[1] = {-1}; pcgf_operator<W32, PCG_THREADS, 0>(a, p, yacc, red, wg, G, 0, nores, noe); }
    for (int i = tid; i < n; i += PCG_THREADS) {
        const double v = yacc[i];
        if (v != 0.0) atomicAdd(bf.y_wr + i, v);
    }
    if (wg == 0) {
        for (int i = tid; i < n; i += PCG_THREADS) { hx[i] = x[i]; hr[i] = r[i]; hp[i] = p[i]; }
        if (d.k == 0) for (int i = tid; i < 6 * n; i += PCG_THREADS) d.minv[i] = Mi[i];
        if (tid == 0) { hs[0] = rz; hs[1] = bb; hs[2] = rr; hs[3] = itc; hs[4] = 0.0; }
    }
    publish(0.0, itc);
}

size_t pcg_lds_bytes(int A, bool coarse) { return ((size_t)10 * 6 * A + (PCGF32_THREADS / 64) * 27 + 8 + (coarse ? (size_t)12 * A + 144 + 72 + (PCGF32_THREADS / 64) * 144 : 0)) * sizeof(double); }   // x | r | p | Mi [6 n] | yacc [n] (k_pcgf) | red | coarse space: zd [12 A] | Ews | einv | scratch

// the largest grid of the persistent PCG kernels that is resident as a whole (their hand-overs wait for every workgroup): what the occupancy query
// admits per CU for the kernel with the larger footprint, times the CUs
int pcg_max_grid(int A, int cus) {
    const size_t lds = pcg_lds_bytes(A);
    static size_t g1 = 48 * 1024, g2 = 48 * 1024;
    static size_t g3 = 48 * 1024, g4 = 48 * 1024;
    allow_dynamic_lds(reinterpret_cast<const void *>(k_pcg), lds, g1);
    allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgd_iter), lds, g2);
    allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<false>), lds, g3);
    allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<true>), lds, g4);
    int n1 = 0, n2 = 0, n3 = 0, n4 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n3, k_pcgf<false>, PCG_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); n3 = 1; }
    {
        static size_t g6 = 48 * 1024;
        int n6 = 0;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<false, 2>), lds, g6);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n6, k_pcgf<false, 2>, PCG_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); n6 = 1; }
        n3 = std::min(n3, n6);
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, k_pcgf<true>, PCGF32_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); n4 = 1; }
    {
        static size_t g5 = 48 * 1024;
        int n5 = 0;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<true, 1>), lds, g5);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n5, k_pcgf<true, 1>, PCGF32_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); n5 = 1; }
        n4 = std::min(n4, n5);
    }
    n3 = std::min(n3, n4);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_pcg, PCG_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); n1 = 1; }
    n1 = std::min(n1, n3);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, k_pcgd_iter, PCG_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); n2 = 1; }
    return std::max(1, std::min(n1, n2)) * cus;
}

void launch_pcg(const DeviceProblem &P, int which, double mu, hipStream_t st) {
    const DeviceProblem::Blocks &b = P.blk[which];
    PcgArgs a;
    a.U = b.S; a.g0 = b.g0; a.W = b.W; a.Vinv = b.Vinv; a.hf = b.hf;
    a.fslot_start = P.fslot_start; a.fslot_ent = P.fslot_ent; a.pair_rec = P.pair_rec; a.ent_fixed = P.ent_fixed; a.up_start = P.up_start; a.up_ent = P.up_ent;
    a.it_ent = P.pcg_it_ent; a.it_begin = P.pcg_it_begin; a.it_end = P.pcg_it_end; a.ent_item_start = P.pcg_ent_item_start; a.n_items = P.pcg_n_items;
    a.A = P.A; a.F = P.F; a.n_pad = P.n_pad; a.mu = mu; a.eta2 = P.pcg_eta_now * P.pcg_eta_now; a.abs2 = P.pcg_abs_tol * P.pcg_abs_tol * mu; a.max_it = P.pcg_max_it;
    a.part = P.pcg_ws; a.t = a.part + (size_t)P.pcg_n_items * 28;
    a.counter = P.pcg_counter; a.hop = P.pcg_hop; a.parity = P.pcg_parity & 1; P.pcg_parity++;
    a.x_out = P.delta_s; a.iters_out = P.pcg_counter + 2; a.flags = P.flags;
    const size_t lds = pcg_lds_bytes(P.A, P.pcg_coarse != 0);
    static size_t granted = 48 * 1024, granted_f = 48 * 1024, granted_f32 = 48 * 1024;
    a.Wf = nullptr;
    HookScope _h(P, KID_PCG);
    if (P.pcg_fused && !P.deterministic) {   // one pass over W and one hand-over per iteration; its atomics take the sums in any order
        // the coarse operator's sums are formed every pcg_e_every-th solve of an LM run (and by the first one that needs them: the kernel looks at their mark)
        const bool e_refresh = P.pcg_e_every <= 1 || P.pcg_e_age <= 0 || P.pcg_e_age >= P.pcg_e_every;
        P.pcg_e_age = e_refresh ? 1 : P.pcg_e_age + 1;
        a.e_refresh = e_refresh ? 1 : 0;
        (void)hipMemsetAsync(P.pcg_yg, 0, ((size_t)3 * PCG_NYV * P.n_pad + (size_t)28 * P.A + (e_refresh ? 160 : 0)) * sizeof(double), st);
        a.ent = P.ent[which]; a.C = P.C; a.M = P.M; a.coarse_from = (P.pcg_coarse && P.C <= 64) ? P.pcg_coarse_from : -1;
        a.eg = P.pcg_yg + (size_t)3 * PCG_NYV * P.n_pad + (size_t)28 * P.A;
        // fp32 blocks (kernels.h, Blocks::Wf): allocated -- and written by pass A INSTEAD of the fp64 blocks -- only where the forcing term is far above what
        // that rounding can show (ba_capi.hip, PCG_W32_MIN_ETA; AAR_PCG_W32=0: never): the allocation is the one place that decides
        if (b.Wf) {
            a.Wf = b.Wf;
            static size_t granted_f32r = 48 * 1024;
            if (P.pcg_resident) {
                allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<true, 1>), lds, granted_f32r);
                hipLaunchKernelGGL((k_pcgf<true, 1>), dim3(P.pcg_grid), dim3(PCGF32_THREADS), lds, st, a, P.pcg_yg, P.pcg_yg + (size_t)3 * PCG_NYV * P.n_pad);
                return;
            }
            allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<true>), lds, granted_f32);
            hipLaunchKernelGGL(k_pcgf<true>, dim3(P.pcg_grid), dim3(PCGF32_THREADS), lds, st, a, P.pcg_yg, P.pcg_yg + (size_t)3 * PCG_NYV * P.n_pad);
            return;
        }
        if (P.pcg_resident) {   // fp64 blocks, one wavefront per SIMD (512 registers per lane): both rounds of a wavefront's first frame stay in registers (2 x 18 double2 = 144: 256 + 210 in all)
            static size_t granted_fr = 48 * 1024;
            allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<false, 2>), lds, granted_fr);
            hipLaunchKernelGGL((k_pcgf<false, 2>), dim3(P.pcg_grid), dim3(PCG_THREADS), lds, st, a, P.pcg_yg, P.pcg_yg + (size_t)3 * PCG_NYV * P.n_pad);
            return;
        }
        allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgf<false>), lds, granted_f);
        hipLaunchKernelGGL(k_pcgf<false>, dim3(P.pcg_grid), dim3(PCG_THREADS), lds, st, a, P.pcg_yg, P.pcg_yg + (size_t)3 * PCG_NYV * P.n_pad);
        return;
    }
    if (b.Wf) { (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.flags), 2, 1, st); return; }   // (k_pcg reads the fp64 blocks, which a block set with Wf does not keep: see pcgf_operator)
    allow_dynamic_lds(reinterpret_cast<const void *>(k_pcg), lds, granted);
    hipLaunchKernelGGL(k_pcg, dim3(P.pcg_grid), dim3(PCG_THREADS), lds, st, a);
}

static inline bool pcgd_fused(const DeviceProblem &P) { return P.pcg_fused && !P.deterministic; }
static PcgDistArgs pcgd_args(const DeviceProblem &P, int which, double mu) {
    const DeviceProblem::Blocks &b = P.blk[which];
    PcgDistArgs d;
    PcgArgs &a = d.a;
    a.U = b.S; a.g0 = b.g0; a.W = b.W; a.Vinv = b.Vinv; a.hf = b.hf;
    a.fslot_start = P.fslot_start; a.fslot_ent = P.fslot_ent; a.pair_rec = P.pair_rec; a.ent_fixed = P.ent_fixed; a.up_start = P.up_start; a.up_ent = P.up_ent;
    a.it_ent = P.pcg_it_ent; a.it_begin = P.pcg_it_begin; a.it_end = P.pcg_it_end; a.ent_item_start = P.pcg_ent_item_start; a.n_items = P.pcg_n_items;
    a.A = P.A; a.F = P.F; a.n_pad = P.n_pad; a.mu = mu; a.eta2 = P.pcg_eta_now * P.pcg_eta_now; a.abs2 = P.pcg_abs_tol * P.pcg_abs_tol * mu; a.max_it = P.pcg_max_it;
    a.part = P.pcg_ws; a.t = a.part + (size_t)P.pcg_n_items * 28;
    a.counter = P.pcg_counter; a.hop = P.pcg_hop; a.parity = P.pcg_parity & 1; P.pcg_parity++;
    a.x_out = P.delta_s; a.iters_out = P.pcg_counter + 2; a.flags = P.flags;
    a.Wf = pcgd_fused(P) ? P.blk[which].Wf : nullptr;   // (allocated only where pass A writes it instead of the fp64 blocks)
    if (pcgd_fused(P)) { a.ent = P.ent[which]; a.C = P.C; a.M = P.M; a.coarse_from = (P.pcg_coarse && P.C <= 64) ? P.pcg_coarse_from : -1; }
    d.add_mu = P.pcg_rank0 ? 1 : 0;
    d.setup_local = P.pcgd_setup; d.minv = P.pcgd_minv; d.state = P.pcgd_state; d.y = P.pcgd_y;
    d.k = 0; d.last = 0; d.host = P.pcgd_host; d.publish_seq = 0;
    return d;
}

// stride (doubles) of one of the three rotating y buffers / of one of the two state buffers of the fused sharded path
static inline size_t pcgd_y_stride(const DeviceProblem &P) { return (size_t)6 * P.A + 8; }
static inline size_t pcgd_state_stride(const DeviceProblem &P) { return (size_t)18 * P.A + 8; }
// the buffer the host all-reduces after launch k
double *pcgd_y_of_launch(const DeviceProblem &P, int k) { return pcgd_fused(P) ? P.pcgd_y + (size_t)(k % 3) * pcgd_y_stride(P) : P.pcgd_y; }

void launch_pcgd_setup(const DeviceProblem &P, int which, double mu, hipStream_t st) {
    const PcgDistArgs d = pcgd_args(P, which, mu);
    HookScope _h(P, KID_PCG);
    if (pcgd_fused(P)) {
        (void)hipMemsetAsync(P.pcgd_setup, 0, ((size_t)28 * P.A + 144) * sizeof(double), st);
        (void)hipMemsetAsync(P.pcgd_y, 0, 3 * pcgd_y_stride(P) * sizeof(double), st);
        const size_t lds = ((size_t)27 * P.A + (P.pcg_coarse ? (size_t)12 * P.A + 144 + (PCG_THREADS / 64) * 144 : 0)) * sizeof(double);
        static size_t granted = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgd_setup_f), lds, granted);
        hipLaunchKernelGGL(k_pcgd_setup_f, dim3(P.pcg_grid), dim3(PCG_THREADS), lds, st, d);
        return;
    }
    if (P.blk[which].Wf) { (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.flags), 2, 1, st); return; }   // (reads the fp64 blocks: never on a block set that keeps Wf instead)
    hipLaunchKernelGGL(k_pcgd_setup, dim3(P.pcg_grid), dim3(PCG_THREADS), 0, st, d);
}

void launch_pcgd_iter(const DeviceProblem &P, int which, double mu, int k, bool last, unsigned long long publish_seq, hipStream_t st) {
    PcgDistArgs d = pcgd_args(P, which, mu);
    d.k = k; d.last = last ? 1 : 0; d.publish_seq = publish_seq;
    const size_t lds = pcg_lds_bytes(P.A, pcgd_fused(P) && P.pcg_coarse != 0);
    static size_t granted = 48 * 1024, granted_f = 48 * 1024;
    HookScope _h(P, KID_PCG);
    if (pcgd_fused(P)) {
        PcgDistBufs bf;
        const size_t ys = pcgd_y_stride(P), ss = pcgd_state_stride(P);
        bf.state_rd = P.pcgd_state + (size_t)(k % 2) * ss; bf.state_wr = P.pcgd_state + (size_t)((k + 1) % 2) * ss;
        bf.y_rd = P.pcgd_y + (size_t)((k + 2) % 3) * ys; bf.y_wr = P.pcgd_y + (size_t)(k % 3) * ys; bf.y_zero = P.pcgd_y + (size_t)((k + 1) % 3) * ys;
        if (d.a.Wf) {
            static size_t granted_f32 = 48 * 1024;
            allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgd_iter_f<true>), lds, granted_f32);
            hipLaunchKernelGGL(k_pcgd_iter_f<true>, dim3(P.pcg_grid), dim3(PCG_THREADS), lds, st, d, bf);
            return;
        }
        allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgd_iter_f<false>), lds, granted_f);
        hipLaunchKernelGGL(k_pcgd_iter_f<false>, dim3(P.pcg_grid), dim3(PCG_THREADS), lds, st, d, bf);
        return;
    }
    allow_dynamic_lds(reinterpret_cast<const void *>(k_pcgd_iter), lds, granted);
    if (P.blk[which].Wf) { (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.flags), 2, 1, st); return; }
    hipLaunchKernelGGL(k_pcgd_iter, dim3(P.pcg_grid), dim3(PCG_THREADS), lds, st, d);
}

}  // namespace aar

#ifdef AAR_PCG_STAMPS
extern "C" int aar_debug_pcg_stamps(unsigned long long *out) {   // [3][32][16]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(aar::g_pcg_st), sizeof(unsigned long long) * 3 * 32 * 16) == hipSuccess ? 0 : -1;
}
#endif
