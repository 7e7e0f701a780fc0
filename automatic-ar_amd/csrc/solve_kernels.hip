// HIP kernels (gfx950, wave64) for the damped normal-equation solve of one LM try.  The reference
// factors the whole sparse (J^T J + mu I) with Eigen::SimplicialLDLT (libs/sparselevmarq.h:384-400); the
// same elimination is done here in block form: per-frame 6x6 inverses (k_frame_inv), Schur complement
// onto the cameras+markers (k_schur), dense blocked LDL^T of the reduced system (k_ldl_*, which also applies
// the damping and the gauge rows on first touch), and
// back-substitution of the frame poses (k_backsub).  mu is added to EVERY diagonal entry, as in the
// reference (:387-392).  fp64 throughout.
#include <type_traits>
#include "geom.hpp"
#include "kernels.h"
#include "backsub.hpp"

namespace aar {

typedef double dg_acc_t __attribute__((ext_vector_type(4)));   // one lane's share of a 16x16 fp64 MFMA accumulator tile

// ------------------------------------------------------------------------------------------------
// (V_f + mu I)^-1 and h_f = (V_f + mu I)^-1 g_f, one thread per frame.  Only launched when pass A's prediction of the
// damping was wrong (first step, a rejected try, gain < 0.94): a mu retry needs no Jacobian pass.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_frame_inv(const double *__restrict__ V, const double *__restrict__ gf, int F,
                                                   double mu, int frames_fixed, double *__restrict__ Vinv,
                                                   double *__restrict__ hf, int32_t *__restrict__ flags,
                                                   const double *__restrict__ mu_dev = nullptr, double mu_scale = 0.0) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    if (mu_dev) mu = mu_scale * mu_dev[0];   // the first step's damping tau * max diag(J^T J), still on its way to the host (aar_lm_init)
    double out[36];
    if (frames_fixed) {
#pragma unroll
        for (int i = 0; i < 36; i++) out[i] = 0.0;
    } else {
        double a[6][6];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) a[i][j] = V[(size_t)f * 36 + i * 6 + j] + (i == j ? mu : 0.0);
        if (!spd6_inverse(a, out)) atomicOr(flags, 1);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double hv = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            Vinv[(size_t)f * 36 + i * 6 + j] = out[i * 6 + j];
            hv += out[i * 6 + j] * gf[(size_t)f * 6 + j];
        }
        hf[(size_t)f * 6 + i] = hv;
    }
}

__global__ void k_publish(const double *__restrict__ scal, const int32_t *__restrict__ flags, double *__restrict__ host,
                          unsigned long long seq, int flags_reduced) {
    if (threadIdx.x == 0) publish_host(scal, flags, host, seq, flags_reduced);
}


struct ReduceArgs {   // the step's scalars: [sum r^2, sum |delta_f|^2, sum delta.g] -> scal -> (publish_seq != 0) mapped host record
    const double *err_part; int n_err; const double *lin_part; int F; int fold_shared;
    double *scal; const int32_t *flags; double *host; unsigned long long publish_seq;
    // md_U != nullptr (aar_lm_init): scal[4] = max over the free diagonal of J^T J as well (mu_0 = tau * max, libs/sparselevmarq.h:369-377):
    // k_maxdiag's sum in the same launch
    const double *md_U, *md_V; const int32_t *md_fixed; int md_n_pad, md_A, md_frames_fixed;
    const int32_t *cg_iters;   // solver spcg: [0] = CG iterations of the step's solve -> scal[7], so that the host sees them with the step's scalars (nullptr: scal[7] = -1)
};

__device__ __forceinline__ void reduce_scalars_body(const ReduceArgs &r) {   // the first 256 threads of the workgroup, fixed summation order
    __shared__ double red[3][256];
    const int tid = threadIdx.x < 256 ? threadIdx.x : 255;
    const bool act = threadIdx.x < 256;
    double e = 0.0, d2 = 0.0, dg = 0.0;
    if (act) {
    // the loads of eight rounds are in flight together (a thread walks 20 rounds at 5 000 frames: one round trip each would be
    // 20 us of a launch that rides nowhere at that size); the additions keep their order
    for (int i0 = tid; i0 < r.n_err; i0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (i0 + 256 * u < r.n_err) ? r.err_part[i0 + 256 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) e += v[u];
    }
    for (int i0 = tid; i0 < r.F; i0 += 8 * 256) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (i0 + 256 * u < r.F) ? reinterpret_cast<const double2 *>(r.lin_part)[i0 + 256 * u] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 8; u++) { d2 += v[u].x; dg += v[u].y; }
    }
    red[0][tid] = e; red[1][tid] = d2; red[2][tid] = dg;
    }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (act && tid < off) {
            red[0][tid] += red[0][tid + off];
            red[1][tid] += red[1][tid + off];
            red[2][tid] += red[2][tid + off];
        }
        __syncthreads();
    }
    const double sum_e = red[0][0], sum_d2 = red[1][0], sum_dg = red[2][0];
    if (r.md_U) {   // uniform per launch
        double m = -1.7976931348623157e308;
        if (act) {
            for (int i = tid; i < 6 * r.md_A; i += 256)
                if (!r.md_fixed[i / 6]) m = fmax(m, r.md_U[(size_t)i * r.md_n_pad + i]);
            if (!r.md_frames_fixed)
                for (int i0 = tid; i0 < 6 * r.F; i0 += 8 * 256) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int i = i0 + u * 256; v[u] = i < 6 * r.F ? r.md_V[(size_t)(i / 6) * 36 + (i % 6) * 7] : m; }
#pragma unroll
                    for (int u = 0; u < 8; u++) m = fmax(m, v[u]);
                }
        }
        __syncthreads();   // (every thread holds the three sums in registers by now: row 1 of the scratch is free)
        if (act) red[1][tid] = m;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (act && tid < off) red[1][tid] = fmax(red[1][tid], red[1][tid + off]);
            __syncthreads();
        }
    }
    if (act && tid == 0) {
        if (r.md_U) r.scal[4] = red[1][0];
        r.scal[0] = sum_e; r.scal[1] = sum_d2;
        // multi-GPU: delta_s . g0 uses this rank's piece of the shared gradient, so it joins the rank sum
        r.scal[2] = sum_dg + (r.fold_shared ? r.lin_part[2 * (size_t)r.F + 1] : 0.0);
        r.scal[3] = encode_flags(r.flags[0] | r.flags[1] | r.flags[2] | r.flags[3]);   // multi-GPU: joins the rank sum
        r.scal[5] = r.lin_part[2 * (size_t)r.F]; r.scal[6] = r.lin_part[2 * (size_t)r.F + 1];
        r.scal[7] = r.cg_iters ? (double)r.cg_iters[0] : -1.0;
        if (r.publish_seq) publish_host(r.scal, r.flags, r.host, r.publish_seq);
    }
}

__global__ void __launch_bounds__(256) k_reduce_scalars(const ReduceArgs r) { reduce_scalars_body(r); }

// ------------------------------------------------------------------------------------------------
// Schur complement, output-stationary: work item = (shared entity a, a range of the frames that see a).
// The workgroup keeps row panel [S(a, b)]_{b <= a} (6 x 6(a+1)) and the rhs rows of a in LDS, its four
// wavefronts walk the (a, f) pairs: Y = W_af (V_f+mu I)^-1, then for every entity b <= a seen in f:
// panel(b) += Y W_bf^T.  One pass of fp64 atomics subtracts the panel from S at the end; all per-frame
// traffic stays in L2/LDS.  Only the lower triangle of S is produced.
// LDS: [A*36] panel | [8] rhs rows | [4][48] per-wave Y scratch
// ------------------------------------------------------------------------------------------------
// PRE = passes of the frame's slot list (10 slots each) whose W rows are fetched up front, before Y is formed: pays at
// config 5 (HBM-bound, -9 %), costs registers / occupancy at configs 3-4, where 0 is used
// NW = wavefronts per workgroup.  The deterministic mode runs ONE (every LDS addition of the panel then comes in program order:
// no two lanes of an instruction share an address, see the slot loop) and leaves the panel as the work item's record in `part`
// (at part_off[w]: [(a+1)*36] panel | 6 rhs entries) for k_schur_reduce instead of the atomic flush.
template <int PRE, int NW = 4>
__global__ void __launch_bounds__(64 * NW) k_schur(const int32_t *__restrict__ sw_ent, const int32_t *__restrict__ sw_begin,
                                               const int32_t *__restrict__ sw_end, const int4 *__restrict__ pair_rec,
                                               const int32_t *__restrict__ fslot_ent, const double *__restrict__ W,
                                               const double *__restrict__ Vinv, const double *__restrict__ hf, int A,
                                               int n_pad, double sign, double *__restrict__ S, double *__restrict__ rhs,
                                               int rider, const ReduceArgs red, double *__restrict__ part = nullptr,
                                               const int64_t *__restrict__ part_off = nullptr) {
    if (rider && blockIdx.x == 0) {   // rider (dispatched first): the step's scalars go to the host from here, beside the Schur workgroups
        reduce_scalars_body(red);
        return;
    }
    extern __shared__ __align__(16) double lds[];
    double *panel = lds;
    double *pg = lds + (size_t)A * 36;
    double *ysc = pg + 8;
    const int w = (int)blockIdx.x - rider, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: pair records in SGPRs
    const int a = sw_ent[w], pb = sw_begin[w], pe = sw_end[w];
    for (int i = tid; i < (a + 1) * 36; i += 64 * NW) panel[i] = 0.0;
    if (tid < 8) pg[tid] = 0.0;
    __syncthreads();
    double *ys = ysc + wave * 48;
    if constexpr (PRE == 0) {
        for (int p = pb + wave; p < pe; p += NW) {
            const int4 rec = pair_rec[p];
            const int f = rec.x, sg = rec.y, s0 = rec.z, sa = sg - s0;
            const double *Wa = W + (size_t)sg * 36;
            if (lane < 36) {
                const int i = lane / 6, j = lane % 6;
                double y = 0.0;
    #pragma unroll
                for (int k = 0; k < 6; k++) y += Wa[i * 6 + k] * Vinv[(size_t)f * 36 + k * 6 + j];
                ys[lane] = y;
            } else if (lane < 42) {
                const int i = lane - 36;
                double y = 0.0;
    #pragma unroll
                for (int k = 0; k < 6; k++) y += Wa[i * 6 + k] * hf[(size_t)f * 6 + k];
                atomicAdd(pg + i, y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double Y[36];
    #pragma unroll
            for (int i = 0; i < 18; i++) { const double2 v2 = reinterpret_cast<const double2 *>(ys)[i]; Y[2 * i] = v2.x; Y[2 * i + 1] = v2.y; }   // 16-byte reads: half the LDS instructions
            __builtin_amdgcn_wave_barrier();
            const int sl = lane / 6, j = lane % 6;
            if (lane < 60) {
                for (int sb = sl; sb <= sa; sb += 10) {
                    const int b = fslot_ent[s0 + sb];
                    const double2 *wr = reinterpret_cast<const double2 *>(W + (size_t)(s0 + sb) * 36 + j * 6);
                    const double2 w0 = wr[0], w1 = wr[1], w2 = wr[2];
                    double *dst = panel + b * 36 + j;
    #pragma unroll
                    for (int i = 0; i < 6; i++) {
                        const double v = Y[i * 6] * w0.x + Y[i * 6 + 1] * w0.y + Y[i * 6 + 2] * w1.x + Y[i * 6 + 3] * w1.y +
                                         Y[i * 6 + 4] * w2.x + Y[i * 6 + 5] * w2.y;
                        atomicAdd(dst + i * 6, v);
                    }
                }
            }
        }
    } else {
        // Every global load of a pair is issued before the first use (the chain pair -> frame -> slot -> W row used to be four
        // dependent round trips per pair), and the next pair's record is fetched one pair ahead.
        const int sl = lane / 6, j = lane % 6;
        int4 rec = make_int4(0, 0, 0, 0);
        if (pb + wave < pe) rec = pair_rec[pb + wave];
        for (int p = pb + wave; p < pe; p += NW) {
            const int f = rec.x, sg = rec.y, s0 = rec.z, sa = sg - s0;
            if (p + NW < pe) rec = pair_rec[p + NW];
            const double *Wa = W + (size_t)sg * 36;
            double wa[6], vv[6];
            if (lane < 36) {
                const int i = lane / 6, jj = lane % 6;
    #pragma unroll
                for (int k = 0; k < 6; k++) { wa[k] = Wa[i * 6 + k]; vv[k] = Vinv[(size_t)f * 36 + k * 6 + jj]; }
            } else if (lane < 42) {
                const int i = lane - 36;
    #pragma unroll
                for (int k = 0; k < 6; k++) { wa[k] = Wa[i * 6 + k]; vv[k] = hf[(size_t)f * 6 + k]; }
            }
            int bm[PRE > 0 ? PRE : 1];
            double2 wr[PRE > 0 ? PRE : 1][3];
    #pragma unroll
            for (int m = 0; m < PRE; m++) {
                const int sb = sl + 10 * m;
                bm[m] = -1;
                if (lane < 60 && sb <= sa) {
                    bm[m] = fslot_ent[s0 + sb];
                    const double2 *q = reinterpret_cast<const double2 *>(W + (size_t)(s0 + sb) * 36 + j * 6);
                    wr[m][0] = q[0]; wr[m][1] = q[1]; wr[m][2] = q[2];
                }
            }
            if (lane < 42) {
                double y = 0.0;
    #pragma unroll
                for (int k = 0; k < 6; k++) y += wa[k] * vv[k];
                if (lane < 36) ys[lane] = y;
                else atomicAdd(pg + (lane - 36), y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double Y[36];
    #pragma unroll
            for (int i = 0; i < 18; i++) { const double2 v2 = reinterpret_cast<const double2 *>(ys)[i]; Y[2 * i] = v2.x; Y[2 * i + 1] = v2.y; }   // 16-byte reads: half the LDS instructions
            __builtin_amdgcn_wave_barrier();
            auto accumulate = [&](int b, const double2 &w0, const double2 &w1, const double2 &w2) {
                double *dst = panel + b * 36 + j;
    #pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double v = Y[i * 6] * w0.x + Y[i * 6 + 1] * w0.y + Y[i * 6 + 2] * w1.x + Y[i * 6 + 3] * w1.y +
                                     Y[i * 6 + 4] * w2.x + Y[i * 6 + 5] * w2.y;
                    atomicAdd(dst + i * 6, v);
                }
            };
    #pragma unroll
            for (int m = 0; m < PRE; m++)
                if (bm[m] >= 0) accumulate(bm[m], wr[m][0], wr[m][1], wr[m][2]);
            if (lane < 60) {
                for (int sb = sl + 10 * PRE; sb <= sa; sb += 10) {
                    const int b = fslot_ent[s0 + sb];
                    const double2 *q = reinterpret_cast<const double2 *>(W + (size_t)(s0 + sb) * 36 + j * 6);
                    accumulate(b, q[0], q[1], q[2]);
                }
            }
        }
    }
    __syncthreads();
    if (part) {   // deterministic mode: the record of this work item
        double *rec = part + part_off[w];
        for (int idx = tid; idx < (a + 1) * 36; idx += 64 * NW) rec[idx] = panel[idx];
        if (tid < 6) rec[(a + 1) * 36 + tid] = pg[tid];
        return;
    }
    for (int idx = tid; idx < (a + 1) * 36; idx += 64 * NW) {
        const double v = panel[idx];
        if (v != 0.0) {
            const int b = idx / 36, e = idx - b * 36, i = e / 6, j = e - i * 6;
            atomicAdd(S + (size_t)(6 * a + i) * n_pad + 6 * b + j, -sign * v);
        }
    }
    if (tid < 6) atomicAdd(rhs + 6 * a + tid, -sign * pg[tid]);
}

// Deterministic mode, second half of the Schur complement: workgroup = entity a; every value of its row panel (and of its six rhs
// entries) is the sum of the records of a's work items, added in ascending frame order, then S(a, .) -= sign * sum -- the only
// writer of that row, so a plain read-modify-write.
__global__ void __launch_bounds__(256) k_schur_reduce(const int32_t *__restrict__ se_start, const int32_t *__restrict__ se_items,
                                                      const int64_t *__restrict__ part_off, const double *__restrict__ part, int n_pad,
                                                      double sign, double *__restrict__ S, double *__restrict__ rhs) {
    const int a = blockIdx.x, i0 = se_start[a], i1 = se_start[a + 1], np = (a + 1) * 36;
    for (int idx = threadIdx.x; idx < np + 6; idx += 256) {
        double s = 0.0;
        int k = i0;
        for (; k + 8 <= i1; k += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = part[part_off[se_items[k + u]] + idx];
#pragma unroll
            for (int u = 0; u < 8; u++) s += t[u];
        }
        for (; k < i1; k++) s += part[part_off[se_items[k]] + idx];
        if (idx < np) {
            const int b = idx / 36, e = idx - b * 36, i = e / 6, j = e - i * 6;
            S[(size_t)(6 * a + i) * n_pad + 6 * b + j] -= sign * s;
        } else {
            rhs[6 * a + (idx - np)] -= sign * s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Schur complement for many shared entities, on the matrix pipes.  The output-stationary kernel above reads every W_bf once
// per (a, f) pair it meets -- at 216 entities that is 36x the bytes of W.  Here the frame blocks are first laid out DENSE
// per frame (k_schur_fill: Wd / Yd [F][Ad][6][6], absent pairs stay zero -- at config 5 a frame sees 122 of the 166 entities
// that are seen at all, so dense panels cost 36 % more bytes than the sparse lists and make every fetch a contiguous,
// index-free stream), and a workgroup owns a 96 x 192 block of S (16 x 32 dense entities) in fp64 MFMA accumulators
// (8 wavefronts x 9 sub-tiles) while it streams a range of frames through LDS two at a time: K = 2 frames x 6 = 12 = three
// v_mfma_f64_16x16x4 per sub-tile, no padding.  Dense entity 0 is the frame's gradient g_f, so the Schur part of the
// right-hand side is simply column (entity 0, parameter 0) of the product.
// LDS (dynamic): stage [2][288][14]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_schur_fill(const int32_t *__restrict__ slot_frame, const int32_t *__restrict__ slot_dense,
                                                    const double *__restrict__ W, const double *__restrict__ Vinv,
                                                    const double *__restrict__ gf, int total_slots, int F, int Ad,
                                                    double *__restrict__ Wd, double *__restrict__ Yd, int rider, const ReduceArgs red) {
    if (rider && blockIdx.x == 0) {   // rider (dispatched first): the step's scalars go to the host from here, as in k_schur
        reduce_scalars_body(red);
        return;
    }
    const int64_t gid = (int64_t)((int)blockIdx.x - rider) * 256 + threadIdx.x;
    if (gid >= (int64_t)total_slots * 6) {   // the pseudo entity: row 0 of block (f, 0) = g_f
        const int64_t q = gid - (int64_t)total_slots * 6;
        if (q < (int64_t)F * 6) Wd[(q / 6) * Ad * 36 + (q % 6)] = gf[q];
        return;
    }
    const int64_t sl = gid / 6;
    const int i = (int)(gid - sl * 6);
    const int f = slot_frame[sl];
    const double2 *wp = reinterpret_cast<const double2 *>(W + sl * 36 + i * 6);
    const double2 w0 = wp[0], w1 = wp[1], w2 = wp[2];
    const double w[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
    double y[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double2 *vp = reinterpret_cast<const double2 *>(Vinv + (size_t)f * 36 + k * 6);
        const double2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
        y[0] = fma(w[k], v0.x, y[0]); y[1] = fma(w[k], v0.y, y[1]); y[2] = fma(w[k], v1.x, y[2]);
        y[3] = fma(w[k], v1.y, y[3]); y[4] = fma(w[k], v2.x, y[4]); y[5] = fma(w[k], v2.y, y[5]);
    }
    const size_t o = ((size_t)f * Ad + slot_dense[sl]) * 36 + i * 6;
    double2 *yp = reinterpret_cast<double2 *>(Yd + o), *wd = reinterpret_cast<double2 *>(Wd + o);
    yp[0] = make_double2(y[0], y[1]); yp[1] = make_double2(y[2], y[3]); yp[2] = make_double2(y[4], y[5]);
    wd[0] = w0; wd[1] = w1; wd[2] = w2;
}

// SM_FPS frames per step of the K loop (K = 6 SM_FPS per barrier), SM_DEPTH steps of row fetches in flight in registers
#ifndef AAR_SM_FPS
#define AAR_SM_FPS 4
#endif
constexpr int SM_GA = 16, SM_GB = 32, SM_AR = 6 * SM_GA, SM_BR = 6 * SM_GB, SM_ROWS = SM_AR + SM_BR, SM_FPS = AAR_SM_FPS, SM_PS = 6 * SM_FPS + 2,
              SM_DEPTH = 8 / SM_FPS, SM_KT = 6 * SM_FPS / 4;
static_assert((6 * SM_FPS) % 4 == 0 && SM_DEPTH >= 1, "whole MFMA k-steps per barrier");
__global__ void __launch_bounds__(512) k_schur_mfma(const int32_t *__restrict__ w_ga, const int32_t *__restrict__ w_gb,
                                                    const int32_t *__restrict__ w_fb, const int32_t *__restrict__ w_fe,
                                                    const int32_t *__restrict__ flist, const int32_t *__restrict__ dense_ent, const double *__restrict__ Wd,
                                                    const double *__restrict__ Yd, int Ad, int n_pad, double sign,
                                                    double *__restrict__ S, double *__restrict__ rhs, int rider, const ReduceArgs red) {
    extern __shared__ __align__(16) double stage[];   // [2][SM_ROWS][SM_PS]
    if (rider && blockIdx.x == 0) {   // the step's scalars ride here when no k_schur_fill launch precedes (pass A wrote the panels)
        reduce_scalars_body(red);
        return;
    }
    const int w = (int)blockIdx.x - rider, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ga = w_ga[w], gb = w_gb[w], fb = w_fb[w], fe = w_fe[w];
    const int lr = lane >> 4, lc = lane & 15;
    // staging: the SM_ROWS rows x SM_FPS frames of a step travel as 16-byte pieces, SM_PPT per thread over ALL wavefronts (piece q = tid + 512 u:
    // frame q / 864 of the step, then piece (row, part) of that frame's 288 rows; rows 0-95 are the a side (Y), the rest the b side (W): within
    // a side consecutive pieces are consecutive in memory) -- every wavefront then carries the same staging work to the barrier
    constexpr int SM_PIECES = SM_ROWS * 3 * SM_FPS, SM_PPT = (SM_PIECES + 511) / 512;
    int p_lds[SM_PPT], p_off[SM_PPT];
    unsigned sideA = 0;   // bit u: piece u is a row of Y
    auto piece_ff = [&](int u) -> int { const int q = tid + 512 * u; return q >= SM_PIECES ? -1 : q / (SM_ROWS * 3); };
#pragma unroll
    for (int u = 0; u < SM_PPT; u++) {
        const int q = tid + 512 * u, ff = q / (SM_ROWS * 3), r3 = q - ff * (SM_ROWS * 3), row = r3 / 3, part = r3 - 3 * row;
        p_lds[u] = row * SM_PS + 6 * ff + 2 * part;
        p_off[u] = row < SM_AR ? SM_GA * ga * 36 + 2 * r3 : SM_GB * gb * 36 + 2 * (r3 - 3 * SM_AR);
        sideA |= (row < SM_AR ? 1u : 0u) << u;
    }
    const size_t fstride = (size_t)Ad * 36;
    dg_acc_t acc[3][3];
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int y = 0; y < 3; y++) acc[x][y] = dg_acc_t{0.0, 0.0, 0.0, 0.0};
    const int rt0 = 3 * (wave & 1), ct0 = 3 * (wave >> 1);
    // Only the lower triangle (in dense order) is ever stored: a wavefront whose whole 48 x 48 piece lies above the diagonal -- in a block
    // on the diagonal of S up to five of the eight (the a-side group is one half of the b-side group) -- issues no MFMAs at all (the
    // matrix pipes bound this kernel); it still stages rows and keeps the barriers.  (Per-sub-tile masks cost 96 spilled registers.)
    const bool all_dead = __builtin_amdgcn_readfirstlane((int)(SM_AR * ga + 16 * (rt0 + 3) - 1 < SM_BR * gb + 16 * ct0)) != 0;

    // position k of the block's frame list (frames in which both entity groups are present)
    auto fetch_rows = [&](int k0, double2 (&v)[SM_PPT]) {   // nothing here waits: the values are stored steps later
        int fr[SM_FPS];   // the step's frames: wave-uniform, through the scalar cache (a per-lane flist[k0 + ff] would put a dependent vector load in front of every piece)
#pragma unroll
        for (int i = 0; i < SM_FPS; i++) fr[i] = k0 + i < fe ? flist[k0 + i] : -1;
#pragma unroll
        for (int u = 0; u < SM_PPT; u++) {
            const int ff = piece_ff(u);
            int frame = -1;
#pragma unroll
            for (int i = 0; i < SM_FPS; i++) frame = ff == i ? fr[i] : frame;
            if (frame >= 0) v[u] = *reinterpret_cast<const double2 *>(((sideA >> u) & 1 ? Yd : Wd) + (size_t)frame * fstride + p_off[u]);
            else v[u] = make_double2(0.0, 0.0);   // (loading unconditionally and zeroing at the store: 1 050 us against 996)
        }
    };
    auto put_rows = [&](int buf, const double2 (&v)[SM_PPT]) {
#pragma unroll
        for (int u = 0; u < SM_PPT; u++)
            if (piece_ff(u) >= 0) *reinterpret_cast<double2 *>(stage + (size_t)buf * SM_ROWS * SM_PS + p_lds[u]) = v[u];
    };
    // Register ring SM_DEPTH steps deep: the rows of step s + SM_DEPTH are requested while step s computes -- one two-frame step of the
    // matrix pipes (27 MFMAs per wavefront, ~1.4 us) is shorter than a loaded chip's memory latency.  (Three steps deep at four frames per
    // step: 1 025 us against 1 000.)
    double2 v[SM_DEPTH][SM_PPT];
    auto compute = [&](int buf) {
        if (all_dead) return;
        const double *sb = stage + (size_t)buf * SM_ROWS * SM_PS;
        double av[3][SM_KT], bv[3][SM_KT];
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int t = 0; t < SM_KT; t++) {
                av[x][t] = sb[(16 * (rt0 + x) + lc) * SM_PS + 4 * t + lr];            // A[i = lc][k = lr]: Y rows
                bv[x][t] = sb[(SM_AR + 16 * (ct0 + x) + lc) * SM_PS + 4 * t + lr];    // B[k = lr][j = lc]: W rows
            }
#pragma unroll
        for (int t = 0; t < SM_KT; t++)
#pragma unroll
            for (int x = 0; x < 3; x++)
#pragma unroll
                for (int y = 0; y < 3; y++) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x][t], bv[y][t], acc[x][y], 0, 0, 0);
    };
    const int nsteps = (fe - fb + SM_FPS - 1) / SM_FPS;
#pragma unroll
    for (int d = 0; d < SM_DEPTH; d++) fetch_rows(fb + SM_FPS * d, v[d]);     // steps 0 .. SM_DEPTH-1 (beyond the range: zeros, nothing loaded)
    put_rows(0, v[0]);
    fetch_rows(fb + SM_FPS * SM_DEPTH, v[0]);
    __syncthreads();
    for (int s0 = 0; s0 < nsteps; s0 += SM_DEPTH) {
#pragma unroll
        for (int d = 0; d < SM_DEPTH; d++) {
            const int s = s0 + d;                       // this step's rows sit in stage[s & 1]; step s + 1's in ring slot (d + 1) % SM_DEPTH
            if (s < nsteps) {
                compute(s & 1);
                put_rows((s + 1) & 1, v[(d + 1) % SM_DEPTH]);
                fetch_rows(fb + SM_FPS * (s + 1 + SM_DEPTH), v[(d + 1) % SM_DEPTH]);
                __syncthreads();
            }
        }
    }
    // S(a rows, b cols) -= sign * acc, lower triangle only; exact zeros (entity pairs never seen together) are skipped.  Column
    // (dense entity 0, parameter 0) is Y g: the Schur part of the right-hand side.
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int y = 0; y < 3; y++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double val = acc[x][y][r];
                if (val == 0.0) continue;
                const int rd = SM_AR * ga + 16 * (rt0 + x) + lr + 4 * r, cd = SM_BR * gb + 16 * (ct0 + y) + lc;
                const int da = rd / 6, db = cd / 6, pa = rd - 6 * da, pb = cd - 6 * db;
                if (db > da || (db == da && pb > pa)) continue;      // the lower triangle in DENSE order: every entity pair exactly once
                const int ea = dense_ent[da], eb = dense_ent[db];
                if (ea < 0) continue;
                if (db == 0) { if (pb == 0) atomicAdd(rhs + 6 * ea + pa, -sign * val); continue; }
                // dense order is by frequency, not by entity: a pair whose real order is the other way round lands transposed
                const int row = 6 * ea + pa, col = 6 * eb + pb;
                atomicAdd(S + (row >= col ? (size_t)row * n_pad + col : (size_t)col * n_pad + row), -sign * val);
            }
}

// ------------------------------------------------------------------------------------------------
// Dense LDL^T of the reduced system, right-looking, tile NB = 96, lower triangle row-major.
//
// Damping and gauge are applied on the FIRST touch of every element (step 0 of the factorisation), not by a kernel of
// their own: S(i,j) -> S(i,j) + mu [i == j] on free rows; rows / columns of fixed entities (root camera, root marker,
// non-optimised groups) and of the padding -> identity with zero rhs, i.e. delta = 0; rhs -> g0 + (Schur part).
//
// Step s = three launches (one for the last tile):
//   k_ldl_diag    ONE workgroup factors the diagonal tile: the serial part, walked in 6x6 block pivots with the trailing
//                 matrix in fp64 MFMA accumulators (see the kernel).  Epilogue: inverses of the six 16x16 diagonal
//                 sub-blocks of L_ss, which turn the panel solve below into small matrix products.  For the last tile it
//                 also solves the right-hand side, forward and backward.
//   k_ldl_trsm    every 16-row slab of block column s (and the right-hand side as a 1-row slab), one wavefront each:
//                 X = A L_ss^-T by blocked substitution over 16-column blocks on the matrix pipes, then L_ts = X D^-1.
//   k_ldl_update  trailing tiles: S(I,J) -= L_Is D_s L_Js^T, rhs rows likewise.
// ------------------------------------------------------------------------------------------------
#ifdef AAR_STAMPS  // diagnostic build only (scripts/probe): cycle stamps of one workgroup, never compiled into libaar.so
__device__ unsigned long long g_stamps[64];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
// completed work: everything before the stamp has retired, nothing after it has been hoisted above it
#define STAMPW(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMPW(i) do { } while (0)
#endif
#ifdef AAR_TIMELINE
__device__ unsigned long long g_tl[3 * 16 * 5];   // per-step timeline of three wavefronts of k_ldl_diag, kept in LDS until the end
#define TL_DECL __shared__ unsigned long long tl_lds[3 * 16 * 5];
#define TL(k, p) do { __builtin_amdgcn_sched_barrier(0); const int w_ = threadIdx.x == 0 ? 0 : (threadIdx.x == 896 ? 1 : (threadIdx.x == 320 ? 2 : -1)); \
                      if (w_ >= 0) tl_lds[(w_ * 16 + (k)) * 5 + (p)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TL_DUMP do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x < 240) g_tl[threadIdx.x] = tl_lds[threadIdx.x]; } while (0)
#else
#define TL_DECL
#define TL(k, p) do { } while (0)
#define TL_DUMP do { } while (0)
#endif

constexpr int NB = CHOL_NB;
constexpr int SBK = 16;           // sub-block of the triangular solves
constexpr int NSB = NB / SBK;     // 6

__device__ __forceinline__ double xform_first(double v, int gi, int gj, int n, double mu, const int32_t *__restrict__ ent_fixed) {
    const bool fi = gi >= n || ent_fixed[gi / 6], fj = gj >= n || ent_fixed[gj / 6];
    if (fi || fj) return gi == gj ? 1.0 : 0.0;
    return gi == gj ? v + mu : v;
}
// the same with the two gauge flags already known
__device__ __forceinline__ double xform_first_flags(double v, int gi, int gj, bool fi, bool fj, double mu) {
    if (fi || fj) return gi == gj ? 1.0 : 0.0;
    return gi == gj ? v + mu : v;
}
// Gauge flags of 2 x 16 rows in ONE wave-wide ballot, so that the first-touch transform costs a single early load per wavefront
// instead of two dependent loads per matrix element: bit l (l < 16) = row base_a + step_a l is a gauge / padding row, bit 16 + l
// the same for base_b + step_b l.  Not on the first touch: 0, nothing loaded.
__device__ __forceinline__ unsigned long long gauge_ballot(bool first, const int32_t *__restrict__ ent_fixed, int n, int base_a, int step_a,
                                                           int base_b, int step_b) {
    if (!first) return 0ull;
    const int lane = threadIdx.x & 63;
    bool f = false;
    if (lane < 32) {
        const int g = lane < 16 ? base_a + step_a * lane : base_b + step_b * (lane - 16);
        f = g >= n || ent_fixed[g / 6] != 0;
    }
    return __ballot(f);
}

// The tile is walked in 6x6 block pivots (16 block steps instead of 96 column steps; a column step is latency, not work).
// A lone wavefront issues an instruction every 5-8 cycles here whatever the instruction is, so the design rule is few
// instructions per wavefront per step: 16 wavefronts, the trailing matrix in fp64 MFMA accumulators as the 21 lower 16x16
// sub-tiles (one per wavefront, two for five of them), dealt out in the order in which their columns retire.  Block step k:
//   rows(k)  one row thread per row below the pivot block (6(15-k) of them) reads its six entries of block column k and the
//            pivot block from LDS; EVERY one of them factors the pivot block in registers (same instruction stream, so the
//            redundancy costs no time and saves a barrier) and substitutes its own row in the shadow of the reciprocals:
//            Y = A_ik L_kk^-T, L_ik = Y D_k^-1.  L goes into the LDS tile in place, -Y into a panel padded to 8 columns.
//   C(k)     every live sub-tile: A -= L Y^T as two v_mfma_f64_16x16x4 (rank 6, padded to 8), operands straight from the
//            tile and the panel; the sub-tiles that hold block column k+1 publish it to LDS.
// Two barriers per block step.  The pivot blocks themselves stay unfactored in the tile until the epilogue, which factors
// all 16 at once; then the factored tile goes to Dfac, plus the inverses of the six 16x16 diagonal sub-blocks of L_ss,
// which turn every triangular solve below into small matrix products.
// f64 MFMA lane maps (cdna_hip_programming.md section 4, checked by scripts/probe/issue_probe.hip):
//   A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[row = (l>>4) + 4 reg][col = l&15].
// LDS (dynamic): T [NB][NB+2] | Yn [2][NB][DG_YS] | Lh [NB][NB/2+2] (look-ahead)
constexpr int DG_THREADS = 1024, DG_ROW0 = 896;
// doubles between panel rows: 8 are used (rank 6 padded to 8); at 10 the rows of 16 consecutive lanes tile the LDS banks (at 8 they
// fall on four bank groups: 4-way conflicts on every panel store of the row phase and every B-operand read)
constexpr int DG_YS = 10;

// LDL^T of the 6x6 pivot block at (c0, c0) in registers plus the substitution of row g of block column c0:
// on return y = (row g) L_kk^-T = L_g D and inv = 1 / D
__device__ __forceinline__ void dg_factor_row(const double *__restrict__ T, int c0, int g, double (&a)[6][6], double (&y)[6], double (&inv)[6]) {
    constexpr int LD = NB + 2;
    // the lower triangle of the pivot block in 16-byte pieces (c0 and LD are even): 12 LDS instructions instead of 21 -- on a lone
    // wavefront every instruction of the row phase is ~10 cycles of the critical path, whatever it does
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const double2 *rq = reinterpret_cast<const double2 *>(T + (c0 + q) * LD + c0);
#pragma unroll
        for (int h = 0; 2 * h <= q; h++) {
            const double2 v = rq[h];
            a[q][2 * h] = v.x;
            if (2 * h + 1 <= q) a[q][2 * h + 1] = v.y;
        }
    }
    {
        const double2 *rp = reinterpret_cast<const double2 *>(T + g * LD + c0);
        const double2 y0 = rp[0], y1 = rp[1], y2 = rp[2];
        y[0] = y0.x; y[1] = y0.y; y[2] = y1.x; y[3] = y1.y; y[4] = y2.x; y[5] = y2.y;
    }
#pragma unroll
    for (int p = 0; p < 6; p++) {
        inv[p] = rcp_refined(a[p][p]);
        double l[6];
#pragma unroll
        for (int q = p + 1; q < 6; q++) l[q] = a[q][p] * inv[p];
#pragma unroll
        for (int q = p + 1; q < 6; q++)
#pragma unroll
            for (int r = p + 1; r <= q; r++) a[q][r] = fma(-l[q], a[r][p], a[q][r]);
#pragma unroll
        for (int c = p + 1; c < 6; c++) y[c] = fma(-y[p], l[c], y[c]);
    }
}

// rows(k): rt = 6.. numbers the rows from the pivot block down
__device__ __forceinline__ void dg_rows(double *__restrict__ T, double *__restrict__ Yn, int k, int rt) {
    constexpr int LD = NB + 2;
    const int c0 = 6 * k, g = c0 + rt;
    if (rt < 6 || g >= NB) return;
    double a[6][6], y[6], inv[6];
    dg_factor_row(T, c0, g, a, y, inv);
    double2 *lt = reinterpret_cast<double2 *>(T + g * LD + c0);
    lt[0] = make_double2(y[0] * inv[0], y[1] * inv[1]);
    lt[1] = make_double2(y[2] * inv[2], y[3] * inv[3]);
    lt[2] = make_double2(y[4] * inv[4], y[5] * inv[5]);
    double2 *yp = reinterpret_cast<double2 *>(Yn + g * DG_YS);
    yp[0] = make_double2(y[0], y[1]);   // +Y: the trailing update negates its B operand (fp64 MFMA: blgp bit 1 = neg B)
    yp[1] = make_double2(y[2], y[3]);
    yp[2] = make_double2(y[4], y[5]);
}

// L_ss^T x = w by one wavefront: lane l keeps rows l and l + 64; row j of L (LDS tile, zero on and above the diagonal) against
// x_j, from the last column up; the finished entry travels through v_readlane
__device__ __forceinline__ void bs_sweep(const double *__restrict__ Ls, const double *__restrict__ w, int lane, double &b0, double &b1) {
    constexpr int LD = NB + 2;
    const int i0 = lane, i1 = lane + 64;
    const bool h1 = i1 < NB;
    b0 = w[i0]; b1 = h1 ? w[i1] : 0.0;
    const int i1c = h1 ? i1 : 0;
#pragma unroll
    for (int jj = NB - 1; jj > 0; jj--) {
        const double v = jj < 64 ? b0 : b1;
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), jj & 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), jj & 63);
        const double xj = __hiloint2double(hi, lo);
        b0 = fma(-Ls[jj * LD + i0], xj, b0);
        if (jj > 64) b1 = fma(-Ls[jj * LD + i1c], xj, b1);
    }
}

__global__ void __launch_bounds__(256) k_backsub(const BacksubArgs b) {
    __shared__ double red[32];
    backsub_body(b, (int)blockIdx.x, red, nullptr, 0, nullptr);
}

// Back-substitution of a system of two or three tiles by ONE workgroup that rides in the launch of the LAST diagonal tile
// (workgroup 1 of k_ldl_diag): every block of L it needs (at most three off-diagonal blocks and two diagonal tiles, 290 KB)
// is fetched while the diagonal tile is still being factored next door, so when x_{nT-1} is published (one agent-scope flag)
// only the arithmetic is left: matvec, sweep, [matvec, sweep].  Replaces the k_ldl_backsolve launch (its own fetch, one flag
// hop per tile column) for nT <= 3; larger systems keep the chained kernel (one CU could not pull their L through in time).
// LDS (the diagonal-tile kernel's allocation): Ls [NB][NB+2] | xs [2][NB] | w [NB] | part [10][NB]
__device__ __forceinline__ void bs_small(double *__restrict__ lds, const double *__restrict__ S, const double *__restrict__ rhs,
                                         const double *__restrict__ Dfac, double *__restrict__ x, int n_pad, int nT,
                                         const int32_t *__restrict__ flag, int epoch, int32_t *__restrict__ err_flags,
                                         const double *__restrict__ Lp, const double *__restrict__ zf, int fused_m, int32_t *__restrict__ done_flag) {
    constexpr int LD = NB + 2, G = 10, RPT = (NB + G - 1) / G, NL = NB * NB / 1024;
    double *Ls = lds, *xs = Ls + NB * LD, *w = xs + 2 * NB, *part = w + NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = tid % NB, gq = tid / NB;
    auto block = [&](int t, int sc, double (&a)[RPT]) {   // rows gq + G k of column j of L(t, sc): in place (k_ldl_trsm) or in Lp (k_ldl_panel)
        const bool from_lp = (nT - 1 - sc) <= fused_m;
        const double *lcol = from_lp ? Lp + (size_t)sc * n_pad * NB : S + sc * NB;
        const size_t lld = from_lp ? (size_t)NB : (size_t)n_pad;
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int i = gq + G * k;
            a[k] = (gq < G && i < NB) ? lcol[(size_t)(t * NB + i) * lld + j] : 0.0;
        }
    };
    auto tile = [&](int sc, double (&v)[NL]) {   // strictly lower part of L_{sc,sc}
        const double *dd = Dfac + (size_t)sc * NB * NB;
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int e = tid + 1024 * u, i = e / NB, jj = e - i * NB;
            v[u] = (jj < i) ? dd[e] : 0.0;
        }
    };
    auto park = [&](const double (&v)[NL]) {
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int e = tid + 1024 * u, i = e / NB, jj = e - i * NB;
            Ls[i * LD + jj] = v[u];
        }
    };
    const int s1 = nT - 2;                       // first tile column to solve
    double a0[RPT], a1[RPT], a2[RPT], v0[NL], v1[NL];
    block(nT - 1, s1, a0);
    tile(s1, v0);
    if (nT == 3) { block(2, 0, a1); block(1, 0, a2); }
    park(v0);
    if (nT == 3) tile(0, v1);
    if (tid == 0 && !hop_wait(flag + (nT - 1), epoch)) atomicOr(err_flags, 4);
    __syncthreads();
    if (tid < NB) xs[tid] = __hip_atomic_load(x + (nT - 1) * NB + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int i = gq + G * k;
        if (i < NB) { acc0 = fma(a0[k], xs[i], acc0); if (nT == 3) acc1 = fma(a1[k], xs[i], acc1); }
    }
    if (gq < G) part[gq * NB + j] = acc0;
    __syncthreads();
    const double *z1 = ((nT - 1 - s1) <= fused_m) ? zf : rhs;
    if (tid < NB) {
        double v = z1[s1 * NB + tid];
#pragma unroll
        for (int g = 0; g < G; g++) v -= part[g * NB + tid];
        w[tid] = v;
    }
    __syncthreads();
    if (wave == 0) {
        double b0, b1;
        bs_sweep(Ls, w, lane, b0, b1);
        __hip_atomic_store(x + s1 * NB + lane, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); xs[NB + lane] = b0;
        if (lane + 64 < NB) { __hip_atomic_store(x + s1 * NB + lane + 64, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); xs[NB + lane + 64] = b1; }
        if (nT != 3 && done_flag) hop_publish(done_flag, epoch, lane == 0);   // all of delta_s is in memory: the frame back-substitution riding in this launch may go
    }
    if (nT != 3) return;
    __syncthreads();
    park(v1);
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int i = gq + G * k;
        if (i < NB) acc1 = fma(a2[k], xs[NB + i], acc1);
    }
    if (gq < G) part[gq * NB + j] = acc1;
    __syncthreads();
    const double *z0 = (2 <= fused_m) ? zf : rhs;
    if (tid < NB) {
        double v = z0[tid];
#pragma unroll
        for (int g = 0; g < G; g++) v -= part[g * NB + tid];
        w[tid] = v;
    }
    __syncthreads();
    if (wave == 0) {
        double b0, b1;
        bs_sweep(Ls, w, lane, b0, b1);
        __hip_atomic_store(x + lane, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane + 64 < NB) __hip_atomic_store(x + lane + 64, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done_flag) hop_publish(done_flag, epoch, lane == 0);
    }
}

// Look-ahead (launch_chol): the trailing update of block column s does not get a launch of its own -- the NEXT diagonal tile's workgroup
// brings its own tile up to date in its prologue (see k_ldl_diag) and starts factoring, while these riders, workgroups of the same launch, do
// every other trailing tile and the right-hand side.  The body of k_ldl_update (below) for one wavefront = one 16 x 16 block, in two halves
// of k so that it fits the 128 registers of a 1024-thread workgroup.  zs = D_s in LDS.
__device__ __forceinline__ void ldl_update_block_lean(double *__restrict__ S, const double *__restrict__ zs, int n_pad, int n, int s, double mu,
                                                      const int32_t *__restrict__ ent_fixed, int ti, int tj, int q) {
    constexpr int RUN = NB / 4, HALF = RUN / 2;
    const int lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
    const int bi = q / (NB / 16), bj = q % (NB / 16);
    if (ti == tj && bj > bi) return;   // blocks above the diagonal of a diagonal tile are never read
    const bool first = (s == 0);
    const int r0 = s * NB, i0 = (s + 1 + ti) * NB + 16 * bi, j0 = (s + 1 + tj) * NB + 16 * bj;
    const unsigned long long gmask = gauge_ballot(first, ent_fixed, n, i0, 1, j0, 1);
    double tgt[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int gi = i0 + lr + 4 * r, gj = j0 + lc;
        tgt[r] = (gj <= gi) ? S[(size_t)gi * n_pad + gj] : 0.0;
    }
    dg_acc_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const double2 *pa = reinterpret_cast<const double2 *>(S + (size_t)(i0 + lc) * n_pad + r0 + RUN * lr + HALF * h);
        const double2 *pb = reinterpret_cast<const double2 *>(S + (size_t)(j0 + lc) * n_pad + r0 + RUN * lr + HALF * h);
        double2 va[HALF / 2], vb[HALF / 2];
#pragma unroll
        for (int u = 0; u < HALF / 2; u++) { va[u] = pa[u]; vb[u] = pb[u]; }
#pragma unroll
        for (int u = 0; u < HALF / 2; u++) {
            const double d0 = zs[RUN * lr + HALF * h + 2 * u], d1 = zs[RUN * lr + HALF * h + 2 * u + 1];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u].x, vb[u].x * d0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u].y, vb[u].y * d1, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int gi = i0 + lr + 4 * r, gj = j0 + lc;
        if (gj <= gi) {
            double v = tgt[r];
            if (first) v = xform_first_flags(v, gi, gj, (gmask >> (lr + 4 * r)) & 1, (gmask >> (16 + lc)) & 1, mu);
            S[(size_t)gi * n_pad + gj] = v - acc[r];
        }
    }
}
// rider workgroup rb of the update of block column s: sixteen blocks of the trailing tiles other than the first (= the next diagonal tile), or
// -- the last workgroups -- the right-hand side rows of ten row tiles each
constexpr int UPD_RHS_TILES = DG_THREADS / NB;
__host__ __device__ constexpr int ldl_update_riders_blk(int m) { return ((m * (m + 1) / 2 - 1) * 36 + 15) / 16; }
__host__ __device__ constexpr int ldl_update_riders(int m) { return ldl_update_riders_blk(m) + (m + UPD_RHS_TILES - 1) / UPD_RHS_TILES; }
__device__ __forceinline__ void ldl_update_rider(double *__restrict__ zs, double *__restrict__ S, double *__restrict__ rhs, const double *__restrict__ g0,
                                                 const double *__restrict__ Dfac, int n_pad, int n, int s, int nT, double mu,
                                                 const int32_t *__restrict__ ent_fixed, int rb) {
    const int m = nT - s - 1, nblk = (m * (m + 1) / 2 - 1) * 36, nbw = ldl_update_riders_blk(m);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = s * NB;
    const double *dd = Dfac + (size_t)s * NB * NB;
    if (rb < nbw) {
        if (tid < NB) zs[tid] = dd[tid * NB + tid];
        __syncthreads();
        const int b = rb * 16 + wave;
        if (b >= nblk) return;
        const int tile = 1 + b / 36;
        int ti = 0, rem = tile;
        while (rem > ti) { rem -= ti + 1; ti++; }
        ldl_update_block_lean(S, zs, n_pad, n, s, mu, ent_fixed, ti, rem, b % 36);
        return;
    }
    if (tid < NB) zs[tid] = rhs[r0 + tid] * dd[tid * NB + tid];   // D_s z_s
    __syncthreads();
    const int k = tid / NB, i = tid - k * NB, t = s + 1 + (rb - nbw) * UPD_RHS_TILES + k;
    if (k < UPD_RHS_TILES && t < nT) {
        const int gi = t * NB + i;
        const double *row = S + (size_t)gi * n_pad + r0;
        double acc = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; c++) acc += row[c] * zs[c];
        double v = rhs[gi];
        if (s == 0) v = (gi >= n || ent_fixed[gi / 6]) ? 0.0 : v + g0[gi];
        rhs[gi] = v - acc;
    }
}

__global__ void __launch_bounds__(DG_THREADS) k_ldl_diag(const double *__restrict__ S, double *__restrict__ Dfac, double *__restrict__ Linv16,
                                                         int n_pad, int n, int s, double mu, const int32_t *__restrict__ ent_fixed,
                                                         int32_t *__restrict__ flags, int nT, const double *__restrict__ rhs,
                                                         const double *__restrict__ g0, double *__restrict__ xout,
                                                         const double *__restrict__ Lp, const double *__restrict__ zf, int fused_m,
                                                         int32_t *__restrict__ bs_flag, int bs_epoch, int ride_bs, int ride_backsub, const BacksubArgs bsub,
                                                         int upd_s, double *__restrict__ Sw, double *__restrict__ rhsw) {
    constexpr int LD = NB + 2, NBK = NB / 6, PER = NB * NB / DG_THREADS, NT16 = NB / 16, NTILE = NT16 * (NT16 + 1) / 2, NWAVE = DG_THREADS / 64, NMW = DG_ROW0 / 64;
    static_assert(NB % 16 == 0 && NB % 6 == 0 && NB * NB % DG_THREADS == 0 && NWAVE >= NSB && DG_THREADS - DG_ROW0 >= NB - 6 &&
                  NTILE <= 2 * NMW && DG_ROW0 % 64 == 0, "diag tile mapping");
    extern __shared__ __align__(16) double T[];
    TL_DECL
    // riders of the LAST tile's launch: workgroup 1 = the back-substitution of the tiles above (two or three tiles, see bs_small);
    // the workgroups behind it = the frame back-substitution (backsub_body), which waits for the last piece of delta_s:
    // flag[nT] from bs_small, or flag[nT - 1] from this tile's own solve when it is the only tile
    if (upd_s >= 0 && blockIdx.x > 0) {   // look-ahead: the rest of block column upd_s's trailing update rides beside this tile's factorisation
        ldl_update_rider(T, Sw, rhsw, g0, Dfac, n_pad, n, upd_s, nT, mu, ent_fixed, (int)blockIdx.x - 1);
        return;
    }
    if (ride_bs && blockIdx.x == 1) {
        bs_small(T, S, rhs, Dfac, xout, n_pad, nT, bs_flag, bs_epoch, flags, Lp, zf, fused_m, ride_backsub ? bs_flag + nT : nullptr);
        return;
    }
    if (blockIdx.x > 0) {
        backsub_body(bsub, (int)blockIdx.x - 1 - ride_bs, T, bs_flag + (ride_bs ? nT : nT - 1), bs_epoch, flags);
        return;
    }
    double *Yn = T + NB * LD;   // [2][NB][DG_YS]: L D of block column k in buffer k & 1, columns 6, 7 zero
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = s * NB;
    const bool first = (s == 0);
    const bool touch = first || upd_s == 0;   // nobody has touched this tile yet: damping / gauge go in as it is loaded (look-ahead: block column 0's update happens HERE)
    STAMP(0);
    const unsigned long long gmask = gauge_ballot(touch, ent_fixed, n, r0, 6, r0, 6);   // bit e: entity e of this tile is a gauge / padding entity
    // look-ahead: the panel tile of block column upd_s (both halves of k) is requested first, so that it travels while the tile is staged
    constexpr int KH = NB / 2, LHS = KH + 2, PVN = (NB * KH / 2 + DG_THREADS - 1) / DG_THREADS;   // LDS rows 50 doubles apart: the 16 rows of an operand fetch tile the banks
    double2 pv[2][PVN];
    if (upd_s >= 0) {
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
#pragma unroll
            for (int u = 0; u < PVN; u++) {
                const int idx = tid + DG_THREADS * u, row = idx / (KH / 2), c2 = idx - row * (KH / 2);
                pv[hh][u] = make_double2(0.0, 0.0);
                if (idx < NB * KH / 2) pv[hh][u] = reinterpret_cast<const double2 *>(S + (size_t)(r0 + row) * n_pad + upd_s * NB + KH * hh)[c2];
            }
    }
    {   // tile -> LDS, coalesced; damping / gauge on first touch; zeros above the diagonal
        double v[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int idx = tid + DG_THREADS * u, i = idx / NB, j = idx - i * NB;
            v[u] = (j <= i) ? S[(size_t)(r0 + i) * n_pad + r0 + j] : 0.0;
        }
        for (int i = tid; i < 2 * NB * DG_YS; i += DG_THREADS) Yn[i] = 0.0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int idx = tid + DG_THREADS * u, i = idx / NB, j = idx - i * NB;
            double x = v[u];
            if (touch && j <= i) x = xform_first_flags(x, r0 + i, r0 + j, (gmask >> (i / 6)) & 1, (gmask >> (j / 6)) & 1, mu);
            T[i * LD + j] = x;
        }
    }
    __syncthreads();
    STAMP(1);
    const int rt = tid - DG_ROW0 + 6;     // row thread: row number counted from the pivot block (< 6: not one)
    // sub-tiles of a matrix wavefront (wave-uniform): tile number 14 slot + wave in the order (column descending, row ascending)
    int j16[2], i16[2], aoff[2], boff[2], toff[2];
    bool has[2];
    dg_acc_t acc[2];
    const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
        const int nn = NMW * sl + wave;
        int jj = 0;
        while ((jj + 1) * (jj + 2) / 2 <= nn) jj++;
        has[sl] = wave < NMW && nn < NTILE;
        const int tj = has[sl] ? NT16 - 1 - jj : 0, ti = has[sl] ? tj + nn - jj * (jj + 1) / 2 : 0;
        j16[sl] = 16 * tj;
        i16[sl] = 16 * ti;
        aoff[sl] = (16 * ti + lc) * LD + lr;          // A operand: L(row, c0 + k) in the tile
        boff[sl] = (16 * tj + lc) * DG_YS + lr;           // B operand: -(L D)(col, k) in the panel
        toff[sl] = (16 * ti + lr) * LD + 16 * tj + lc;   // accumulator register r: row + 4 r
#pragma unroll
        for (int r = 0; r < 4; r++) acc[sl][r] = T[toff[sl] + 4 * r * LD];
    }
    if (upd_s >= 0) {
        // Look-ahead: block column upd_s's update of THIS tile, A -= L D L^T with L = this tile row of the column's panel (k_ldl_trsm left it in
        // S), straight into the accumulators.  The 96 x 96 panel tile comes through LDS in two halves of k (one CU pulls ~13 bytes per cycle:
        // every sub-tile fetching its own operand slabs from memory, as k_ldl_update's wavefronts do on 9 CUs, costs 7x the bytes here);
        // D_upd sits in the padding column of the tile rows.  The sub-tiles then go back to the LDS tile for the row threads.
        double *Lh = Yn + 2 * NB * DG_YS;          // [NB][LHS]
        const double *du = Dfac + (size_t)upd_s * NB * NB;
        if (tid < NB) T[tid * LD + NB] = du[tid * NB + tid];
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            if (hh) __syncthreads();   // the first half has been read
#pragma unroll
            for (int u = 0; u < PVN; u++) {
                const int idx = tid + DG_THREADS * u, row = idx / (KH / 2), c2 = idx - row * (KH / 2);
                if (idx < NB * KH / 2) *reinterpret_cast<double2 *>(Lh + row * LHS + 2 * c2) = pv[hh][u];
            }
            __syncthreads();
            auto sub = [&](dg_acc_t &a, bool live, int ia, int ja) {
                if (!live) return;   // wave-uniform
                const double *la = Lh + (ia + lc) * LHS + lr, *lb = Lh + (ja + lc) * LHS + lr;
#pragma unroll
                for (int t = 0; t < KH / 4; t++) {
                    const double d = T[(KH * hh + 4 * t + lr) * LD + NB];
                    a = __builtin_amdgcn_mfma_f64_16x16x4f64(la[4 * t], lb[4 * t] * d, a, 0, 0, 2);   // (blgp bit 1: B negated)
                }
            };
            sub(acc[0], has[0], i16[0], j16[0]);
            sub(acc[1], has[1], i16[1], j16[1]);
        }
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
            if (has[sl])
#pragma unroll
                for (int r = 0; r < 4; r++) T[toff[sl] + 4 * r * LD] = acc[sl][r];
        __syncthreads();
    }
    dg_rows(T, Yn, 0, rt);
    __syncthreads();

    for (int k = 0; k + 1 < NBK; k++) {
        const int c0 = 6 * k, lim = c0 + 6;
        const double *yk = Yn + (k & 1) * NB * DG_YS;
        TL(k, 0);
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {   // first the sub-tiles that hold (part of) block column k+1: update, publish
            if (!has[sl] || j16[sl] + 16 <= lim || j16[sl] >= lim + 6) continue;   // wave-uniform
            const double a1 = T[aoff[sl] + c0], a2 = T[aoff[sl] + c0 + 4];   // rank columns 6, 7: whatever follows in the row, times the zeros of the panel
            const double b1 = yk[boff[sl]], b2 = yk[boff[sl] + 4];
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[sl], 0, 0, 2);
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[sl], 0, 0, 2);
            const int col = j16[sl] + lc;
            if (col >= lim && col < lim + 6) {   // rows above the block land above the diagonal
#pragma unroll
                for (int r = 0; r < 4; r++) T[toff[sl] + 4 * r * LD] = acc[sl][r];
            }
        }
        TL(k, 1);
        __syncthreads();
        TL(k, 2);
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {   // then the other live sub-tiles, beside rows(k+1) on the row wavefronts
            if (!has[sl] || j16[sl] < lim + 6) continue;   // wave-uniform
            const double a1 = T[aoff[sl] + c0], a2 = T[aoff[sl] + c0 + 4];
            const double b1 = yk[boff[sl]], b2 = yk[boff[sl] + 4];
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[sl], 0, 0, 2);
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[sl], 0, 0, 2);
        }
        dg_rows(T, Yn + ((k + 1) & 1) * NB * DG_YS, k + 1, rt);
        TL(k, 3);
        __syncthreads();
        TL(k, 4);
    }
    STAMP(8);
    double *dsave = Yn;   // [NB] the pivots (the panel is free after the last block step)
    {   // the 16 pivot blocks, still unfactored: thread (block, row) -> unit L left of the diagonal, zeros from it on; D aside
        const int c0 = 6 * (tid / 6), q = tid % 6;
        double o[6];
        if (tid < NB) {
            double a[6][6], y[6], inv[6];
            dg_factor_row(T, c0, c0 + q, a, y, inv);
            double dq = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                o[c] = (c < q) ? y[c] * inv[c] : 0.0;
                dq = (c == q) ? y[c] : dq;
            }
            dsave[c0 + q] = dq;   // D: kept beside the tile, whose diagonal stays zero for the triangular solves below
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 6; c++) bad |= !(a[c][c] > 0.0);
            if (bad && q == 0) atomicOr(flags, 2);
        }
        __syncthreads();   // every row thread of a block has read it before any of them stores
        if (tid < NB) {
            double2 *lt = reinterpret_cast<double2 *>(T + (c0 + q) * LD + c0);
            lt[0] = make_double2(o[0], o[1]);
            lt[1] = make_double2(o[2], o[3]);
            lt[2] = make_double2(o[4], o[5]);
        }
#pragma unroll
        for (int u = 0; u < PER; u++) {   // what the publications left above the diagonal blocks
            const int idx = tid + DG_THREADS * u, i = idx / NB, j = idx - i * NB;
            if (j / 6 > i / 6) T[i * LD + j] = 0.0;
        }
    }
    __syncthreads();
    if (s == nT - 1 && wave == NWAVE - 1) {
        // The last tile has no panel below it: its right-hand side is solved here, forward and backward, by one wavefront
        // beside the stores and inverses of the others -- instead of a k_ldl_trsm launch and a tile of k_ldl_backsolve.
        // Lane l keeps rows l and l + 64; column by column, the finished entry travels through v_readlane.
        const int i0 = lane, i1 = lane + 64;
        const bool h1 = i1 < NB;
        double b0 = rhs[r0 + i0], b1 = h1 ? rhs[r0 + i1] : 0.0;
        if (first) {   // B = g0 + Schur part on free rows, 0 on gauge / padding rows (as k_ldl_trsm does for its rhs slab)
            const int g0i = r0 + i0, g1i = r0 + i1;
            b0 = ((gmask >> (i0 / 6)) & 1) ? 0.0 : b0 + g0[g0i];
            if (h1) b1 = ((gmask >> (i1 / 6)) & 1) ? 0.0 : b1 + g0[g1i];
        }
        const double *c0p = T + i0 * LD, *c1p = T + (h1 ? i1 : 0) * LD;
        auto bcast = [&](double v0, double v1, int j) -> double {   // entry j of the vector held as (v0: rows 0-63, v1: rows 64-95)
            const double v = j < 64 ? v0 : v1;
            const int lo = __builtin_amdgcn_readlane(__double2loint(v), j & 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), j & 63);
            return __hiloint2double(hi, lo);
        };
#pragma unroll
        for (int j = 0; j < NB - 1; j++) {   // L y = b, unit lower: T is zero on and above the diagonal, so no masks
            const double yj = bcast(b0, b1, j);
            if (j < 63) b0 = fma(-c0p[j], yj, b0);
            b1 = fma(-c1p[j], yj, b1);
        }
        b0 *= rcp_refined(dsave[i0]);           // z = D^-1 y
        b1 *= rcp_refined(h1 ? dsave[i1] : 1.0);
#pragma unroll
        for (int j = NB - 1; j > 0; j--) {   // L^T x = z: row j of L against x_j
            const double xj = bcast(b0, b1, j);
            b0 = fma(-T[j * LD + i0], xj, b0);
            if (j > 64) b1 = fma(-T[j * LD + i1], xj, b1);
        }
        __hip_atomic_store(xout + r0 + i0, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h1) __hip_atomic_store(xout + r0 + i1, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gridDim.x > 1) hop_publish(bs_flag + s, bs_epoch, lane == 0);   // riders next door: x_{nT-1} is in memory
    }
    if (s == nT - 1) { STAMP(9); STAMP(10); TL_DUMP; return; }   // nobody reads the last tile's factor: its solve is done
    {   // factored tile -> Dfac (not back into S: other workgroups may still be reading it); only the lower part is ever read
        double *outp = Dfac + (size_t)s * NB * NB;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int idx = tid + DG_THREADS * u, i = idx / NB, j = idx - i * NB;
            if (j <= i) outp[idx] = (i == j) ? dsave[i] : T[i * LD + j];
        }
    }
    STAMP(9);
    // inverses of the unit-lower 16x16 diagonal sub-blocks, one wavefront each, by halves:
    //   [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1  C^-1]: the two 8x8 inverses column by column on 16 lanes (28 fma deep instead of
    //   120), then the two 8x8x8 products on all 64 lanes, through a wave-private LDS scratch (T is zero on and above the diagonal)
    if (wave < NSB) {
        const int q = wave, o = q * SBK;
        double *sc = Yn + 128 + wave * 192;   // Ai [8][8] | Ci [8][8] | B Ai [8][8]   (the panel is free by now; dsave = Yn[0..NB))
        if (lane < 16) {
            const int half = lane >> 3, c = lane & 7, ob = o + 8 * half;
            double x[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // x_p = 0 for p < c, so the full row can be used: the L loads are lane-uniform per half and independent of x
                double a2 = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int pp = 0; pp < i; pp++) a2 = fma(-T[(ob + i) * LD + ob + pp], x[pp], a2);
                x[i] = a2;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) sc[half * 64 + i * 8 + c] = x[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int r = lane >> 3, c = lane & 7;
        double t = 0.0;
#pragma unroll
        for (int pp = 0; pp < 8; pp++) t = fma(T[(o + 8 + r) * LD + o + pp], sc[pp * 8 + c], t);
        sc[128 + r * 8 + c] = t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double v = 0.0;
#pragma unroll
        for (int pp = 0; pp < 8; pp++) v = fma(-sc[64 + r * 8 + pp], sc[128 + pp * 8 + c], v);
        double *li = Linv16 + ((size_t)s * NSB + q) * SBK * SBK;
        li[r * SBK + c] = sc[r * 8 + c];
        li[r * SBK + 8 + c] = 0.0;
        li[(8 + r) * SBK + c] = v;
        li[(8 + r) * SBK + 8 + c] = sc[64 + r * 8 + c];
    }
    STAMP(10);
    TL_DUMP;
}

// X = A L_ss^-T D^-1 for one 16-row slab of block column s (blockIdx < 6m), or for the right-hand side (last block),
// right-looking over 16-column blocks on the matrix pipes.  ONE wavefront per slab: its six 16x16 fp64 MFMA accumulator tiles,
// the fifteen strictly-lower 16x16 blocks of L_ss and the six inverted diagonal blocks all live in registers, fetched straight
// from global memory up front (no LDS staging of L_ss, no workgroup barrier), so the first block's MFMAs start as soon as ITS
// operands have landed and the rest of the fetch hides behind the chain.  Per block q
//   X_q = R_q inv(L_qq)^T        (4 MFMAs; R_q goes through a wave-private LDS tile to become the A operand)
//   R_q'' -= X_q L(q'', q)^T      for every later block (4 MFMAs each)
// and L_ts = X_q D^-1 goes to global memory.  Lane (lc, lr) holds columns [4 lr, 4 lr + 4) of row lc of every operand block, so
// MFMA step t contracts columns {t, 4 + t, 8 + t, 12 + t}: the same permutation of k on both operands.
// LDS (static): St [16][18]
__global__ void __launch_bounds__(64) k_ldl_trsm(double *__restrict__ S, double *__restrict__ rhs, const double *__restrict__ g0,
                                                 const double *__restrict__ Dfac, const double *__restrict__ Linv16, int n_pad,
                                                 int n, int s, int nT, double mu, const int32_t *__restrict__ ent_fixed) {
    constexpr int PL = SBK + 2;
    __shared__ __align__(16) double St[SBK * PL];
    const int m = nT - s - 1, b = blockIdx.x, lane = threadIdx.x;
    const int r0 = s * NB;
    const bool first = (s == 0), is_rhs = (b == NSB * m);
    const int row0 = (s + 1) * NB + SBK * b;   // first slab row of this wavefront
    const double *dd = Dfac + (size_t)s * NB * NB;
    const int lr = lane >> 4, lc = lane & 15;
    const unsigned long long gmask = gauge_ballot(first, ent_fixed, n, is_rhs ? r0 : row0, 1, r0, 6);   // bits 0-15: slab rows, 16-31: entities of tile s
    dg_acc_t acc[NSB];
    double4 li[NSB];                   // inv(L_qq)[lc][4 lr ..]
    double4 lb[NSB * (NSB - 1) / 2];   // L(16 q2 + lc, 16 q + 4 lr ..), q < q2, in the order they are consumed
    double dq[NSB];
#pragma unroll
    for (int q = 0; q < NSB; q++) {   // issue order = use order: block q's operands first
#pragma unroll
        for (int r = 0; r < 4; r++) {   // accumulator register r of tile q: slab row lr + 4 r, column 16 q + lc
            double v = 0.0;
            if (!is_rhs) v = S[(size_t)(row0 + lr + 4 * r) * n_pad + r0 + 16 * q + lc];
            else if (lr + 4 * r == 0) v = rhs[r0 + 16 * q + lc];
            acc[q][r] = v;
        }
        li[q] = *reinterpret_cast<const double4 *>(Linv16 + ((size_t)s * NSB + q) * SBK * SBK + lc * SBK + 4 * lr);
        dq[q] = dd[(16 * q + lc) * (NB + 1)];
#pragma unroll
        for (int q2 = q + 1; q2 < NSB; q2++)
            lb[q * NSB - q * (q + 1) / 2 + (q2 - q - 1)] = *reinterpret_cast<const double4 *>(dd + (size_t)(16 * q2 + lc) * NB + 16 * q + 4 * lr);
    }
    if (first) {
#pragma unroll
        for (int q = 0; q < NSB; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gj = r0 + 16 * q + lc;
                const bool fj = (gmask >> (16 + (16 * q + lc) / 6)) & 1;
                if (!is_rhs) acc[q][r] = xform_first_flags(acc[q][r], row0 + lr + 4 * r, gj, (gmask >> (lr + 4 * r)) & 1, fj, mu);
                else if (lr + 4 * r == 0) acc[q][r] = fj ? 0.0 : acc[q][r] + g0[gj];  // B = g0 + Schur part
            }
    }
#pragma unroll
    for (int q = 0; q < NSB; q++) {
        // R_q -> wave-private LDS tile (accumulator layout: row lr + 4 r, column lc), read back as the A operand: row lc, columns 4 lr ..
#pragma unroll
        for (int r = 0; r < 4; r++) St[(lr + 4 * r) * PL + lc] = acc[q][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double ar[4];
#pragma unroll
        for (int t = 0; t < 4; t++) ar[t] = St[lc * PL + 4 * lr + t];
        dg_acc_t x = {0.0, 0.0, 0.0, 0.0};   // X[r][c] = sum_p R[r][p] inv(L_qq)[c][p]
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[0], li[q].x, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[1], li[q].y, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2], li[q].z, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[3], li[q].w, x, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
        const double dinv = rcp_refined(dq[q]);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            St[(lr + 4 * r) * PL + lc] = -x[r];
            if (!is_rhs) S[(size_t)(row0 + lr + 4 * r) * n_pad + r0 + 16 * q + lc] = x[r] * dinv;
            else if (lr + 4 * r == 0) rhs[r0 + 16 * q + lc] = x[r] * dinv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (q + 1 < NSB) {
            double ax[4];
#pragma unroll
            for (int t = 0; t < 4; t++) ax[t] = St[lc * PL + 4 * lr + t];   // -X_q as the A operand
#pragma unroll
            for (int q2 = q + 1; q2 < NSB; q2++) {   // R_q2[r][c] -= sum_p X[r][p] L(16 q2 + c, 16 q + p)
                const double4 l4 = lb[q * NSB - q * (q + 1) / 2 + (q2 - q - 1)];
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[0], l4.x, acc[q2], 0, 0, 0);
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[1], l4.y, acc[q2], 0, 0, 0);
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[2], l4.z, acc[q2], 0, 0, 0);
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[3], l4.w, acc[q2], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// trailing update of step s in 16x16 output blocks on the matrix pipe: S(I, J) -= L_Is D_s L_Js^T; rhs rows: b_t -= L_ts D_s z_s.
// grid: [tiles (ti >= tj)] x 9 workgroups of four wavefronts (one 16x16 block each), then one workgroup per rhs row tile.
__global__ void __launch_bounds__(256) k_ldl_update(double *__restrict__ S, double *__restrict__ rhs, const double *__restrict__ g0,
                                                    const double *__restrict__ Dfac, int n_pad, int n, int s, int nT,
                                                    double mu, const int32_t *__restrict__ ent_fixed) {
    // Tile part: one wavefront per 16 x 16 block of a trailing tile, S(I,J) -= L_Is (L_Js D_s)^T on the matrix pipe.  Lane (lc, lr)
    // fetches the contiguous run [24 lr, 24 lr + 24) of row lc of both panels straight from global memory (12 16-byte loads
    // each); MFMA step t then contracts columns {t, 24 + t, 48 + t, 72 + t} -- the same permutation of k on both operands, so
    // the 24 steps cover every k once.  No LDS staging, no workgroup barrier on the way to the first MFMA.
    constexpr int NSUB = 9, RUN = NB / 4;   // 9 workgroups x 4 wavefronts = the 36 blocks of a 96 x 96 tile
    __shared__ double zs[NB];
    const int m = nT - s - 1, tid = threadIdx.x;
    const int ntile = m * (m + 1) / 2;
    const int r0 = s * NB;
    const bool first = (s == 0);
    const double *dd = Dfac + (size_t)s * NB * NB;
    if ((int)blockIdx.x >= ntile * NSUB) {  // rhs entries of row tile t
        const int t = s + 1 + ((int)blockIdx.x - ntile * NSUB);
        if (tid < NB) zs[tid] = rhs[r0 + tid] * dd[tid * NB + tid];  // D_s z_s
        __syncthreads();
        if (tid < NB) {
            const int gi = t * NB + tid;
            const double *row = S + (size_t)gi * n_pad + r0;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; k++) acc += row[k] * zs[k];
            double v = rhs[gi];
            if (first) v = (gi >= n || ent_fixed[gi / 6]) ? 0.0 : v + g0[gi];
            rhs[gi] = v - acc;
        }
        return;
    }
    const int tile = blockIdx.x / NSUB, sub = blockIdx.x % NSUB;
    int ti = 0, rem = tile;  // decode (ti >= tj) from the linear lower-triangular tile index
    while (rem > ti) { rem -= ti + 1; ti++; }
    const int tj = rem;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane >> 4, lc = lane & 15;
    const int q = sub * 4 + wave, bi = q / (NB / 16), bj = q % (NB / 16);
    // D_s for the B operand: the diagonal of the factor tile, through LDS (every wavefront needs all 96 values)
    double dv = 0.0;
    if (tid < NB) dv = dd[tid * NB + tid];
    const bool live = !(ti == tj && bj > bi);   // blocks above the diagonal of a diagonal tile are never read
    const int i0 = (s + 1 + ti) * NB + 16 * bi, j0 = (s + 1 + tj) * NB + 16 * bj;
    const unsigned long long gmask = gauge_ballot(first, ent_fixed, n, i0, 1, j0, 1);   // bits 0-15: rows of this block, 16-31: its columns
    double2 va[RUN / 2], vb[RUN / 2];
    double tgt[4];
    if (live) {
        const double2 *pa = reinterpret_cast<const double2 *>(S + (size_t)(i0 + lc) * n_pad + r0 + RUN * lr);
        const double2 *pb = reinterpret_cast<const double2 *>(S + (size_t)(j0 + lc) * n_pad + r0 + RUN * lr);
#pragma unroll
        for (int u = 0; u < RUN / 2; u++) { va[u] = pa[u]; vb[u] = pb[u]; }
#pragma unroll
        for (int r = 0; r < 4; r++) {   // accumulator register r: row lr + 4 r, column lc
            const int gi = i0 + lr + 4 * r, gj = j0 + lc;
            tgt[r] = (gj <= gi) ? S[(size_t)gi * n_pad + gj] : 0.0;
        }
    }
    if (tid < NB) zs[tid] = dv;
    __syncthreads();
    if (!live) return;
    dg_acc_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < RUN / 2; u++) {
        const double d0 = zs[RUN * lr + 2 * u], d1 = zs[RUN * lr + 2 * u + 1];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u].x, vb[u].x * d0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u].y, vb[u].y * d1, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int gi = i0 + lr + 4 * r, gj = j0 + lc;
        if (gj <= gi) {
            double v = tgt[r];
            if (first) v = xform_first_flags(v, gi, gj, (gmask >> (lr + 4 * r)) & 1, (gmask >> (16 + lc)) & 1, mu);
            S[(size_t)gi * n_pad + gj] = v - acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Panel solve AND trailing update of step s in ONE launch, for short block columns (m = nT - s - 1 <= 3 tiles below the
// diagonal tile by default: every reduced system up to 384 unknowns; AAR_FUSED_PANEL moves the limit -- measured at config 5,
// 14 tiles: the same time up to m = 4, slower beyond, 37.6 us against 10.8 + 11.8 at m = 13).  At these sizes each of the two kernels above is ~5 us of fixed
// cost (launch, first fetch of data another XCD has just written, completion) around <= 3 us of matrix work, and chaining them
// with flags costs what the launch boundary does.  Here nobody waits for anybody: the workgroup of output block (I, J)
// (16 x 16, slabs I >= J of the block column, or J = the right-hand side) solves BOTH slabs it needs itself -- one wavefront
// each, the same blocked substitution as k_ldl_trsm, redundantly with the other workgroups that need the same slab -- and
// then forms S(I, J) -= X_I D^-1 X_J^T from the two results in LDS (12 MFMAs per wavefront).  The diagonal workgroups (I, I)
// also store L_I = X_I D^-1 for the back-substitution (into Lp, not in place: the block column is still being read by the
// others); workgroup (0, rhs) stores z_s (into zf).  78 + 12 workgroups at m = 2.
// LDS (static): St [2][16][18] | Xl [2][16][NB + 2] | dinv [NB] | red [256]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) k_ldl_panel(double *__restrict__ S, double *__restrict__ rhs, const double *__restrict__ g0,
                                                   const double *__restrict__ Dfac, const double *__restrict__ Linv16, int n_pad,
                                                   int n, int s, int nT, double mu, const int32_t *__restrict__ ent_fixed,
                                                   double *__restrict__ Lp, double *__restrict__ zf) {
    constexpr int PL = SBK + 2, XL = NB + 2;
    __shared__ __align__(16) double St_all[2 * SBK * PL];
    __shared__ __align__(16) double Xl_all[2 * SBK * XL];
    __shared__ double dinv_l[NB];
    __shared__ double red[256];
    const int m = nT - s - 1, ns = NSB * m, npair = ns * (ns + 1) / 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lr = lane >> 4, lc = lane & 15;
    int I, J;   // slabs (16 rows each) counted from the first row below tile s; J == ns: the right-hand side
    if ((int)blockIdx.x < npair) {
        int ii = 0, rem = (int)blockIdx.x;
        while (rem > ii) { rem -= ii + 1; ii++; }
        I = ii; J = rem;
    } else {
        I = (int)blockIdx.x - npair; J = ns;
    }
    const int mine = wave == 0 ? I : J;           // the slab this wavefront solves
    const bool is_rhs = (mine == ns);
    const int r0 = s * NB;
    const bool first = (s == 0);
    const int row0 = (s + 1) * NB + SBK * mine;
    const double *dd = Dfac + (size_t)s * NB * NB;
    double *St = St_all + wave * SBK * PL;
    double *Xl = Xl_all + wave * SBK * XL;
    const unsigned long long gmask = gauge_ballot(first, ent_fixed, n, is_rhs ? r0 : row0, 1, r0, 6);   // bits 0-15: slab rows, 16-31: entities of tile s
    dg_acc_t acc[NSB];
    double4 li[NSB];
    double4 lb[NSB * (NSB - 1) / 2];
    double dq[NSB];
#pragma unroll
    for (int q = 0; q < NSB; q++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = 0.0;
            if (!is_rhs) v = S[(size_t)(row0 + lr + 4 * r) * n_pad + r0 + 16 * q + lc];
            else if (lr + 4 * r == 0) v = rhs[r0 + 16 * q + lc];
            acc[q][r] = v;
        }
        li[q] = *reinterpret_cast<const double4 *>(Linv16 + ((size_t)s * NSB + q) * SBK * SBK + lc * SBK + 4 * lr);
        dq[q] = dd[(16 * q + lc) * (NB + 1)];
#pragma unroll
        for (int q2 = q + 1; q2 < NSB; q2++)
            lb[q * NSB - q * (q + 1) / 2 + (q2 - q - 1)] = *reinterpret_cast<const double4 *>(dd + (size_t)(16 * q2 + lc) * NB + 16 * q + 4 * lr);
    }
    // the output block's own values (wavefront 0 finishes the block): rows of slab I, columns of slab J (or the rhs entries of slab I)
    const int i0 = (s + 1) * NB + SBK * I, j0 = (s + 1) * NB + SBK * (J < ns ? J : 0);
    double tgt[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long omask = 0ull;
    if (wave == 0) {
        omask = gauge_ballot(first, ent_fixed, n, i0, 1, j0, 1);   // bits 0-15: rows of the block, 16-31: its columns
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int gi = i0 + lr + 4 * r, gj = j0 + lc;
            if (J < ns) tgt[r] = (gj <= gi) ? S[(size_t)gi * n_pad + gj] : 0.0;
            else if (lc == 0) tgt[r] = rhs[gi];
        }
    }
    if (first) {
#pragma unroll
        for (int q = 0; q < NSB; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gj = r0 + 16 * q + lc;
                const bool fj = (gmask >> (16 + (16 * q + lc) / 6)) & 1;
                if (!is_rhs) acc[q][r] = xform_first_flags(acc[q][r], row0 + lr + 4 * r, gj, (gmask >> (lr + 4 * r)) & 1, fj, mu);
                else if (lr + 4 * r == 0) acc[q][r] = fj ? 0.0 : acc[q][r] + g0[gj];
            }
    }
    const bool store_l = (wave == 0 && I == J) || (wave == 1 && is_rhs && I == 0);   // every slab / the rhs leaves exactly once
#pragma unroll
    for (int q = 0; q < NSB; q++) {
#pragma unroll
        for (int r = 0; r < 4; r++) St[(lr + 4 * r) * PL + lc] = acc[q][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double ar[4];
#pragma unroll
        for (int t = 0; t < 4; t++) ar[t] = St[lc * PL + 4 * lr + t];
        dg_acc_t x = {0.0, 0.0, 0.0, 0.0};
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[0], li[q].x, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[1], li[q].y, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2], li[q].z, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[3], li[q].w, x, 0, 0, 0);
        const double dinv = rcp_refined(dq[q]);
        if (wave == 0 && lr == 0) dinv_l[16 * q + lc] = dinv;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            Xl[(lr + 4 * r) * XL + 16 * q + lc] = -x[r];   // -X_q stays in LDS: A operand of the substitution below and of the final product
            if (store_l) {
                // NOT in place: other workgroups are still reading the block column and the right-hand side of tile s
                if (!is_rhs) Lp[((size_t)s * n_pad + row0 + lr + 4 * r) * NB + 16 * q + lc] = x[r] * dinv;
                else if (lr + 4 * r == 0) zf[r0 + 16 * q + lc] = x[r] * dinv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (q + 1 < NSB) {
            double ax[4];
#pragma unroll
            for (int t = 0; t < 4; t++) ax[t] = Xl[lc * XL + 16 * q + 4 * lr + t];
#pragma unroll
            for (int q2 = q + 1; q2 < NSB; q2++) {
                const double4 l4 = lb[q * NSB - q * (q + 1) / 2 + (q2 - q - 1)];
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[0], l4.x, acc[q2], 0, 0, 0);
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[1], l4.y, acc[q2], 0, 0, 0);
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[2], l4.z, acc[q2], 0, 0, 0);
                acc[q2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[3], l4.w, acc[q2], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // P = X_I D^-1 X_J^T: this wavefront's half of k (three 16-column blocks); both operands are read as "row lc, columns 4 lr + t",
    // the same permutation of k on both sides; (-X_I)(-X_J) = +
    const double *XI = Xl_all, *XJ = Xl_all + SBK * XL;
    dg_acc_t pacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int qq = 0; qq < NSB / 2; qq++) {
        const int q = (NSB / 2) * wave + qq;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int k = 16 * q + 4 * lr + t;
            pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(XI[lc * XL + k], XJ[lc * XL + k] * dinv_l[k], pacc, 0, 0, 0);
        }
    }
    if (wave == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) red[r * 64 + lane] = pacc[r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double pv = pacc[r] + red[r * 64 + lane];
        const int gi = i0 + lr + 4 * r, gj = j0 + lc;
        if (J < ns) {
            if (gj <= gi) {
                double v = tgt[r];
                if (first) v = xform_first_flags(v, gi, gj, (omask >> (lr + 4 * r)) & 1, (omask >> (16 + lc)) & 1, mu);
                S[(size_t)gi * n_pad + gj] = v - pv;
            }
        } else if (lc == 0) {   // rhs entries of slab I: b -= L_I y  (the rhs "slab" has y in its row 0 only)
            double v = tgt[r];
            if (first) v = ((omask >> (lr + 4 * r)) & 1) ? 0.0 : v + g0[gi];
            rhs[gi] = v - pv;
        }
    }
}

// L^T x = z for the tiles above the last one (k_ldl_diag has solved that).  One workgroup per tile column s, all resident at
// once, chained by flags in global memory: workgroup s accumulates w_s = z_s - sum_{t>s} L_ts^T x_t block by block as the
// x_t are published (block (t, s) is already in registers when the flag arrives), then one wavefront back-substitutes
// L_ss^T x_s = w_s column by column (v_readlane sweep, no barriers) and publishes x_s.  Every CU fetches only its own tile
// column, so the solve no longer pulls all of L through one CU.  flag[s] == epoch means "x_s of this launch is in memory";
// x travels as agent-scope atomics past the XCDs' L2s (hop_publish / hop_wait: no cache-maintenance fence on the chain).
// LDS (dynamic): Ls [NB][NB+2] | xs [NB] | w [NB] | part [10][NB]
__global__ void __launch_bounds__(1024) k_ldl_backsolve(const double *__restrict__ S, const double *__restrict__ rhs,
                                                        const double *__restrict__ Dfac, double *__restrict__ x, int n_pad, int nT,
                                                        int32_t *__restrict__ flag, int epoch, int32_t *__restrict__ err_flags,
                                                        const double *__restrict__ Lp, const double *__restrict__ zf, int fused_m) {
    constexpr int LD = NB + 2, G = 10, RPT = (NB + G - 1) / G;
    extern __shared__ __align__(16) double lds[];
    double *Ls = lds;
    double *xs = Ls + NB * LD;
    double *w = xs + NB;
    double *part = w + NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = nT - 2 - (int)blockIdx.x;   // the head of the chain is dispatched first
    const int r0 = s * NB;
    const int j = tid % NB, gq = tid / NB;    // G*NB = 960 threads in the block products
    const double *dd = Dfac + (size_t)s * NB * NB;
    // block column s of L and z_s: in place (k_ldl_trsm) or, for the stages the fused panel kernel handled, in Lp / zf
    const bool from_lp = (nT - 1 - s) <= fused_m;
    const double *lcol = from_lp ? Lp + (size_t)s * n_pad * NB : S + r0;
    const size_t lld = from_lp ? (size_t)NB : (size_t)n_pad;
    const double *zsrc = from_lp ? zf : rhs;
    // Fetch order = use order: block (nT-1, s) first -- its product only waits for x_{nT-1}, which the previous kernel left in
    // memory -- then L_ss, which is not needed before the final substitution and is parked in LDS after that first product.
    constexpr int NL = NB * NB / 1024;
    double a[RPT], vl[NL];
    auto fetch_block = [&](int t) {
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int i = gq + G * k;
            a[k] = (gq < G && i < NB) ? lcol[(size_t)(t * NB + i) * lld + j] : 0.0;
        }
    };
    fetch_block(nT - 1);
#pragma unroll
    for (int u = 0; u < NL; u++) {   // strictly lower part only; the diagonal (D) and above stay zero in LDS
        const int e = tid + 1024 * u, i = e / NB, jj = e - i * NB;
        vl[u] = (jj < i) ? dd[e] : 0.0;
    }
    double acc = 0.0;
    for (int t = nT - 1; t > s; t--) {
        if (t < nT - 1) {   // x_{nT-1} comes from the previous kernel; the others from the workgroup next door
            // bounded: a chain that cannot complete (it always can: a workgroup only waits for ones dispatched before it)
            // must end in an error flag on the host, never in a hung device
            if (tid == 0 && !hop_wait(flag + t, epoch)) atomicOr(err_flags, 4);
            __syncthreads();
        }
        if (tid < NB) xs[tid] = __hip_atomic_load(x + t * NB + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int i = gq + G * k;
            if (i < NB) acc = fma(a[k], xs[i], acc);
        }
        if (t - 1 > s) fetch_block(t - 1);   // block (t-1, s), before x_{t-1} is needed
        if (t == nT - 1) {
#pragma unroll
            for (int u = 0; u < NL; u++) {
                const int e = tid + 1024 * u, i = e / NB, jj = e - i * NB;
                Ls[i * LD + jj] = vl[u];
            }
        }
        __syncthreads();
    }
    if (gq < G) part[gq * NB + j] = acc;
    __syncthreads();
    if (tid < NB) {
        double v = zsrc[r0 + tid];
#pragma unroll
        for (int g = 0; g < G; g++) v -= part[g * NB + tid];
        w[tid] = v;
    }
    __syncthreads();
    if (wave == 0) {   // L_ss^T x = w: lane l keeps rows l and l + 64; row j of L against x_j, from the last column up
        const int i0 = lane, i1 = lane + 64;
        const bool h1 = i1 < NB;
        double b0 = w[i0], b1 = h1 ? w[i1] : 0.0;
        const int i1c = h1 ? i1 : 0;
#pragma unroll
        for (int jj = NB - 1; jj > 0; jj--) {
            const double v = jj < 64 ? b0 : b1;
            const int lo = __builtin_amdgcn_readlane(__double2loint(v), jj & 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), jj & 63);
            const double xj = __hiloint2double(hi, lo);
            b0 = fma(-Ls[jj * LD + i0], xj, b0);
            if (jj > 64) b1 = fma(-Ls[jj * LD + i1c], xj, b1);
        }
        __hip_atomic_store(x + r0 + i0, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h1) __hip_atomic_store(x + r0 + i1, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hop_publish(flag + s, epoch, lane == 0);
    }
}

// ------------------------------------------------------------------------------------------------
// fixed-order sums of the per-block / per-frame partials:
//   scal[0] = sum err_part[0..n_err)  (per-frame sums of pass A, or per-block sums of k_residual)   scal[1] = sum |delta_f|^2   scal[2] = sum delta_f.g_f
//   scal[5], scal[6] = the shared-parameter pieces |delta_s|^2 (identical on every rank; NOT all-reduced)
//   and delta_s.g0 (folded into scal[2] when fold_shared, because g0 is a per-rank partial sum)
// ------------------------------------------------------------------------------------------------
// publish the scalars and the error flags to the mapped host record; the sequence number goes last, system scope

// ------------------------------------------------------------------------------------------------
void launch_frame_inv(const DeviceProblem &P, int which, double mu, hipStream_t st, const double *mu_dev, double mu_scale) {
    if (P.F == 0) return;
    const DeviceProblem::Blocks &b = P.blk[which];
    { HookScope _h(P, KID_FRAME_INV); hipLaunchKernelGGL(k_frame_inv, dim3((P.F + 255) / 256), dim3(256), 0, st, b.V, b.gf, P.F, mu, P.frames_fixed, b.Vinv, b.hf, P.flags,
                                                         mu_dev, mu_scale); }
}

static ReduceArgs reduce_args(const DeviceProblem &P, int n_err, bool fold_shared, unsigned long long publish_seq, double *scal_out = nullptr) {
    ReduceArgs r;
    r.err_part = P.err_part; r.n_err = n_err; r.lin_part = P.lin_part; r.F = P.F; r.fold_shared = fold_shared ? 1 : 0;
    r.scal = scal_out ? scal_out : P.scal; r.flags = P.flags; r.host = P.host_result; r.publish_seq = publish_seq;
    r.md_U = r.md_V = nullptr; r.md_fixed = nullptr; r.md_n_pad = r.md_A = r.md_frames_fixed = 0;
    r.cg_iters = P.use_spcg ? P.spcg_iters : nullptr;
    return r;
}

// ride_seq != 0: one extra workgroup of the launch reduces the step's scalars (ride_n_err partial sums of r^2) and publishes
// them under that sequence number -- the speculative Schur complement and that reduction only depend on the passes before
// them, not on each other.  Returns false if the scalars did not ride (the caller launches k_reduce_scalars).
bool launch_schur(const DeviceProblem &P, int which, double sign, hipStream_t st, unsigned long long ride_seq, int ride_n_err, double *ride_scal,
                  bool panels_ready) {
    const DeviceProblem::Blocks &b = P.blk[which];
    if (P.n_smwork > 0) {   // many shared entities: dense panels + block-of-S-stationary MFMA kernel
        const size_t lds = (size_t)2 * SM_ROWS * SM_PS * sizeof(double);
        static size_t granted = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_schur_mfma), lds, granted);
        const ReduceArgs red = reduce_args(P, ride_n_err, false, ride_scal ? 0ull : ride_seq, ride_scal);
        const int extra = (ride_seq || ride_scal) ? 1 : 0;
        HookScope _h(P, KID_SCHUR);
        if (!panels_ready)   // the panels for this damping: W from memory, Y = W (V + mu I)^-1 (the scalars ride here when there is such a launch)
            hipLaunchKernelGGL(k_schur_fill, dim3((unsigned)(((int64_t)P.total_slots * 6 + (int64_t)P.F * 6 + 255) / 256) + extra), dim3(256), 0, st, P.slot_frame, P.slot_dense,
                               b.W, b.Vinv, b.gf, P.total_slots, P.F, P.Ad, P.Wd, P.Yd, extra, red);
        const int extra_m = panels_ready ? extra : 0;
        hipLaunchKernelGGL(k_schur_mfma, dim3(P.n_smwork + extra_m), dim3(512), lds, st, P.sm_ga, P.sm_gb, P.sm_fb, P.sm_fe, P.sm_frames, P.dense_ent, P.Wd, P.Yd, P.Ad, P.n_pad, sign,
                           b.S, b.rhs, extra_m, red);
        return extra != 0;
    }
    if (P.n_swork == 0) return false;
    if (P.deterministic) {   // one wavefront per work item, records instead of atomics, fixed-order sums; the scalars do not ride
        const size_t lds1 = ((size_t)P.A * 36 + 8 + 48) * sizeof(double);
        static size_t granted_d = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_schur<0, 1>), lds1, granted_d);
        HookScope _h(P, KID_SCHUR);
        hipLaunchKernelGGL((k_schur<0, 1>), dim3(P.n_swork), dim3(64), lds1, st, P.sw_ent, P.sw_begin, P.sw_end, P.pair_rec, P.fslot_ent, b.W, b.Vinv, b.hf, P.A, P.n_pad, sign,
                           b.S, b.rhs, 0, ReduceArgs(), P.sp_part, P.sp_off);
        hipLaunchKernelGGL(k_schur_reduce, dim3(P.A), dim3(256), 0, st, P.se_start, P.se_items, P.sp_off, P.sp_part, P.n_pad, sign, b.S, b.rhs);
        return false;
    }
    const size_t lds = ((size_t)P.A * 36 + 8 + 4 * 48) * sizeof(double);
    static size_t granted0 = 48 * 1024, granted3 = 48 * 1024;
    const ReduceArgs red = reduce_args(P, ride_n_err, false, ride_scal ? 0ull : ride_seq, ride_scal);
    const int extra = (ride_seq || ride_scal) ? 1 : 0;
    HookScope _h(P, KID_SCHUR);
    if (P.max_kf > 64) {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_schur<3>), lds, granted3);
        hipLaunchKernelGGL(k_schur<3>, dim3(P.n_swork + extra), dim3(256), lds, st, P.sw_ent, P.sw_begin, P.sw_end, P.pair_rec, P.fslot_ent, b.W, b.Vinv, b.hf,
                           P.A, P.n_pad, sign, b.S, b.rhs, extra, red);
    } else {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_schur<0>), lds, granted0);
        hipLaunchKernelGGL(k_schur<0>, dim3(P.n_swork + extra), dim3(256), lds, st, P.sw_ent, P.sw_begin, P.sw_end, P.pair_rec, P.fslot_ent, b.W, b.Vinv, b.hf,
                           P.A, P.n_pad, sign, b.S, b.rhs, extra, red);
    }
    return extra != 0;
}

BacksubArgs backsub_args(const DeviceProblem &P, int cur, int trial, int waves_per_block) {
    const DeviceProblem::Blocks &b = P.blk[cur];
    BacksubArgs a;
    a.fslot_start = P.fslot_start; a.fslot_ent = P.fslot_ent; a.W = b.W; a.Vinv = b.Vinv; a.gf = b.gf; a.g0 = b.g0;
    a.Wf = P.use_pcg ? b.Wf : nullptr;   // (allocated only where the solve itself reads the fp32 blocks: ba_capi.hip)
    a.delta_s = P.delta_s; a.zc = P.z[cur]; a.zt = P.z[trial]; a.A = P.A; a.F = P.F;
    a.n_frame_blocks = (P.F + waves_per_block - 1) / waves_per_block;
    a.lin_part = P.lin_part; a.ent_out = P.ent[trial]; a.k_ent0 = P.intr ? P.C + P.M : P.A;
    return a;
}

// trial >= 0: the frame back-substitution z[trial] = z[which] + delta may ride in the last tile's launch (systems of up to three
// tiles); returns true if it did (the caller then skips launch_backsub)
bool launch_chol(const DeviceProblem &P, int which, double mu, hipStream_t st, int trial) {
    const DeviceProblem::Blocks &b = P.blk[which];
    const size_t lds_diag = ((size_t)NB * (NB + 2) + 2 * NB * DG_YS + (size_t)NB * (NB / 2 + 2)) * sizeof(double);   // tile | panel | look-ahead half panel
    static size_t g_diag = 48 * 1024, g_bs = 48 * 1024;
    // block columns with at most this many tiles below the diagonal take the fused panel kernel (0: never); its redundancy grows
    // with the square of the column's height, the two-kernel path's fixed cost does not  (3: +2.6 % on a four-tile system, nothing at 14 tiles)
    const int fused_m = P.tune.fused_panel;
    const bool bs_rides = P.tune.bs_rides != 0;
    // opt-in (AAR_BACKSUB_RIDES=1): measured on one box, it buys nothing -- 7 224 vs 7 233 LM it/s at config 3, 17 480 vs 17 780 at config 2
    // (profiles/r03_attempts.txt): the launch then ends with the riders' work instead of a 6 us kernel, and their 1024-thread workgroups
    // cost at dispatch what the kernel boundary did
    const bool backsub_rides = P.tune.backsub_rides != 0;
    // Look-ahead: where trsm and update are launches of their own (tall block columns), the update is not launched: the next diagonal tile's
    // workgroup applies it to its own tile and starts factoring, riders of that launch do the other tiles (AAR_LDL_LOOKAHEAD=0: off)
    const bool lookahead = P.tune.lookahead != 0;
    allow_dynamic_lds(reinterpret_cast<const void *>(k_ldl_diag), lds_diag, g_diag);
    bool rode_backsub = false;
    int upd_s = -1;   // the block column whose update the next k_ldl_diag launch carries
    for (int s = 0; s < P.nT; s++) {
        const int m = P.nT - s - 1;
        const bool last = s == P.nT - 1;
        const bool ride = bs_rides && last && (P.nT == 2 || P.nT == 3);   // the back-substitution rides in the last tile's launch
        // ... and so does the frame back-substitution when the whole of delta_s comes out of that launch (one tile, or the rider above);
        // not under per-kernel profiling (the riders would be billed to k_ldl_diag)
        const bool ride2 = backsub_rides && last && trial >= 0 && (P.nT == 1 || ride) && !P.hook.pre;
        BacksubArgs ba = backsub_args(P, which, trial >= 0 ? trial : which, DG_THREADS / 64);
        if (ride || ride2) P.bs_epoch++;
        const int grid = 1 + (ride ? 1 : 0) + (ride2 ? ba.n_frame_blocks + 1 : 0) + (upd_s >= 0 ? ldl_update_riders(P.nT - upd_s - 1) : 0);
        { HookScope _h(P, KID_LDL_DIAG); hipLaunchKernelGGL(k_ldl_diag, dim3(grid), dim3(DG_THREADS), lds_diag, st, b.S, P.Dfac, P.Linv16, P.n_pad, P.n, s, mu, P.ent_fixed, P.flags,
                                                         P.nT, b.rhs, b.g0, P.delta_s, P.Lp, P.zf, fused_m, P.bs_flags, P.bs_epoch, ride ? 1 : 0, ride2 ? 1 : 0, ba,
                                                         upd_s, b.S, b.rhs); }
        upd_s = -1;
        rode_backsub = ride2;
        if (m > 0 && m <= fused_m) {   // short block column: panel solve and trailing update in one launch
            const int ns = NSB * m;
            HookScope _h(P, KID_LDL_PANEL);
            hipLaunchKernelGGL(k_ldl_panel, dim3(ns * (ns + 1) / 2 + ns), dim3(128), 0, st, b.S, b.rhs, b.g0, P.Dfac, P.Linv16, P.n_pad, P.n, s, P.nT, mu, P.ent_fixed, P.Lp, P.zf);
        } else if (m > 0) {   // the last tile's right-hand side is solved inside k_ldl_diag
            { HookScope _h(P, KID_LDL_TRSM); hipLaunchKernelGGL(k_ldl_trsm, dim3(NSB * m + 1), dim3(64), 0, st, b.S, b.rhs, b.g0, P.Dfac, P.Linv16, P.n_pad, P.n, s, P.nT, mu, P.ent_fixed); }
            if (lookahead && m >= 2) upd_s = s;   // (m >= 2: the next tile is not the last one, whose launch has riders of its own)
            else { HookScope _h(P, KID_LDL_UPDATE); hipLaunchKernelGGL(k_ldl_update, dim3(m * (m + 1) / 2 * 9 + m), dim3(256), 0, st, b.S, b.rhs, b.g0, P.Dfac, P.n_pad, P.n, s, P.nT, mu, P.ent_fixed); }
        }
    }
    if (P.nT > 1 && !(bs_rides && P.nT <= 3)) {
        const size_t lds = ((size_t)NB * (NB + 2) + 2 * NB + 10 * NB) * sizeof(double);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_ldl_backsolve), lds, g_bs);
        P.bs_epoch++;
        HookScope _h(P, KID_LDL_BACKSOLVE);
        hipLaunchKernelGGL(k_ldl_backsolve, dim3(P.nT - 1), dim3(1024), lds, st, b.S, b.rhs, P.Dfac, P.delta_s, P.n_pad, P.nT, P.bs_flags, P.bs_epoch, P.flags, P.Lp, P.zf,
                           fused_m);
    }
    return rode_backsub;
}

void launch_backsub(const DeviceProblem &P, int cur, int trial, hipStream_t st) {
    const BacksubArgs ba = backsub_args(P, cur, trial, 4);
    { HookScope _h(P, KID_BACKSUB); hipLaunchKernelGGL(k_backsub, dim3(ba.n_frame_blocks + 1), dim3(256), 0, st, ba); }
}

void launch_reduce_scalars(const DeviceProblem &P, int n_err, bool fold_shared, unsigned long long publish_seq, hipStream_t st, double *scal_out, int maxdiag_blk) {
    ReduceArgs r = reduce_args(P, n_err, fold_shared, publish_seq, scal_out);
    if (maxdiag_blk >= 0) {
        r.md_U = P.blk[maxdiag_blk].S; r.md_V = P.blk[maxdiag_blk].V; r.md_fixed = P.ent_fixed; r.md_n_pad = P.n_pad; r.md_A = P.A; r.md_frames_fixed = P.frames_fixed;
    }
    { HookScope _h(P, KID_REDUCE); hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(256), 0, st, r); }
}

void launch_publish(const DeviceProblem &P, unsigned long long publish_seq, hipStream_t st, const double *src, bool flags_reduced) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, src ? src : P.scal, P.flags, P.host_result, publish_seq, flags_reduced ? 1 : 0);
}

}  // namespace aar
