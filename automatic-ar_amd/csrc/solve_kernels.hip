// HIP kernels (gfx950, wave64) for the damped normal-equation solve of one LM try.  The reference
// factors the whole sparse (J^T J + mu I) with Eigen::SimplicialLDLT (libs/sparselevmarq.h:384-400); the
// same elimination is done here in block form: per-frame 6x6 inverses (k_frame_inv), Schur complement
// onto the cameras+markers (k_schur), dense blocked LDL^T of the reduced system (k_ldl_*), and
// back-substitution of the frame poses (k_backsub).  mu is added to EVERY diagonal entry, as in the
// reference (:387-392).  fp64 throughout.
#include "geom.hpp"
#include "kernels.h"

namespace aar {

// ------------------------------------------------------------------------------------------------
// (V_f + mu I)^-1 and h_f = (V_f + mu I)^-1 g_f, one thread per frame; the same launch refreshes the
// working copy S <- U0, rhs <- g0 (grid-stride), so a mu retry needs no Jacobian pass.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_frame_inv(const double *__restrict__ V, const double *__restrict__ gf, int F,
                                                   double mu, int frames_fixed, double *__restrict__ Vinv,
                                                   double *__restrict__ hf, const double *__restrict__ U0,
                                                   const double *__restrict__ g0, double *__restrict__ S,
                                                   double *__restrict__ rhs, int64_t nn, int n_pad,
                                                   int32_t *__restrict__ flags) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // 16-byte copies of the shared system
    const double2 *src = reinterpret_cast<const double2 *>(U0);
    double2 *dst = reinterpret_cast<double2 *>(S);
    for (int64_t i = gid; i < nn / 2; i += stride) dst[i] = src[i];
    for (int64_t i = gid; i < n_pad; i += stride) rhs[i] = g0[i];
    if (gid >= F) return;
    const int f = (int)gid;
    double a[6][6];
    if (frames_fixed) {
#pragma unroll
        for (int i = 0; i < 36; i++) Vinv[(size_t)f * 36 + i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) hf[(size_t)f * 6 + i] = 0.0;
        return;
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) a[i][j] = V[(size_t)f * 36 + i * 6 + j] + (i == j ? mu : 0.0);
    // Cholesky a = L L^T (lower), in place
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double d = a[k][k];
#pragma unroll
        for (int p = 0; p < k; p++) d -= a[k][p] * a[k][p];
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double l = sqrt(d), il = 1.0 / l;
        a[k][k] = l;
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            double s = a[i][k];
#pragma unroll
            for (int p = 0; p < k; p++) s -= a[i][p] * a[k][p];
            a[i][k] = s * il;
        }
    }
    if (bad) atomicOr(flags, 1);
    // Linv (lower): column by column
    double li[6][6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i < c) { li[i][c] = 0.0; continue; }
            double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int p = c; p < i; p++) s -= a[i][p] * li[p][c];
            li[i][c] = s / a[i][i];
        }
    }
    // Vinv = Linv^T Linv
    double g[6], hv[6];
#pragma unroll
    for (int i = 0; i < 6; i++) g[i] = gf[(size_t)f * 6 + i];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        hv[i] = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 6; p++)
                if (p >= i && p >= j) s += li[p][i] * li[p][j];
            Vinv[(size_t)f * 36 + i * 6 + j] = s;
            hv[i] += s * g[j];
        }
        hf[(size_t)f * 6 + i] = hv[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Schur complement, output-stationary: work item = (shared entity a, a range of the frames that see a).
// The workgroup keeps row panel [S(a, b)]_{b <= a} (6 x 6(a+1)) and the rhs rows of a in LDS, its four
// wavefronts walk the (a, f) pairs: Y = W_af (V_f+mu I)^-1, then for every entity b <= a seen in f:
// panel(b) += Y W_bf^T.  One pass of fp64 atomics subtracts the panel from S at the end; all per-frame
// traffic stays in L2/LDS.  Only the lower triangle of S is produced.
// LDS: [A*36] panel | [8] rhs rows | [4][48] per-wave Y scratch
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_schur(const int32_t *__restrict__ sw_ent, const int32_t *__restrict__ sw_begin,
                                               const int32_t *__restrict__ sw_end, const int32_t *__restrict__ pair_frame,
                                               const int32_t *__restrict__ pair_slot, const int32_t *__restrict__ fslot_start,
                                               const int32_t *__restrict__ fslot_ent, const double *__restrict__ W,
                                               const double *__restrict__ Vinv, const double *__restrict__ hf, int A,
                                               int n_pad, double *__restrict__ S, double *__restrict__ rhs) {
    extern __shared__ double lds[];
    double *panel = lds;
    double *pg = lds + (size_t)A * 36;
    double *ysc = pg + 8;
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = sw_ent[w], pb = sw_begin[w], pe = sw_end[w];
    for (int i = tid; i < (a + 1) * 36; i += 256) panel[i] = 0.0;
    if (tid < 8) pg[tid] = 0.0;
    __syncthreads();
    double *ys = ysc + wave * 48;
    for (int p = pb + wave; p < pe; p += 4) {
        const int f = pair_frame[p], sg = pair_slot[p];
        const int s0 = fslot_start[f], sa = sg - s0;
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
        for (int i = 0; i < 36; i++) Y[i] = ys[i];
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
    __syncthreads();
    for (int idx = tid; idx < (a + 1) * 36; idx += 256) {
        const double v = panel[idx];
        if (v != 0.0) {
            const int b = idx / 36, e = idx - b * 36, i = e / 6, j = e - i * 6;
            atomicAdd(S + (size_t)(6 * a + i) * n_pad + 6 * b + j, -v);
        }
    }
    if (tid < 6) atomicAdd(rhs + 6 * a + tid, -pg[tid]);
}

// ------------------------------------------------------------------------------------------------
// damping + gauge: S += mu I on free rows; rows/columns of fixed entities (root camera, root marker,
// non-optimised groups) and of the padding become identity with zero rhs, i.e. delta = 0 there.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_finalize(double *__restrict__ S, double *__restrict__ rhs, int n, int n_pad,
                                                  double mu, const int32_t *__restrict__ ent_fixed) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nn = (int64_t)n_pad * n_pad;
    for (int64_t e = gid; e < nn; e += stride) {
        const int i = (int)(e / n_pad), j = (int)(e - (int64_t)i * n_pad);
        if (j > i) continue;
        const bool fi = i >= n || ent_fixed[i / 6], fj = j >= n || ent_fixed[j / 6];
        if (fi || fj) S[e] = (i == j) ? 1.0 : 0.0;
        else if (i == j) S[e] += mu;
    }
    for (int64_t i = gid; i < n_pad; i += stride)
        if (i >= n || ent_fixed[i / 6]) rhs[i] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// Dense LDL^T, right-looking, tile NB = 48, lower triangle row-major in place (L below the diagonal, D on it).
// Step s = one k_ldl_panel (every row tile of block column s in its own workgroup, each re-deriving the
// diagonal tile's eliminations in LDS so that no triangular solve is needed; one barrier per column)
// followed by one k_ldl_update (trailing tiles).  The right-hand side rides along as one more row.
// ------------------------------------------------------------------------------------------------
constexpr int NB = CHOL_NB;
constexpr int NBP = CHOL_NB + 1;

__global__ void __launch_bounds__(256) k_ldl_panel(double *__restrict__ S, double *__restrict__ rhs,
                                                   double *__restrict__ Dfac, int n_pad, int s, int nT,
                                                   int32_t *__restrict__ flags) {
    __shared__ double Dg[NB][NBP];
    __shared__ double Tt[NB][NBP];
    __shared__ double bv[NB];
    const int nrt = nT - s, b = blockIdx.x, tid = threadIdx.x;
    const bool is_rhs = (b == nrt), has_tile = (b > 0 && b < nrt);
    const int r0 = s * NB, t0 = (s + b) * NB;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e - i * NB;
        Dg[i][j] = (j <= i) ? S[(size_t)(r0 + i) * n_pad + r0 + j] : 0.0;
        if (has_tile) Tt[i][j] = S[(size_t)(t0 + i) * n_pad + r0 + j];
    }
    if (is_rhs && tid < NB) bv[tid] = rhs[r0 + tid];
    const int ty = tid >> 4, tx = tid & 15;
    for (int k = 0; k < NB; k++) {
        __syncthreads();
        const double dk = Dg[k][k];
        const double inv = 1.0 / dk;
        double lj[3], li[3], ti[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            lj[q] = Dg[tx + 16 * q][k];
            li[q] = Dg[ty + 16 * q][k] * inv;
            ti[q] = has_tile ? Tt[ty + 16 * q][k] * inv : 0.0;
        }
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int i = ty + 16 * p;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int j = tx + 16 * q;
                if (j > k) {
                    if (i > k && j <= i) Dg[i][j] -= li[p] * lj[q];
                    if (has_tile) Tt[i][j] -= ti[p] * lj[q];
                }
            }
        }
        if (is_rhs && tid < NB && tid > k) bv[tid] -= Dg[tid][k] * inv * bv[k];
    }
    __syncthreads();
    if (b == 0) {
        // The factored diagonal tile goes to Dfac, NOT back into S: the other workgroups of this launch
        // still read the unfactored tile from S, and nothing orders them against this store.
        double *out = Dfac + (size_t)s * NB * NB;
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e - i * NB;
            out[e] = (j < i) ? Dg[i][j] / Dg[j][j] : (j == i ? Dg[i][i] : 0.0);
        }
        if (tid < NB && !(Dg[tid][tid] > 0.0)) atomicOr(flags, 2);
    } else if (has_tile) {
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e - i * NB;
            S[(size_t)(t0 + i) * n_pad + r0 + j] = Tt[i][j] / Dg[j][j];
        }
    } else if (tid < NB) {
        rhs[r0 + tid] = bv[tid];
    }
}

__global__ void __launch_bounds__(256) k_ldl_update(double *__restrict__ S, double *__restrict__ rhs,
                                                    const double *__restrict__ Dfac, int n_pad, int s, int nT) {
    __shared__ double Li[NB][NBP];
    __shared__ double Lj[NB][NBP];
    __shared__ double ys[NB];
    const int m = nT - s - 1, tid = threadIdx.x;
    const int ntile = m * (m + 1) / 2;
    const int r0 = s * NB;
    if ((int)blockIdx.x >= ntile) {  // rhs rows of tile t: b_t -= L_ts y_s
        const int t = s + 1 + ((int)blockIdx.x - ntile);
        if (tid < NB) ys[tid] = rhs[r0 + tid];
        __syncthreads();
        if (tid < NB) {
            const double *row = S + (size_t)(t * NB + tid) * n_pad + r0;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; k++) acc += row[k] * ys[k];
            rhs[t * NB + tid] -= acc;
        }
        return;
    }
    int ti = 0, rem = blockIdx.x;  // decode (ti >= tj) from the linear lower-triangular tile index
    while (rem > ti) { rem -= ti + 1; ti++; }
    const int tj = rem;
    const int i0 = (s + 1 + ti) * NB, j0 = (s + 1 + tj) * NB;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, k = e - i * NB;
        Li[i][k] = S[(size_t)(i0 + i) * n_pad + r0 + k];
        Lj[i][k] = S[(size_t)(j0 + i) * n_pad + r0 + k] * Dfac[(size_t)s * NB * NB + k * NB + k];  // L_js * D_s
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    double acc[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) acc[p][q] = 0.0;
#pragma unroll 4
    for (int k = 0; k < NB; k++) {
        double a[3], c[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { a[q] = Li[ty + 16 * q][k]; c[q] = Lj[tx + 16 * q][k]; }
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) acc[p][q] += a[p] * c[q];
    }
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int i = ty + 16 * p, j = tx + 16 * q;
            if (ti != tj || j <= i) S[(size_t)(i0 + i) * n_pad + j0 + j] -= acc[p][q];
        }
}

// z = D^-1 y, then L^T x = z from the last tile up.  One workgroup; the reduced system is small.
__global__ void __launch_bounds__(256) k_ldl_backsolve(const double *__restrict__ S, const double *__restrict__ rhs,
                                                       const double *__restrict__ Dfac, double *__restrict__ x,
                                                       int n_pad, int nT) {
    extern __shared__ double lds[];
    double *xs = lds;                  // [n_pad]
    double *Ls = xs + n_pad;           // [NB][NBP]
    double *part = Ls + NB * NBP;      // [5][NB]
    double *v = part + 5 * NB;         // [NB]
    const int tid = threadIdx.x;
    for (int i = tid; i < n_pad; i += 256) xs[i] = rhs[i] / Dfac[(size_t)(i / NB) * NB * NB + (i % NB) * (NB + 1)];
    __syncthreads();
    for (int s = nT - 1; s >= 0; s--) {
        const int r0 = s * NB;
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e - i * NB;
            Ls[i * NBP + j] = (j < i) ? Dfac[(size_t)s * NB * NB + e] : 0.0;
        }
        // v_j = z_j - sum_{i >= (s+1)NB} L[i][r0+j] x_i
        const int j = tid % NB, gq = tid / NB;  // 5 row groups use 240 threads
        if (gq < 5) {
            double acc = 0.0;
            for (int i = (s + 1) * NB + gq; i < n_pad; i += 5) acc += S[(size_t)i * n_pad + r0 + j] * xs[i];
            part[gq * NB + j] = acc;
        }
        __syncthreads();
        if (tid < NB) v[tid] = xs[r0 + tid] - (part[tid] + part[NB + tid] + part[2 * NB + tid] + part[3 * NB + tid] + part[4 * NB + tid]);
        for (int k = NB - 1; k >= 0; k--) {
            __syncthreads();
            if (tid < k) v[tid] -= Ls[k * NBP + tid] * v[k];
        }
        __syncthreads();
        if (tid < NB) xs[r0 + tid] = v[tid];
        __syncthreads();
    }
    for (int i = tid; i < n_pad; i += 256) x[i] = xs[i];
}

// ------------------------------------------------------------------------------------------------
// Frame back-substitution, one wavefront per frame: delta_f = (V_f+mu I)^-1 (g_f - sum_a W_af^T delta_a);
// z_trial = z_cur + delta; per-frame pieces of L = 0.5 delta^T (mu delta - B) (libs/sparselevmarq.h:406).
// The last workgroup updates the shared (camera / marker) parameters.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_backsub(const int32_t *__restrict__ fslot_start, const int32_t *__restrict__ fslot_ent,
                                                 const double *__restrict__ W, const double *__restrict__ Vinv,
                                                 const double *__restrict__ gf, const double *__restrict__ g0,
                                                 const double *__restrict__ delta_s, const double *__restrict__ zc,
                                                 double *__restrict__ zt, int A, int F, int n_frame_blocks,
                                                 double *__restrict__ lin_part) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x == n_frame_blocks) {  // shared part
        double d2 = 0.0, dg = 0.0;
        for (int i = tid; i < 6 * A; i += 256) {
            const double d = delta_s[i];
            zt[i] = zc[i] + d;
            d2 += d * d;
            dg += d * g0[i];
        }
        __shared__ double red[8];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { d2 += __shfl_xor(d2, off); dg += __shfl_xor(dg, off); }
        if (lane == 0) { red[wave] = d2; red[4 + wave] = dg; }
        __syncthreads();
        if (tid == 0) {
            lin_part[2 * (size_t)F] = red[0] + red[1] + red[2] + red[3];
            lin_part[2 * (size_t)F + 1] = red[4] + red[5] + red[6] + red[7];
        }
        return;
    }
    const int f = blockIdx.x * 4 + wave;
    if (f >= F) return;
    const int s0 = fslot_start[f], s1 = fslot_start[f + 1];
    double c[6] = {0, 0, 0, 0, 0, 0};
    for (int s = s0 + lane; s < s1; s += 64) {
        const int a = fslot_ent[s];
        const double2 *wb = reinterpret_cast<const double2 *>(W + (size_t)s * 36);
        double da[6];
#pragma unroll
        for (int i = 0; i < 6; i++) da[i] = delta_s[6 * a + i];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double2 w0 = wb[3 * i], w1 = wb[3 * i + 1], w2 = wb[3 * i + 2];
            c[0] += w0.x * da[i]; c[1] += w0.y * da[i]; c[2] += w1.x * da[i];
            c[3] += w1.y * da[i]; c[4] += w2.x * da[i]; c[5] += w2.y * da[i];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int i = 0; i < 6; i++) c[i] += __shfl_xor(c[i], off);
    double g[6], vv[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { g[i] = gf[(size_t)f * 6 + i]; vv[i] = g[i] - c[i]; }
    double d = 0.0;
    if (lane < 6) {
#pragma unroll
        for (int k = 0; k < 6; k++) d += Vinv[(size_t)f * 36 + lane * 6 + k] * vv[k];
        const size_t zi = (size_t)6 * (A + f) + lane;
        zt[zi] = zc[zi] + d;
    }
    double d2 = (lane < 6) ? d * d : 0.0;
    double dgv = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) dgv += (lane == i) ? d * g[i] : 0.0;
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { d2 += __shfl_xor(d2, off); dgv += __shfl_xor(dgv, off); }
    if (lane == 0) { lin_part[2 * (size_t)f] = d2; lin_part[2 * (size_t)f + 1] = dgv; }
}

// ------------------------------------------------------------------------------------------------
// fixed-order sums of the per-block / per-frame partials:
//   scal[0] = sum err_part[0..n_err)   scal[1] = sum |delta_f|^2   scal[2] = sum delta_f.g_f
//   scal[5], scal[6] = the shared-parameter pieces |delta_s|^2 (identical on every rank; NOT all-reduced)
//   and delta_s.g0 (folded into scal[2] when fold_shared, because g0 is a per-rank partial sum)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_reduce_scalars(const double *__restrict__ err_part, int n_err,
                                                        const double *__restrict__ lin_part, int F, int fold_shared,
                                                        double *__restrict__ scal) {
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    double e = 0.0, d2 = 0.0, dg = 0.0;
    for (int i = tid; i < n_err; i += 256) e += err_part[i];
    for (int i = tid; i < F; i += 256) { d2 += lin_part[2 * (size_t)i]; dg += lin_part[2 * (size_t)i + 1]; }
    red[0][tid] = e; red[1][tid] = d2; red[2][tid] = dg;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            red[0][tid] += red[0][tid + off];
            red[1][tid] += red[1][tid + off];
            red[2][tid] += red[2][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        scal[0] = red[0][0]; scal[1] = red[1][0];
        // multi-GPU: delta_s . g0 uses this rank's piece of the shared gradient, so it joins the rank sum
        scal[2] = red[2][0] + (fold_shared ? lin_part[2 * (size_t)F + 1] : 0.0);
        scal[5] = lin_part[2 * (size_t)F]; scal[6] = lin_part[2 * (size_t)F + 1];
    }
}

// ------------------------------------------------------------------------------------------------
void launch_frame_inv(const DeviceProblem &P, double mu, hipStream_t st) {
    const int64_t nn = (int64_t)P.n_pad * P.n_pad;
    int64_t work = nn / 2 > P.F ? nn / 2 : P.F;
    int blocks = (int)((work + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < (P.F + 255) / 256) blocks = (P.F + 255) / 256;
    if (blocks < 1) blocks = 1;
    { HookScope _h(P, KID_FRAME_INV); hipLaunchKernelGGL(k_frame_inv, dim3(blocks), dim3(256), 0, st, P.V, P.gf, P.F, mu, P.frames_fixed, P.Vinv, P.hf,
                       P.U0, P.g0, P.S, P.rhs, nn, P.n_pad, P.flags); }
}

void launch_schur(const DeviceProblem &P, hipStream_t st) {
    if (P.n_swork == 0) return;
    const size_t lds = ((size_t)P.A * 36 + 8 + 4 * 48) * sizeof(double);
    { HookScope _h(P, KID_SCHUR); hipLaunchKernelGGL(k_schur, dim3(P.n_swork), dim3(256), lds, st, P.sw_ent, P.sw_begin, P.sw_end, P.pair_frame,
                       P.pair_slot, P.fslot_start, P.fslot_ent, P.W, P.Vinv, P.hf, P.A, P.n_pad, P.S, P.rhs); }
}

void launch_finalize(const DeviceProblem &P, double mu, hipStream_t st) {
    const int64_t nn = (int64_t)P.n_pad * P.n_pad;
    int blocks = (int)((nn + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    { HookScope _h(P, KID_FINALIZE); hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(256), 0, st, P.S, P.rhs, P.n, P.n_pad, mu, P.ent_fixed); }
}

void launch_chol(const DeviceProblem &P, hipStream_t st) {
    for (int s = 0; s < P.nT; s++) {
        { HookScope _h(P, KID_LDL_PANEL); hipLaunchKernelGGL(k_ldl_panel, dim3(P.nT - s + 1), dim3(256), 0, st, P.S, P.rhs, P.Dfac, P.n_pad, s, P.nT, P.flags); }
        const int m = P.nT - s - 1;
        if (m > 0)
            { HookScope _h(P, KID_LDL_UPDATE); hipLaunchKernelGGL(k_ldl_update, dim3(m * (m + 1) / 2 + m), dim3(256), 0, st, P.S, P.rhs, P.Dfac, P.n_pad, s, P.nT); }
    }
    const size_t lds = ((size_t)P.n_pad + NB * NBP + 5 * NB + NB) * sizeof(double);
    { HookScope _h(P, KID_LDL_BACKSOLVE); hipLaunchKernelGGL(k_ldl_backsolve, dim3(1), dim3(256), lds, st, P.S, P.rhs, P.Dfac, P.delta_s, P.n_pad, P.nT); }
}

void launch_backsub(const DeviceProblem &P, int cur, int trial, hipStream_t st) {
    const int nfb = (P.F + 3) / 4;
    { HookScope _h(P, KID_BACKSUB); hipLaunchKernelGGL(k_backsub, dim3(nfb + 1), dim3(256), 0, st, P.fslot_start, P.fslot_ent, P.W, P.Vinv, P.gf, P.g0,
                       P.delta_s, P.z[cur], P.z[trial], P.A, P.F, nfb, P.lin_part); }
}

int residual_blocks(const DeviceProblem &P);

void launch_reduce_scalars(const DeviceProblem &P, bool fold_shared, hipStream_t st) {
    { HookScope _h(P, KID_REDUCE); hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(256), 0, st, P.err_part, residual_blocks(P), P.lin_part, P.F,
                       fold_shared ? 1 : 0, P.scal); }
}

}  // namespace aar
