// HIP kernels (gfx950, wave64) for the damped normal-equation solve of one LM try.  The reference
// factors the whole sparse (J^T J + mu I) with Eigen::SimplicialLDLT (libs/sparselevmarq.h:384-400); the
// same elimination is done here in block form: per-frame 6x6 inverses (k_frame_inv), Schur complement
// onto the cameras+markers (k_schur), dense blocked LDL^T of the reduced system (k_ldl_*, which also applies
// the damping and the gauge rows on first touch), and
// back-substitution of the frame poses (k_backsub).  mu is added to EVERY diagonal entry, as in the
// reference (:387-392).  fp64 throughout.
#include "geom.hpp"
#include "kernels.h"

namespace aar {

// ------------------------------------------------------------------------------------------------
// (V_f + mu I)^-1 and h_f = (V_f + mu I)^-1 g_f, one thread per frame.  Only launched when pass A's prediction of the
// damping was wrong (first step, a rejected try, gain < 0.94): a mu retry needs no Jacobian pass.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_frame_inv(const double *__restrict__ V, const double *__restrict__ gf, int F,
                                                   double mu, int frames_fixed, double *__restrict__ Vinv,
                                                   double *__restrict__ hf, int32_t *__restrict__ flags) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
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
// Dense LDL^T of the reduced system, right-looking, tile NB = 96, lower triangle row-major.
//
// Damping and gauge are applied on the FIRST touch of every element (step 0 of the factorisation), not by
// a kernel of their own: S(i,j) -> S(i,j) + mu [i == j] on free rows; rows / columns of fixed entities (root
// camera, root marker, non-optimised groups) and of the padding -> identity with zero rhs, i.e. delta = 0.
//
// Step s = one k_ldl_panel + one k_ldl_update.  Panel: every row tile of block column s gets its own
// workgroup; each keeps the diagonal tile AND its own tile in registers (thread (ty,tx) owns rows ty+16p,
// columns tx+16q) and replays the diagonal tile's column eliminations, so no triangular solve is needed:
// per column ONE barrier, the column travels through a double-buffered LDS vector, the pivot reciprocal is
// computed by the pivot's owner.  Two extra "row tiles" ride along: the right-hand side (a 1-row tile; its
// output is z_s = D^-1 L^-1 b) and the identity (its output is M_s = L_ss^-T D_s^-1, which turns the backward
// substitution into matrix-vector products).
// ------------------------------------------------------------------------------------------------
constexpr int NB = CHOL_NB;

__device__ __forceinline__ double xform_first(double v, int gi, int gj, int n, double mu, const int32_t *__restrict__ ent_fixed) {
    const bool fi = gi >= n || ent_fixed[gi / 6], fj = gj >= n || ent_fixed[gj / 6];
    if (fi || fj) return gi == gj ? 1.0 : 0.0;
    return gi == gj ? v + mu : v;
}

__device__ __forceinline__ double rcp_refined(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(x, fma(-d, x, 1.0), x);
    x = fma(x, fma(-d, x, 1.0), x);
    return x;
}

// Workgroup roles of one panel launch (block column s, m = nT-s-1 row tiles below the diagonal):
//   0                 the diagonal tile itself            -> Dfac[s]
//   1 .. 2m           half row tiles (48 rows x 96 cols)   -> L_ts in place
//   2m+1              the right-hand side (one row)        -> z_s
//   2m+2, 2m+3        the two halves of the identity       -> Minv[s]
__global__ void __launch_bounds__(256) k_ldl_panel(double *__restrict__ S, double *__restrict__ rhs, const double *__restrict__ g0,
                                                   double *__restrict__ Dfac, double *__restrict__ Minv, int n_pad, int n, int s,
                                                   int nT, double mu, const int32_t *__restrict__ ent_fixed,
                                                   int32_t *__restrict__ flags) {
    constexpr int R = NB / 16, RT = R / 2, HALF = NB / 2;
    __shared__ double colD[2][NB], colT[2][HALF], pinv[2], dinv[NB];
    const int m = nT - s - 1, b = blockIdx.x, tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const int r0 = s * NB;
    const bool first = (s == 0);
    int kind, row0 = 0;  // row0: first global row (tiles) or first identity row (identity halves)
    if (b == 0) kind = 0;
    else if (b <= 2 * m) { kind = 1; row0 = (s + 1) * NB + (b - 1) * HALF; }
    else if (b == 2 * m + 1) kind = 2;
    else { kind = 3; row0 = (b - 2 * m - 2) * HALF; }
    double D[R][R], T[RT][R];
#pragma unroll
    for (int p = 0; p < R; p++)
#pragma unroll
        for (int q = 0; q < R; q++) {
            if (q > p) { D[p][q] = 0.0; continue; }  // strictly above the diagonal blocks: never used
            const int i = ty + 16 * p, j = tx + 16 * q;
            double v = 0.0;
            if (j <= i) {
                v = S[(size_t)(r0 + i) * n_pad + r0 + j];
                if (first) v = xform_first(v, r0 + i, r0 + j, n, mu, ent_fixed);
            }
            D[p][q] = v;
        }
#pragma unroll
    for (int p = 0; p < RT; p++)
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int i = ty + 16 * p, j = tx + 16 * q;
            double t = 0.0;
            if (kind == 1) {
                t = S[(size_t)(row0 + i) * n_pad + r0 + j];
                if (first) t = xform_first(t, row0 + i, r0 + j, n, mu, ent_fixed);
            } else if (kind == 2) {
                if (i == 0) {
                    t = rhs[r0 + j];
                    if (first) t = (r0 + j >= n || ent_fixed[(r0 + j) / 6]) ? 0.0 : t + g0[r0 + j];  // B = g0 + Schur part
                }
            } else if (kind == 3) {
                t = (row0 + i == j) ? 1.0 : 0.0;
            }
            T[p][q] = t;
        }
#pragma unroll
    for (int kq = 0; kq < R; kq++) {
        for (int kk = 0; kk < 16; kk++) {
            const int k = 16 * kq + kk, buf = k & 1;
            if (tx == kk) {  // owners of column k publish it (final after step k-1)
#pragma unroll
                for (int p = 0; p < R; p++)
                    if (p >= kq) colD[buf][ty + 16 * p] = D[p][kq];
#pragma unroll
                for (int p = 0; p < RT; p++) colT[buf][ty + 16 * p] = T[p][kq];
                if (ty == kk) {  // pivot owner
                    const double d = D[kq][kq];
                    const double inv = rcp_refined(d);
                    pinv[buf] = inv;
                    dinv[k] = inv;
                    if (!(d > 0.0) && kind == 0) atomicOr(flags, 2);
                }
            }
            __syncthreads();
            const double inv = pinv[buf];
            double lj[R], li[R], ti[RT];
#pragma unroll
            for (int q = 0; q < R; q++)
                if (q >= kq) {
                    lj[q] = colD[buf][tx + 16 * q];
                    li[q] = colD[buf][ty + 16 * q] * inv;
                }
#pragma unroll
            for (int p = 0; p < RT; p++) ti[p] = colT[buf][ty + 16 * p] * inv;
#pragma unroll
            for (int q = 0; q < R; q++) {
                if (q < kq) continue;  // finished columns (compile-time)
                const int j = tx + 16 * q;
                const bool jact = (q > kq) || (j > k);
#pragma unroll
                for (int p = 0; p < R; p++) {
                    if (p < q) continue;  // lower block triangle only (compile-time)
                    const int i = ty + 16 * p;
                    const bool act = jact && ((p > kq) || (i > k)) && ((p > q) || (j <= i));
                    if (act) D[p][q] = fma(-li[p], lj[q], D[p][q]);
                }
#pragma unroll
                for (int p = 0; p < RT; p++)
                    if (jact) T[p][q] = fma(-ti[p], lj[q], T[p][q]);
            }
        }
    }
    __syncthreads();
    if (kind == 0) {
        // The factored diagonal tile goes to Dfac, NOT back into S: the other workgroups of this launch read
        // the unfactored tile from S, and nothing orders them against this store.
        double *out = Dfac + (size_t)s * NB * NB;
#pragma unroll
        for (int p = 0; p < R; p++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = ty + 16 * p, j = tx + 16 * q;
                out[i * NB + j] = (j < i) ? D[p][q] * dinv[j] : (j == i ? D[p][q] : 0.0);
            }
    } else if (kind == 1) {
#pragma unroll
        for (int p = 0; p < RT; p++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = ty + 16 * p, j = tx + 16 * q;
                S[(size_t)(row0 + i) * n_pad + r0 + j] = T[p][q] * dinv[j];
            }
    } else if (kind == 2) {
        if (ty == 0) {
#pragma unroll
            for (int q = 0; q < R; q++) rhs[r0 + tx + 16 * q] = T[0][q] * dinv[tx + 16 * q];
        }
    } else {
        double *out = Minv + (size_t)s * NB * NB;
#pragma unroll
        for (int p = 0; p < RT; p++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = row0 + ty + 16 * p, j = tx + 16 * q;
                out[i * NB + j] = T[p][q] * dinv[j];
            }
    }
}

// trailing update of step s in 32x32 output sub-tiles: S(I, J) -= L_Is D_s L_Js^T; rhs rows: b_t -= L_ts D_s z_s.
// grid: [tiles (ti >= tj)] x 9 sub-tiles, then one workgroup per rhs row tile.  LDS: Li [32][NB+1], Ljd [32][NB+1]
__global__ void __launch_bounds__(256) k_ldl_update(double *__restrict__ S, double *__restrict__ rhs, const double *__restrict__ g0,
                                                    const double *__restrict__ Dfac, int n_pad, int n, int s, int nT,
                                                    double mu, const int32_t *__restrict__ ent_fixed) {
    constexpr int LD = NB + 1, SB = 32, NSUB = (NB / SB) * (NB / SB);
    __shared__ double Li[SB * LD], Lj[SB * LD];
    const int m = nT - s - 1, tid = threadIdx.x;
    const int ntile = m * (m + 1) / 2;
    const int r0 = s * NB;
    const bool first = (s == 0);
    const double *dd = Dfac + (size_t)s * NB * NB;
    if ((int)blockIdx.x >= ntile * NSUB) {  // rhs entries of row tile t
        const int t = s + 1 + ((int)blockIdx.x - ntile * NSUB);
        double *zs = Li;
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
    const int si = sub / (NB / SB), sj = sub % (NB / SB);
    if (ti == tj && sj > si) return;  // above the diagonal
    const int i0 = (s + 1 + ti) * NB + si * SB, j0 = (s + 1 + tj) * NB + sj * SB;
    for (int e = tid; e < SB * NB; e += 256) {
        const int i = e / NB, k = e - i * NB;
        Li[i * LD + k] = S[(size_t)(i0 + i) * n_pad + r0 + k];
        Lj[i * LD + k] = S[(size_t)(j0 + i) * n_pad + r0 + k] * dd[k * NB + k];  // L_Js * D_s
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll 8
    for (int k = 0; k < NB; k++) {
        const double a0 = Li[ty * LD + k], a1 = Li[(ty + 16) * LD + k];
        const double c0 = Lj[tx * LD + k], c1 = Lj[(tx + 16) * LD + k];
        acc[0][0] = fma(a0, c0, acc[0][0]); acc[0][1] = fma(a0, c1, acc[0][1]);
        acc[1][0] = fma(a1, c0, acc[1][0]); acc[1][1] = fma(a1, c1, acc[1][1]);
    }
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int gi = i0 + ty + 16 * p, gj = j0 + tx + 16 * q;
            if (gj <= gi) {
                double v = S[(size_t)gi * n_pad + gj];
                if (first) v = xform_first(v, gi, gj, n, mu, ent_fixed);
                S[(size_t)gi * n_pad + gj] = v - acc[p][q];
            }
        }
}

// L^T x = z from the last tile up, with the tile inverses: x_s = M_s (d_s o (z_s - sum_{t>s} L_ts^T x_t)).
// One workgroup of 1024 threads; LDS: xs [n_pad] | part [10][NB] | w [NB]
__global__ void __launch_bounds__(1024) k_ldl_backsolve(const double *__restrict__ S, const double *__restrict__ rhs,
                                                        const double *__restrict__ Dfac, const double *__restrict__ Minv,
                                                        double *__restrict__ x, int n_pad, int nT) {
    extern __shared__ double lds[];
    double *xs = lds;
    double *part = xs + n_pad;
    double *w = part + 10 * NB;
    const int tid = threadIdx.x;
    constexpr int G = 10;  // row groups: G*NB = 960 threads busy in the reductions
    const int j = tid % NB, gq = tid / NB;
    for (int s = nT - 1; s >= 0; s--) {
        const int r0 = s * NB;
        if (gq < G) {
            double acc = 0.0;
            for (int i = (s + 1) * NB + gq; i < n_pad; i += G) acc += S[(size_t)i * n_pad + r0 + j] * xs[i];
            part[gq * NB + j] = acc;
        }
        __syncthreads();
        if (tid < NB) {
            double a = rhs[r0 + tid];
#pragma unroll
            for (int g = 0; g < G; g++) a -= part[g * NB + tid];
            w[tid] = a * Dfac[(size_t)s * NB * NB + tid * NB + tid];
        }
        __syncthreads();
        if (gq < G) {  // x_s[j] = sum_c M[j][c] w[c], c >= j (M is upper triangular); group gq takes c = gq, gq+G, ...
            const double *Mr = Minv + (size_t)s * NB * NB + (size_t)j * NB;
            double acc = 0.0;
            for (int c = gq; c < NB; c += G) acc += Mr[c] * w[c];
            part[gq * NB + j] = acc;
        }
        __syncthreads();
        if (tid < NB) {
            double a = 0.0;
#pragma unroll
            for (int g = 0; g < G; g++) a += part[g * NB + tid];
            xs[r0 + tid] = a;
            x[r0 + tid] = a;
        }
        __syncthreads();
    }
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
//   scal[0] = sum err_part[0..n_err)  (per-frame sums of pass A, or per-block sums of k_residual)   scal[1] = sum |delta_f|^2   scal[2] = sum delta_f.g_f
//   scal[5], scal[6] = the shared-parameter pieces |delta_s|^2 (identical on every rank; NOT all-reduced)
//   and delta_s.g0 (folded into scal[2] when fold_shared, because g0 is a per-rank partial sum)
// ------------------------------------------------------------------------------------------------
// publish the scalars and the error flags to the mapped host record; the sequence number goes last, system scope
__device__ __forceinline__ void publish_host(const double *__restrict__ scal, const int32_t *__restrict__ flags,
                                             double *__restrict__ host, unsigned long long seq) {
#pragma unroll
    for (int i = 0; i < 8; i++) host[i] = scal[i];
    reinterpret_cast<long long *>(host)[8] = (long long)(flags[0] | flags[1] | flags[2] | flags[3]);
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host) + 9, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_publish(const double *__restrict__ scal, const int32_t *__restrict__ flags, double *__restrict__ host,
                          unsigned long long seq) {
    if (threadIdx.x == 0) publish_host(scal, flags, host, seq);
}

__global__ void __launch_bounds__(256) k_reduce_scalars(const double *__restrict__ err_part, int n_err,
                                                        const double *__restrict__ lin_part, int F, int fold_shared,
                                                        double *__restrict__ scal, const int32_t *__restrict__ flags,
                                                        double *__restrict__ host, unsigned long long publish_seq) {
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
        if (publish_seq) publish_host(scal, flags, host, publish_seq);
    }
}

// ------------------------------------------------------------------------------------------------
void launch_frame_inv(const DeviceProblem &P, int which, double mu, hipStream_t st) {
    if (P.F == 0) return;
    const DeviceProblem::Blocks &b = P.blk[which];
    { HookScope _h(P, KID_FRAME_INV); hipLaunchKernelGGL(k_frame_inv, dim3((P.F + 255) / 256), dim3(256), 0, st, b.V, b.gf, P.F, mu, P.frames_fixed, b.Vinv, b.hf, P.flags); }
}

void launch_schur(const DeviceProblem &P, int which, hipStream_t st) {
    if (P.n_swork == 0) return;
    const DeviceProblem::Blocks &b = P.blk[which];
    const size_t lds = ((size_t)P.A * 36 + 8 + 4 * 48) * sizeof(double);
    static size_t granted = 48 * 1024;
    allow_dynamic_lds(reinterpret_cast<const void *>(k_schur), lds, granted);
    { HookScope _h(P, KID_SCHUR); hipLaunchKernelGGL(k_schur, dim3(P.n_swork), dim3(256), lds, st, P.sw_ent, P.sw_begin, P.sw_end, P.pair_frame,
                       P.pair_slot, P.fslot_start, P.fslot_ent, b.W, b.Vinv, b.hf, P.A, P.n_pad, b.S, b.rhs); }
}

void launch_chol(const DeviceProblem &P, int which, double mu, hipStream_t st) {
    const DeviceProblem::Blocks &b = P.blk[which];
    for (int s = 0; s < P.nT; s++) {
        const int m = P.nT - s - 1;
        { HookScope _h(P, KID_LDL_PANEL); hipLaunchKernelGGL(k_ldl_panel, dim3(2 * m + 4), dim3(256), 0, st, b.S, b.rhs, b.g0, P.Dfac, P.Minv, P.n_pad, P.n, s, P.nT, mu, P.ent_fixed, P.flags); }
        if (m > 0) {
            { HookScope _h(P, KID_LDL_UPDATE); hipLaunchKernelGGL(k_ldl_update, dim3(m * (m + 1) / 2 * 9 + m), dim3(256), 0, st, b.S, b.rhs, b.g0, P.Dfac, P.n_pad, P.n, s, P.nT, mu, P.ent_fixed); }
        }
    }
    const size_t lds = ((size_t)P.n_pad + 10 * NB + NB) * sizeof(double);
    static size_t granted_bs = 48 * 1024;
    allow_dynamic_lds(reinterpret_cast<const void *>(k_ldl_backsolve), lds, granted_bs);
    { HookScope _h(P, KID_LDL_BACKSOLVE); hipLaunchKernelGGL(k_ldl_backsolve, dim3(1), dim3(1024), lds, st, b.S, b.rhs, P.Dfac, P.Minv, P.delta_s, P.n_pad, P.nT); }
}

void launch_backsub(const DeviceProblem &P, int cur, int trial, hipStream_t st) {
    const DeviceProblem::Blocks &b = P.blk[cur];
    const int nfb = (P.F + 3) / 4;
    { HookScope _h(P, KID_BACKSUB); hipLaunchKernelGGL(k_backsub, dim3(nfb + 1), dim3(256), 0, st, P.fslot_start, P.fslot_ent, b.W, b.Vinv, b.gf, b.g0,
                       P.delta_s, P.z[cur], P.z[trial], P.A, P.F, nfb, P.lin_part); }
}

void launch_reduce_scalars(const DeviceProblem &P, int n_err, bool fold_shared, unsigned long long publish_seq, hipStream_t st) {
    { HookScope _h(P, KID_REDUCE); hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(256), 0, st, P.err_part, n_err, P.lin_part, P.F,
                       fold_shared ? 1 : 0, P.scal, P.flags, P.host_result, publish_seq); }
}

void launch_publish(const DeviceProblem &P, unsigned long long publish_seq, hipStream_t st) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, P.scal, P.flags, P.host_result, publish_seq);
}

}  // namespace aar
