// HIP kernels (gfx950, wave64) for the observation passes of one LM iteration:
//   k_unpack    z -> per-entity {R, t, J_l}            (eVec2Mats, libs/multicam_mapper.cpp:595-606)
//   k_residual  sum of squared residuals of a trial point (error_function, :731-737, :996-1028)
//   k_passA     frame-owned blocks V_f, g_f, W_cf, W_mf   \  together: J^T J and B = -J^T r of
//   k_passB     shared blocks U_cc, U_mm, W_cm, g_c, g_m  /  libs/sparselevmarq.h:355-367 in block form
// The reference builds J by central differences and multiplies sparse matrices; here each observation's
// analytic 8x18 Jacobian lives only in registers and its 6x6 block products are reduced on chip.
// fp64 throughout.  No atomics on frame-owned data; shared blocks get one fp64 atomic per value per
// (camera, marker) chunk.
#include "geom.hpp"
#include "kernels.h"

namespace aar {

// ------------------------------------------------------------------------------------------------
// wave-level sum of NV per-lane values through LDS, result handed to `sink(i, total)` on one lane.
// scratch: 32*64 doubles per wave.  Values are processed 32 at a time; lane l sums half of the lanes of
// value l>>1 with a rotated start so that a ds_read_b64 wave-instruction touches every bank once.
// ------------------------------------------------------------------------------------------------
template <int NV, class Sink>
__device__ __forceinline__ void wave_sum_lds(const double (&vals)[NV], double *__restrict__ sc, int lane, Sink sink) {
#pragma unroll
    for (int base = 0; base < NV; base += 32) {
        const int cnt = (NV - base) < 32 ? (NV - base) : 32;
#pragma unroll
        for (int i = 0; i < 32; i++)
            if (i < cnt) sc[i * 64 + lane] = vals[base + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int vi = lane >> 1, half = lane & 1;
        double s = 0;
        if (vi < cnt) {
            const double *row = sc + vi * 64 + half * 32;
#pragma unroll 8
            for (int k = 0; k < 32; k++) s += row[(k + lane) & 31];
        }
        s += __shfl_xor(s, 1);
        if (half == 0 && vi < cnt) sink(base + vi, s);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void k_unpack(const double *__restrict__ z, double *__restrict__ ent, int n_ent, double *__restrict__ zero_a,
                         int64_t zero_a_n, double *__restrict__ zero_b, int64_t zero_b_n) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (gid < n_ent) make_ent_row(z + 6 * gid, ent + gid * ENT_STRIDE);
    for (int64_t i = gid; i < zero_a_n; i += stride) zero_a[i] = 0.0;
    for (int64_t i = gid; i < zero_b_n; i += stride) zero_b[i] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// one thread per observation; per-frame partial sums are not needed here, so per-block partials
__global__ void __launch_bounds__(256) k_residual(const ObsIdx *__restrict__ idx, const float *__restrict__ uv,
                                                  const double *__restrict__ ent, const double *__restrict__ Kmat,
                                                  int64_t N, int A, double h, int res_f32, double *__restrict__ r_out,
                                                  double *__restrict__ err_part) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double ss = 0;
    if (o < N) {
        const ObsIdx id = idx[o];
        const float4 uv0 = reinterpret_cast<const float4 *>(uv)[2 * o];
        const float4 uv1 = reinterpret_cast<const float4 *>(uv)[2 * o + 1];
        const float ou[8] = {uv0.x, uv0.y, uv0.z, uv0.w, uv1.x, uv1.y, uv1.z, uv1.w};
        EntRT ec, em, ef;
        load_ent_rt(ent, id.cam, ec);
        load_ent_rt(ent, id.marker, em);  // marker entity index is already offset by C on the host
        load_ent_rt(ent, A + id.frame, ef);
        double K[9];
#pragma unroll
        for (int i = 0; i < 9; i++) K[i] = Kmat[9 * id.cam + i];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, h, k, g);
            double rx, ry;
            corner_residual(ou[2 * k], ou[2 * k + 1], g.u, g.v, res_f32, rx, ry);
            ss += rx * rx + ry * ry;
            if (r_out) {
                r_out[8 * o + 2 * k] = rx;
                r_out[8 * o + 2 * k + 1] = ry;
            }
        }
    }
    // block reduction: wave shuffle then LDS
    __shared__ double wsum[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) err_part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ------------------------------------------------------------------------------------------------
// Pass A: one workgroup per frame.  Threads stride over the frame's observations.  Frame-owned
// accumulators: V_f (symmetric 6x6), g_f, sum r^2 in registers -> wave sum -> LDS; W_cf / W_mf blocks
// accumulate in LDS (ds_add_f64) at the frame-local slot of the camera / marker and leave as one
// coalesced copy.  Nothing here is shared with another workgroup, so there are no global atomics.
// LDS: [max_kf*36] W blocks | [28] V,g,err | [nwaves*2048] wave-sum scratch
// ------------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_passA(const ObsIdx *__restrict__ idx, const float *__restrict__ uv,
                                                 const double *__restrict__ ent, const double *__restrict__ Kmat,
                                                 const int32_t *__restrict__ frame_obs_start,
                                                 const int32_t *__restrict__ fslot_start, int A, double h, int res_f32,
                                                 int max_kf, double *__restrict__ Vout, double *__restrict__ gout,
                                                 double *__restrict__ Wout, double *__restrict__ err_part) {
    extern __shared__ double lds[];
    double *Wl = lds;                       // [kf][36]
    double *acc = lds + (size_t)max_kf * 36;  // [28]
    double *scratch = acc + 32;             // [BLOCK/64][2048]
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o0 = frame_obs_start[f], o1 = frame_obs_start[f + 1];
    const int s0 = fslot_start[f], kf = fslot_start[f + 1] - s0;
    for (int i = tid; i < kf * 36; i += BLOCK) Wl[i] = 0.0;
    if (tid < 32) acc[tid] = 0.0;
    __syncthreads();

    double vals[28];  // 21 V (packed lower), 6 g, 1 err
#pragma unroll
    for (int i = 0; i < 28; i++) vals[i] = 0.0;

    Ent ef;
    load_ent(ent, A + f, ef);
    const int nobs = o1 - o0;
    // Interleaved assignment: consecutive lanes take observations `stride` apart so that lanes of one wave
    // mostly hold different cameras (the order inside a frame is camera-major); this keeps the ds_add_f64
    // same-address conflicts on W_cf low.  stride is coprime with nobs -> a permutation of the frame.
    int stride = 1;
    if (nobs > 16) {
        stride = nobs / 8 + 1;
        while (true) {  // gcd(stride, nobs) == 1
            int a = stride, b = nobs;
            while (b) { int t = a % b; a = b; b = t; }
            if (a == 1) break;
            stride++;
        }
    }
    for (int it = tid; it < nobs; it += BLOCK) {
        const int o = o0 + (int)(((int64_t)it * stride) % nobs);
        const ObsIdx id = idx[o];
        const float4 uv0 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o];
        const float4 uv1 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o + 1];
        const float ou[8] = {uv0.x, uv0.y, uv0.z, uv0.w, uv1.x, uv1.y, uv1.z, uv1.w};
        Ent ec, em;
        load_ent(ent, id.cam, ec);
        load_ent(ent, id.marker, em);
        double K[9];
#pragma unroll
        for (int i = 0; i < 9; i++) K[i] = Kmat[9 * id.cam + i];
        double Wc[36], Wm[36];
#pragma unroll
        for (int i = 0; i < 36; i++) { Wc[i] = 0.0; Wm[i] = 0.0; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, h, k, g);
            double r[2];
            corner_residual(ou[2 * k], ou[2 * k + 1], g.u, g.v, res_f32, r[0], r[1]);
            double Gc[2][6], Gm[2][6], Gf[2][6];
            corner_jacobian<true, true, true>(ec, em, ef, K, g, Gc, Gm, Gf);
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                vals[27] += r[rr] * r[rr];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    vals[21 + i] += Gf[rr][i] * r[rr];
#pragma unroll
                    for (int j = 0; j <= i; j++) vals[i * (i + 1) / 2 + j] += Gf[rr][i] * Gf[rr][j];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        Wc[i * 6 + j] += Gc[rr][i] * Gf[rr][j];
                        Wm[i * 6 + j] += Gm[rr][i] * Gf[rr][j];
                    }
                }
            }
        }
        double *wc = Wl + (id.slots & 0xffff) * 36;
        double *wm = Wl + ((id.slots >> 16) & 0xffff) * 36;
#pragma unroll
        for (int i = 0; i < 36; i++) atomicAdd(wc + i, Wc[i]);
#pragma unroll
        for (int i = 0; i < 36; i++) atomicAdd(wm + i, Wm[i]);
    }
    // V, g, err: wave sums, then one LDS add per wave and value
    wave_sum_lds<28>(vals, scratch + wave * 2048, lane, [&](int i, double s) { atomicAdd(acc + i, s); });
    __syncthreads();
    // coalesced write-out
    for (int i = tid; i < kf * 36; i += BLOCK) Wout[(size_t)s0 * 36 + i] = Wl[i];
    if (tid < 36) {
        const int i = tid / 6, j = tid % 6;
        Vout[(size_t)f * 36 + tid] = acc[sym6(i, j)];
    }
    if (tid >= 36 && tid < 42) gout[(size_t)f * 6 + (tid - 36)] = acc[21 + (tid - 36)];
    if (tid == 0) err_part[f] = acc[27];
}

// ------------------------------------------------------------------------------------------------
// Pass B: observations sorted by (camera, marker, frame); one wavefront per chunk of <= PASSB_CHUNK
// observations of a single (camera, marker) pair.  Each lane accumulates the 90 values of
// U_cc (21), U_mm (21), W_cm (36), g_c (6), g_m (6) over its observations in registers; one LDS wave sum;
// one fp64 atomic per value into the dense shared system (lower triangle, row-major, cameras first).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_passB(const ObsIdx *__restrict__ idx, const float *__restrict__ uv,
                                               const double *__restrict__ ent, const double *__restrict__ Kmat,
                                               const int32_t *__restrict__ chunk_start, int n_chunks, int A, double h,
                                               int res_f32, int n_pad, double *__restrict__ U0,
                                               double *__restrict__ g0) {
    __shared__ double scratch[4 * 2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x * 4 + wave;
    if (chunk >= n_chunks) return;
    const int o0 = chunk_start[chunk], o1 = chunk_start[chunk + 1];
    double vals[90];
#pragma unroll
    for (int i = 0; i < 90; i++) vals[i] = 0.0;
    const ObsIdx head = idx[o0];
    Ent ec, em;
    load_ent(ent, head.cam, ec);
    load_ent(ent, head.marker, em);
    double K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = Kmat[9 * head.cam + i];
    for (int o = o0 + lane; o < o1; o += 64) {
        const ObsIdx id = idx[o];
        const float4 uv0 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o];
        const float4 uv1 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o + 1];
        const float ou[8] = {uv0.x, uv0.y, uv0.z, uv0.w, uv1.x, uv1.y, uv1.z, uv1.w};
        Ent ef;
        load_ent(ent, A + id.frame, ef);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, h, k, g);
            double r[2];
            corner_residual(ou[2 * k], ou[2 * k + 1], g.u, g.v, res_f32, r[0], r[1]);
            double Gc[2][6], Gm[2][6], Gf[2][6];
            corner_jacobian<true, true, false>(ec, em, ef, K, g, Gc, Gm, Gf);
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    vals[78 + i] += Gc[rr][i] * r[rr];
                    vals[84 + i] += Gm[rr][i] * r[rr];
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        vals[i * (i + 1) / 2 + j] += Gc[rr][i] * Gc[rr][j];
                        vals[21 + i * (i + 1) / 2 + j] += Gm[rr][i] * Gm[rr][j];
                    }
#pragma unroll
                    for (int j = 0; j < 6; j++) vals[42 + i * 6 + j] += Gc[rr][i] * Gm[rr][j];
                }
            }
        }
    }
    const int rc = 6 * head.cam, rm = 6 * head.marker;  // first row of the camera / marker block
    wave_sum_lds<90>(vals, scratch + wave * 2048, lane, [&](int v, double s) {
        if (v < 42) {  // U_cc / U_mm, packed lower (i >= j)
            const int base = v < 21 ? rc : rm, p = v < 21 ? v : v - 21;
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= p) i++;
            const int j = p - i * (i + 1) / 2;
            atomicAdd(U0 + (size_t)(base + i) * n_pad + base + j, s);
        } else if (v < 78) {  // W_cm[i][j] -> row of the marker (below the cameras), column of the camera
            const int i = (v - 42) / 6, j = (v - 42) % 6;
            atomicAdd(U0 + (size_t)(rm + j) * n_pad + rc + i, s);
        } else if (v < 84) {
            atomicAdd(g0 + rc + (v - 78), s);
        } else {
            atomicAdd(g0 + rm + (v - 84), s);
        }
    });
}

// ------------------------------------------------------------------------------------------------
// max over the diagonal of J^T J restricted to free parameters (mu_0 = tau * max, libs/sparselevmarq.h:369-377)
__global__ void __launch_bounds__(256) k_maxdiag(const double *__restrict__ U0, int n_pad, int A,
                                                 const int32_t *__restrict__ ent_fixed, const double *__restrict__ V,
                                                 int F, int frames_fixed, double *__restrict__ out) {
    double m = -1.7976931348623157e308;
    for (int i = threadIdx.x; i < 6 * A; i += blockDim.x)
        if (!ent_fixed[i / 6]) m = fmax(m, U0[(size_t)i * n_pad + i]);
    if (!frames_fixed)
        for (int i = threadIdx.x; i < 6 * F; i += blockDim.x) m = fmax(m, V[(size_t)(i / 6) * 36 + (i % 6) * 7]);
    __shared__ double wm[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmax(fmax(wm[0], wm[1]), fmax(wm[2], wm[3]));
}

// ------------------------------------------------------------------------------------------------
void launch_unpack(const DeviceProblem &P, int which, bool zero_shared, hipStream_t st) {
    const int n_ent = P.A + P.F;
    const int64_t za = zero_shared ? (int64_t)P.n_pad * P.n_pad : 0, zb = zero_shared ? P.n_pad : 0;
    int64_t work = n_ent > za ? n_ent : za;
    int blocks = (int)((work + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < (n_ent + 255) / 256) blocks = (n_ent + 255) / 256;
    { HookScope _h(P, KID_UNPACK); hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(256), 0, st, P.z[which], P.ent[which], n_ent, P.U0, za, P.g0, zb); }
}

void launch_residual(const DeviceProblem &P, int which, double *r_out, hipStream_t st) {
    const int blocks = (int)((P.N + 255) / 256);
    if (blocks == 0) return;
    { HookScope _h(P, KID_RESIDUAL); hipLaunchKernelGGL(k_residual, dim3(blocks), dim3(256), 0, st, P.a_idx, P.a_uv, P.ent[which], P.K, P.N, P.A,
                       P.half_size, P.res_f32, r_out, P.err_part); }
}

int residual_blocks(const DeviceProblem &P) { return (int)((P.N + 255) / 256); }

void launch_passA(const DeviceProblem &P, int which, hipStream_t st) {
    if (P.F == 0) return;
    const double avg = (double)P.N / (double)P.F;
    if (avg <= 96) {
        constexpr int B = 64;
        const size_t lds = ((size_t)P.max_kf * 36 + 32 + (B / 64) * 2048) * sizeof(double);
        static size_t granted = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passA<B>), lds, granted);
        { HookScope _h(P, KID_PASSA); hipLaunchKernelGGL(k_passA<B>, dim3(P.F), dim3(B), lds, st, P.a_idx, P.a_uv, P.ent[which], P.K,
                           P.frame_obs_start, P.fslot_start, P.A, P.half_size, P.res_f32, P.max_kf, P.V, P.gf, P.W,
                           P.err_part); }
    } else {
        constexpr int B = 256;
        const size_t lds = ((size_t)P.max_kf * 36 + 32 + (B / 64) * 2048) * sizeof(double);
        static size_t granted = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passA<B>), lds, granted);
        { HookScope _h(P, KID_PASSA); hipLaunchKernelGGL(k_passA<B>, dim3(P.F), dim3(B), lds, st, P.a_idx, P.a_uv, P.ent[which], P.K,
                           P.frame_obs_start, P.fslot_start, P.A, P.half_size, P.res_f32, P.max_kf, P.V, P.gf, P.W,
                           P.err_part); }
    }
}

void launch_passB(const DeviceProblem &P, int which, hipStream_t st) {
    if (P.n_chunks == 0) return;
    { HookScope _h(P, KID_PASSB); hipLaunchKernelGGL(k_passB, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, P.b_idx, P.b_uv, P.ent[which], P.K,
                       P.chunk_start, P.n_chunks, P.A, P.half_size, P.res_f32, P.n_pad, P.U0, P.g0); }
}

void launch_maxdiag(const DeviceProblem &P, hipStream_t st) {
    { HookScope _h(P, KID_MAXDIAG); hipLaunchKernelGGL(k_maxdiag, dim3(1), dim3(256), 0, st, P.U0, P.n_pad, P.A, P.ent_fixed, P.V, P.F,
                       P.frames_fixed, P.scal + 4); }
}

}  // namespace aar
