// HIP kernels (gfx950, wave64) for the observation passes of one LM iteration:
//   k_passA     z -> per-entity {R, t, J_l} staged in LDS (eVec2Mats, libs/multicam_mapper.cpp:595-606), then the
//               frame-owned blocks V_f, g_f, W_cf, W_mf, the frame's sum r^2 (error_function, :731-737) and
//               (V_f + mu I)^-1 for the predicted damping        \  together: J^T J and B = -J^T r of
//   k_passB     shared blocks U_cc, U_mm, W_cm, g_c, g_m         /  libs/sparselevmarq.h:355-367 in block form
//   k_unpack, k_residual   residual rows only (aar_eval_residuals)
// The reference builds J by central differences and multiplies sparse matrices; here each observation's
// analytic 8x18 Jacobian lives only in registers and its 6x6 block products are reduced on chip.
// fp64 throughout.  No atomics on frame-owned data; shared blocks get one fp64 atomic per value per
// (camera, marker) chunk.
#include <algorithm>
#include <vector>
#include "geom.hpp"
#include "kernels.h"
#include "hostcopy.h"
#include "wave.hpp"

namespace aar {

volatile int g_last_kernel_id = -1;   // (kernels.h: diagnostics)

// ------------------------------------------------------------------------------------------------
// wave-level sum of NV per-lane values through LDS, result handed to `sink(i, total)` on one lane.
// scratch: CH*64 doubles per wave.  Values are processed CH at a time; 64/CH lanes share a value, each sums CH of its 64
// entries (start rotated so that a ds_read_b64 wave-instruction touches every bank equally often), then xor-shuffles.
// CH = 32 is the fewest rounds; CH = 8 costs 4 KB of LDS per wave instead of 16 (pass A's four-wave variant: two
// workgroups per CU instead of one).
// ------------------------------------------------------------------------------------------------
template <int NV, int CH = 32, class Sink>
__device__ __forceinline__ void wave_sum_lds(const double (&vals)[NV], double *__restrict__ sc, int lane, Sink sink) {
    constexpr int L = 64 / CH;        // lanes per value
    constexpr int SPREAD = 32 / CH > 0 ? 32 / CH : 1;   // lanes whose CH-entry segments cover the 32 double-wide banks once
#pragma unroll
    for (int base = 0; base < NV; base += CH) {
        const int cnt = (NV - base) < CH ? (NV - base) : CH;
#pragma unroll
        for (int i = 0; i < CH; i++)
            if (i < cnt) sc[i * 64 + lane] = vals[base + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int vi = lane / L, part = lane % L;
        double s = 0;
        if (vi < cnt) {
            const double *row = sc + lane * CH;   // = vi * 64 + part * CH
#pragma unroll 8
            for (int k = 0; k < CH; k++) s += row[(k + lane / SPREAD) & (CH - 1)];
        }
#pragma unroll
        for (int off = 1; off < L; off <<= 1) s += __shfl_xor(s, off);
        if (part == 0 && vi < cnt) sink(base + vi, s);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------
// zero_n > 0: the launch also clears zero_p[0 .. zero_n) grid-stride (aar_lm_init: the block set pass B is about to accumulate into and the
// linear-model partials -- one launch less at the start of every solve)
__global__ void k_unpack(const double *__restrict__ z, double *__restrict__ ent, int n_ent, int k0, int k1, double *__restrict__ zero_p, int64_t zero_n,
                         double *__restrict__ zero_q, int64_t zero_qn) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = gid; i < zero_n; i += (int64_t)gridDim.x * blockDim.x) zero_p[i] = 0.0;
    for (int64_t i = gid; i < zero_qn; i += (int64_t)gridDim.x * blockDim.x) zero_q[i] = 0.0;
    if (gid >= n_ent) return;
    if (gid >= k0 && gid < k1) make_k_row(z + 6 * gid, ent + gid * ENT_STRIDE);   // intrinsics entities [k0, k1)
    else make_ent_row(z + 6 * gid, ent + gid * ENT_STRIDE);
}

// ------------------------------------------------------------------------------------------------
// one thread per observation; per-frame partial sums are not needed here, so per-block partials
__global__ void __launch_bounds__(256) k_residual(const ObsIdx *__restrict__ idx, const float *__restrict__ uv,
                                                  const double *__restrict__ ent, const double *__restrict__ Kmat, int kstride,
                                                  int64_t N, int A, double h, int res_f32, float huber, double *__restrict__ r_out,
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
        for (int i = 0; i < 9; i++) K[i] = Kmat[kstride * id.cam + i];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, h, k, g);
            double rx, ry;
            corner_residual(ou[2 * k], ou[2 * k + 1], g.u, g.v, res_f32, huber, rx, ry);
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
__device__ __forceinline__ void load_ent_lds(const double *row, Ent &e) {
#pragma unroll
    for (int i = 0; i < 9; i++) e.R[i] = row[i];
#pragma unroll
    for (int i = 0; i < 3; i++) e.t[i] = row[9 + i];
#pragma unroll
    for (int i = 0; i < 9; i++) e.Jl[i] = row[12 + i];
}

// ------------------------------------------------------------------------------------------------
// Pass A: one workgroup per frame.  The SE(3) parameters of everything the frame touches (its own pose and the
// k_f cameras / markers of its slot list) are turned into {R, t, J_l} rows in LDS first; threads then stride over
// the frame's observations.  Frame-owned accumulators: V_f (symmetric 6x6), g_f, sum r^2 in registers -> wave sum ->
// LDS; W_cf / W_mf blocks accumulate in LDS (ds_add_f64) at the frame-local slot of the camera / marker and leave as
// one coalesced copy.  Nothing here is shared with another workgroup: no global atomics.  The epilogue inverts
// V_f + mu_pred I (the damping the NEXT solve is expected to use) and clears a dead block set grid-stride.
// LDS: [max_kf*37] W blocks | [64] V,g,err (+ H_f, sum w r of the wrench form) | [(max_kf+1)*25] entity rows | [nwaves * 64 * passA_sum_chunk] wave-sum scratch
// ------------------------------------------------------------------------------------------------
struct PassAArgs {
    const ObsIdx *idx; const float *uv; const double *ent; const double *Kmat;   // ent: the {R, t, J_l} table of the point (k_backsub / k_unpack)
    int kstride;                                                                 // Kmat[kstride * camera + i]
    const int32_t *frame_obs_start, *fslot_start, *fslot_ent, *frame_stride;
    unsigned long long *stamps;   // AAR_PASSA_STAMPS builds only
    int k_ent0;                   // first intrinsics entity (C + M); = A without them
    int A, F, C, res_f32, max_kf, frames_fixed;   // C: entities below it are cameras
    float huber;
    double h, mu_pred;
    double *V, *gf, *W, *Vinv, *hf, *err_part;
    float *Wf;                    // fp32 copy of W for k_pcgf (kernels.h, Blocks::Wf); nullptr: not wanted
    double *zero0; int64_t zero0_n; double *zero1; int64_t zero1_n; double *zero2; int64_t zero2_n;
    int32_t *flags;
    // MFMA Schur path: the dense per-frame panels Wd / Yd = W (V_f + mu_pred I)^-1 leave from HERE (the W blocks are still in LDS),
    // instead of a k_schur_fill launch that reads W back; null: not wanted (or no prediction of the damping)
    double *Wd, *Yd; const int32_t *slot_dense; int Ad;
};

// values per round of the V/g/err wave sum: the four-wave variant trades rounds for LDS (75 KB instead of 123 KB at 122
// slots per frame: a second workgroup per CU); the one-wave variants share their launch with pass B's 32-wide rounds
constexpr int ENT_LDS = 25;   // doubles between entity rows in pass A's LDS copy (24 used): odd, so that lanes reading one field of different entities hit different banks
constexpr int WLS = 37;   // doubles between the W blocks of consecutive frame-local slots in pass A's LDS panel
__host__ __device__ constexpr size_t passA_w_doubles(int max_kf) { return ((size_t)max_kf * WLS + 1) & ~(size_t)1; }   // what follows stays 16-byte aligned
__host__ __device__ constexpr int passA_sum_chunk(int block) { return block >= 128 ? 8 : 32; }

// what both forms of pass A end with (W blocks, V / g / err in LDS; after a __syncthreads): the coalesced copies, (V_f + mu_pred I)^-1 and h_f on one lane,
// the dense panels of the MFMA Schur path
template <int BLOCK>
__device__ __forceinline__ void passA_epilogue(const PassAArgs &a, const double *Wl, const double *acc, double *scratch, const int f, const int s0, const int kf) {
    const int tid = threadIdx.x;
    // coalesced write-out
    if (a.W) for (int i = tid; i < kf * 36; i += BLOCK) a.W[(size_t)s0 * 36 + i] = Wl[i + i / 36];
    if (a.Wf)   // (element r of slot j -> piece r / 4 of slot j)
        for (int i = tid; i < kf * 36; i += BLOCK) {
            const int j = i / 36, r = i - 36 * j;
            a.Wf[(size_t)s0 * 36 + ((size_t)(r >> 2) * kf + j) * 4 + (r & 3)] = (float)Wl[i + j];
        }
    for (int t = tid; t < 36; t += BLOCK) a.V[(size_t)f * 36 + t] = acc[sym6(t / 6, t % 6)];
    for (int t = tid; t < 6; t += BLOCK) a.gf[(size_t)f * 6 + t] = acc[21 + t];
    if (tid == 0) a.err_part[f] = acc[27];
    const bool dense = a.Yd != nullptr && a.mu_pred >= 0.0;
    if (a.mu_pred >= 0.0 && tid == BLOCK - 1) {  // (V_f + mu I)^-1 and h_f for the damping the next solve is expected to use
        double out[36];
        if (a.frames_fixed) {
#pragma unroll
            for (int i = 0; i < 36; i++) out[i] = 0.0;
        } else {
            double m[6][6];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) m[i][j] = acc[sym6(i, j)] + (i == j ? a.mu_pred : 0.0);
            if (!spd6_inverse(m, out)) atomicOr(a.flags, 1);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double hv = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                a.Vinv[(size_t)f * 36 + i * 6 + j] = out[i * 6 + j];
                hv += out[i * 6 + j] * acc[21 + j];
                if (dense) scratch[i * 6 + j] = out[i * 6 + j];   // (the wave-sum scratch is free by now)
            }
            a.hf[(size_t)f * 6 + i] = hv;
        }
    }
    if (dense) {   // row (slot, i) of the frame's panels: W as it is, Y = W (V_f + mu I)^-1; the pseudo entity 0 carries g_f in its row 0
        __syncthreads();
        double vi[36];
#pragma unroll
        for (int q = 0; q < 36; q++) vi[q] = scratch[q];
        const size_t fbase = (size_t)f * a.Ad * 36;
        for (int r = tid; r < kf * 6; r += BLOCK) {
            const int sl = r / 6, i = r - sl * 6;
            const double *wr = Wl + sl * WLS + i * 6;
            double w[6], y[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 6; k++) w[k] = wr[k];
#pragma unroll
            for (int k = 0; k < 6; k++)
#pragma unroll
                for (int j = 0; j < 6; j++) y[j] = fma(w[k], vi[k * 6 + j], y[j]);
            const size_t o = fbase + (size_t)a.slot_dense[s0 + sl] * 36 + i * 6;
            double2 *yp = reinterpret_cast<double2 *>(a.Yd + o), *wd = reinterpret_cast<double2 *>(a.Wd + o);
            yp[0] = make_double2(y[0], y[1]); yp[1] = make_double2(y[2], y[3]); yp[2] = make_double2(y[4], y[5]);
            wd[0] = make_double2(w[0], w[1]); wd[1] = make_double2(w[2], w[3]); wd[2] = make_double2(w[4], w[5]);
        }
        if (tid < 6) a.Wd[fbase + tid] = acc[21 + tid];
    }
}

// CPL = corners per lane: 4 = one lane per observation; 2 / 1 = two / four lanes per observation for frames with few
// observations (the wavefront's instruction stream gets that much shorter; the sums over lanes do not care)
// INTR: camera intrinsics are optimised -- every observation also feeds W_kf, the block of its camera's intrinsics entity
// (four live rows: fx, cx, fy, cy) against the frame, at that entity's frame-local slot
template <int BLOCK, int CPL, bool INTR = false>
__device__ __forceinline__ void passA_body(const PassAArgs &a, double *lds, const int f, const int n_blocks_a) {
    double *Wl = lds;                                   // [kf][WLS]: 36 values per slot, 37 apart (an odd stride spreads the slots over all banks)
    double *acc = lds + passA_w_doubles(a.max_kf);         // [32] (+ 32 the wrench form uses)
    double *entl = acc + 64;                            // [(max_kf+1)][ENT_LDS]
    constexpr int CH = passA_sum_chunk(BLOCK);
    double *scratch = entl + (size_t)(a.max_kf + 1) * ENT_LDS;  // [BLOCK/64][CH * 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o0 = a.frame_obs_start[f], o1 = a.frame_obs_start[f + 1];
    const int s0 = a.fslot_start[f], kf = a.fslot_start[f + 1] - s0;
    for (int i = tid; i < kf * WLS; i += BLOCK) Wl[i] = 0.0;
    for (int t = tid; t < 32; t += BLOCK) acc[t] = 0.0;
    // entity rows of this frame's slot list (+ the frame itself at row kf): global table -> LDS in 16-byte pieces, all the
    // slot-list loads of a batch in flight before the first row load, all row loads before the first LDS store
    {
        constexpr int PCS = ENT_STRIDE / 2, UNR = 8;
        const int npc = (kf + 1) * PCS;
        for (int base = 0; base < npc; base += UNR * BLOCK) {
            int e[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * BLOCK + tid, t = idx / PCS;
                e[u] = idx < npc ? (t < kf ? a.fslot_ent[s0 + t] : a.A + f) : -1;
            }
            double2 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * BLOCK + tid, c = idx % PCS;
                if (e[u] >= 0) v[u] = reinterpret_cast<const double2 *>(a.ent + (size_t)e[u] * ENT_STRIDE)[c];
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * BLOCK + tid, t = idx / PCS, c = idx - t * PCS;
                if (e[u] >= 0) { entl[t * ENT_LDS + 2 * c] = v[u].x; entl[t * ENT_LDS + 2 * c + 1] = v[u].y; }
            }
        }
    }
    // grid-stride clearing of the block set that is dead by now
    {
        const int64_t gid = (int64_t)f * BLOCK + tid, stride = (int64_t)n_blocks_a * BLOCK;
        for (int64_t i = gid; i < a.zero0_n; i += stride) a.zero0[i] = 0.0;
        for (int64_t i = gid; i < a.zero1_n; i += stride) a.zero1[i] = 0.0;
        for (int64_t i = gid; i < a.zero2_n; i += stride) a.zero2[i] = 0.0;
    }
    __syncthreads();

    double vals[28];  // 21 V (packed lower), 6 g, 1 err
#pragma unroll
    for (int i = 0; i < 28; i++) vals[i] = 0.0;

    // the frame's own row is the same for every lane of the workgroup: fetched through the scalar cache it lives in SGPRs, not in
    // 42 VGPRs per lane (k_passA_intr<256,4>: 81 spilled registers -> see profiles/r03_vgpr_counts.txt)
    Ent ef;
    load_ent(a.ent, a.A + __builtin_amdgcn_readfirstlane(f), ef);
    const int nobs = o1 - o0;
    // Interleaved assignment: consecutive lanes take observations `stride` apart so that lanes of one wave
    // mostly hold different cameras (the order inside a frame is camera-major); this keeps the ds_add_f64
    // same-address conflicts on W_cf low.  stride is coprime with nobs -> a permutation of the frame.
    int stride = 1;
    if (nobs > 16) {
        stride = nobs / 8 + 1;
        while (true) {  // gcd(stride, nobs) == 1
            int x = stride, y = nobs;
            while (y) { int t = x % y; x = y; y = t; }
            if (x == 1) break;
            stride++;
        }
    }
    constexpr int LPO = 4 / CPL;   // lanes per observation
    for (int t = tid; t < nobs * LPO; t += BLOCK) {
        const int it = t / LPO, part = t - it * LPO;
        const int o = o0 + (int)(((int64_t)it * stride) % nobs);
        const ObsIdx id = a.idx[o];
        const int sc = id.slots & ((1 << SLOT_C_BITS) - 1), sm = (id.slots >> SLOT_C_BITS) & ((1 << SLOT_M_BITS) - 1);
        const int sk = (id.slots >> (SLOT_C_BITS + SLOT_M_BITS)) & ((1 << SLOT_M_BITS) - 1);
        Ent ec, em;
        load_ent_lds(entl + (size_t)sc * ENT_LDS, ec);
        load_ent_lds(entl + (size_t)sm * ENT_LDS, em);
        double K[9];
#pragma unroll
        for (int i = 0; i < 9; i++) K[i] = a.Kmat[a.kstride * id.cam + i];
        double Wc[36], Wm[36], Wk[INTR ? 24 : 1];
#pragma unroll
        for (int i = 0; i < 36; i++) { Wc[i] = 0.0; Wm[i] = 0.0; }
        if (INTR) {
#pragma unroll
            for (int i = 0; i < 24; i++) Wk[i] = 0.0;
        }
#pragma unroll   // (NOT unrolling it for the intrinsics variant, whose 24 extra accumulators spill 80-90 registers, spills 130-180)
        for (int kk = 0; kk < CPL; kk++) {
            const int k = part * CPL + kk;
            const float2 ouv = reinterpret_cast<const float2 *>(a.uv)[4 * (int64_t)o + k];
            CornerGeom g;
            project_corner(ec, em, ef, K, a.h, k, g);
            double r[2];
            corner_residual(ouv.x, ouv.y, g.u, g.v, a.res_f32, a.huber, r[0], r[1]);
            double Gc[2][6], Gm[2][6], Gf[2][6];
            corner_jacobian<true, true, true>(ec, em, ef, K, g, Gc, Gm, Gf);
            if (INTR) {
                double Gk[2][4];
                corner_jacobian_intr(g, Gk);
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) Wk[i * 6 + j] += Gk[rr][i] * Gf[rr][j];
            }
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
        double *wc = Wl + sc * WLS;
        double *wm = Wl + sm * WLS;
#pragma unroll
        for (int i = 0; i < 36; i++) atomicAdd(wc + i, Wc[i]);
#pragma unroll
        for (int i = 0; i < 36; i++) atomicAdd(wm + i, Wm[i]);
        if (INTR) {
            double *wk = Wl + sk * WLS;
#pragma unroll
            for (int i = 0; i < 24; i++) atomicAdd(wk + i, Wk[i]);
        }
    }
    // V, g, err: wave sums, then one LDS add per wave and value
    wave_sum_lds<28, CH>(vals, scratch + wave * (CH * 64), lane, [&](int i, double s) { atomicAdd(acc + i, s); });
    __syncthreads();
    passA_epilogue<BLOCK>(a, Wl, acc, scratch, f, s0, kf);
}

// ------------------------------------------------------------------------------------------------
// Pass A in wrench form (geom.hpp, corner_wrench): the same workgroup per frame and the same results up to rounding, but a lane only forms
// H = sum w w^T (21 values) of its observation and adds it to the observation's camera slot and marker slot in LDS; sum w r and sum r^2 stay in
// registers.  After the loop: H_f = the sum of the camera slots (cameras come first in a frame's ascending slot list and every observation has one),
// V_f = F^T H_f F, g_f = F^T sum w r, and one lane per slot turns its H into the W block T^T H F and stores it.
//   ~720 instead of ~1 500 fp64 instructions per observation, 42 instead of 72 LDS atomics, 28 accumulators instead of 100 (profiles/r04_vgpr_counts.txt);
//   LDS holds nothing but the H slots (21 doubles each: 21 KB instead of 69 KB at config 5's 122 slots per frame -- the row form also stages the
//   entities' {R, t, J_l} rows and the W blocks): the kernel is bound by the dependent fetches and barriers of a workgroup, i.e. by how many workgroups
//   a CU holds, and the entity rows are L1/L2 hits fetched next to the observation's index record without a staging phase before them.
// LDS: [max_kf*21] H slots | [32] V,g,err | [32] H_f, sum w r, sum r^2, [30] = camera slots | [24] the frame's row | [36] H_f F | [36] (V_f + mu I)^-1
// ------------------------------------------------------------------------------------------------
constexpr int HLS = 21, HLS_INTR = 25;   // doubles per slot: H (21); with intrinsics entities their slots hold M = sum G_k^T w^T (4 x 6 = 24)
__host__ __device__ constexpr size_t passA_h_doubles(int max_kf, int hls = HLS) { return ((size_t)max_kf * hls + 1) & ~(size_t)1; }

#ifdef AAR_PASSA_STAMPS   // diagnostic build (make HIPFLAGS+=-DAAR_PASSA_STAMPS; AAR_STAMPS_A=<file>; scripts/dev/passA_stamps_report.py): cycle stamps of a workgroup's phases
#define PA_STAMP(n) do { if (a.stamps && f < 512 && tid == 0) a.stamps[f * 16 + (n)] = __builtin_readcyclecounter(); } while (0)
#else
#define PA_STAMP(n) do { } while (0)
#endif

// INTR: camera intrinsics are optimised -- the observation's block against its camera's intrinsics entity is W_kf = sum G_k^T G_f = (sum G_k^T w^T) F: the slot of that
// entity collects M = sum G_k^T w^T (4 x 6) and gets F once, like the others
template <int BLOCK, int CPL, bool INTR = false>
__device__ __forceinline__ void passA_wrench_body(const PassAArgs &a, double *lds, const int f, const int n_blocks_a) {
    constexpr int SL = INTR ? HLS_INTR : HLS;
    double *Hl = lds;
    double *acc = lds + passA_h_doubles(a.max_kf, SL);
    double *hacc = acc + 32;
    double *frow = acc + 64;
    double *Yl = frow + 24;
    double *vil = Yl + 36;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o0 = a.frame_obs_start[f], nobs = a.frame_obs_start[f + 1] - o0;
    const int s0 = a.fslot_start[f], kf = a.fslot_start[f + 1] - s0;
    PA_STAMP(0);
    // A workgroup is a chain of dependent fetches (index record -> entity rows ... slot entity -> its row) and barriers; everything that does not depend
    // on LDS is issued up front: the frame's own row (SGPRs: wave-uniform), the first observation's index record and corners, the slot's entity
    Ent ef;
    load_ent(a.ent, a.A + __builtin_amdgcn_readfirstlane(f), ef);
    // Consecutive lanes take observations `stride` apart (a permutation of the frame: the host picks stride coprime with nobs) so that the lanes of a
    // wavefront mostly hold different cameras -- the order inside a frame is camera-major, and same-address ds_add_f64 serialise (+50 % at config 5).
    constexpr int LPO = 4 / CPL;   // lanes per observation
    const int stride = a.frame_stride[f];
    const int part = tid % LPO;
    int pos = 0, step = 0;
    if (nobs > 0) {
        pos = (int)(((unsigned long long)(tid / LPO) * (unsigned)stride) % (unsigned)nobs);
        step = (int)(((unsigned long long)(BLOCK / LPO) * (unsigned)stride) % (unsigned)nobs);
    }
    int t = tid;
    bool have = t < nobs * LPO;
    ObsIdx id = {0, 0, 0, 0};
    float2 ouv[CPL];
    EntRT ec, em;
    double K[9];
    auto fetch_obs = [&]() {
        const int o = o0 + pos;
        id = a.idx[o];
#pragma unroll
        for (int kk = 0; kk < CPL; kk++) ouv[kk] = reinterpret_cast<const float2 *>(a.uv)[4 * (int64_t)o + part * CPL + kk];
    };
    auto fetch_rows = [&]() {
        load_ent_rt(a.ent, id.cam, ec);
        load_ent_rt(a.ent, id.marker, em);
#pragma unroll
        for (int i = 0; i < 9; i++) K[i] = a.Kmat[a.kstride * id.cam + i];
    };
    if (have) fetch_obs();
    const int e_slot = tid < kf ? a.fslot_ent[s0 + tid] : -1;
    double2 frv = make_double2(0.0, 0.0);
    if (tid < ENT_STRIDE / 2) frv = reinterpret_cast<const double2 *>(a.ent + (size_t)(a.A + f) * ENT_STRIDE)[tid];
    for (int i = tid; i < kf * SL; i += BLOCK) Hl[i] = 0.0;
    for (int i = tid; i < 64; i += BLOCK) acc[i] = 0.0;
    if (have) fetch_rows();
    {   // grid-stride clearing of the block set that is dead by now
        const int64_t gid = (int64_t)f * BLOCK + tid, gstride = (int64_t)n_blocks_a * BLOCK;
        for (int64_t i = gid; i < a.zero0_n; i += gstride) a.zero0[i] = 0.0;
        for (int64_t i = gid; i < a.zero1_n; i += gstride) a.zero1[i] = 0.0;
        for (int64_t i = gid; i < a.zero2_n; i += gstride) a.zero2[i] = 0.0;
    }
    __syncthreads();
    PA_STAMP(1);

    double vals[7];  // sum w r (6), sum r^2
#pragma unroll
    for (int i = 0; i < 7; i++) vals[i] = 0.0;
    while (have) {
        const int sc = id.slots & ((1 << SLOT_C_BITS) - 1), sm = (id.slots >> SLOT_C_BITS) & ((1 << SLOT_M_BITS) - 1);
        const int sk = (id.slots >> (SLOT_C_BITS + SLOT_M_BITS)) & ((1 << SLOT_M_BITS) - 1);
        double H[21], Mk[INTR ? 24 : 1];
#pragma unroll
        for (int i = 0; i < 21; i++) H[i] = 0.0;
        if (INTR) {
#pragma unroll
            for (int i = 0; i < 24; i++) Mk[i] = 0.0;
        }
#pragma unroll (INTR && CPL == 4 ? 2 : CPL)
        for (int kk = 0; kk < CPL; kk++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, a.h, part * CPL + kk, g);
            double r[2], w[2][6];
            corner_residual(ouv[kk].x, ouv[kk].y, g.u, g.v, a.res_f32, a.huber, r[0], r[1]);
            corner_wrench(ec, K, g, w);
            if (INTR) {
                double Gk[2][4];
                corner_jacobian_intr(g, Gk);
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) Mk[i * 6 + j] += Gk[rr][i] * w[rr][j];
            }
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                vals[6] += r[rr] * r[rr];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    vals[i] += w[rr][i] * r[rr];
#pragma unroll
                    for (int j = 0; j <= i; j++) H[i * (i + 1) / 2 + j] += w[rr][i] * w[rr][j];
                }
            }
        }
        t += BLOCK;
        have = t < nobs * LPO;
        if (have) {   // the next observation's fetches go out before this one's LDS additions
            pos += step;
            if (pos >= nobs) pos -= nobs;
            fetch_obs();
        }
        double *hc = Hl + sc * SL, *hm = Hl + sm * SL;
#pragma unroll
        for (int i = 0; i < 21; i++) atomicAdd(hc + i, H[i]);
#pragma unroll
        for (int i = 0; i < 21; i++) atomicAdd(hm + i, H[i]);
        if (INTR) {
            double *hk = Hl + sk * SL;
#pragma unroll
            for (int i = 0; i < 24; i++) atomicAdd(hk + i, Mk[i]);
        }
        if (have) fetch_rows();
    }
    PA_STAMP(2);
    // the slot's entity row (doubles 8 .. 21: R[8] | t | J_l) is on its way during the sums and barriers below
    double2 rv[7];
    if (e_slot >= 0) {
        const double2 *rowp = reinterpret_cast<const double2 *>(a.ent + (size_t)e_slot * ENT_STRIDE);
#pragma unroll
        for (int i = 0; i < 7; i++) rv[i] = rowp[4 + i];
        if (e_slot < a.C) atomicAdd(hacc + 30, 1.0);
    }
    for (int ts = tid + BLOCK; ts < kf; ts += BLOCK)
        if (a.fslot_ent[s0 + ts] < a.C) atomicAdd(hacc + 30, 1.0);
    if (tid < ENT_STRIDE / 2) { frow[2 * tid] = frv.x; frow[2 * tid + 1] = frv.y; }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double sv = wave_sum_dpp(vals[i]);
        if (lane == 0) atomicAdd(hacc + 21 + i, sv);
    }
    PA_STAMP(3);
    __syncthreads();
    PA_STAMP(4);
    if (tid < 21) {   // H_f: the camera slots of the frame
        const int nc = (int)hacc[30];
        double sv = 0.0;
        for (int ts = 0; ts < nc; ts++) sv += Hl[ts * SL + tid];
        hacc[tid] = sv;
    }
    __syncthreads();
    PA_STAMP(5);
    if (wave == BLOCK / 64 - 1) {   // V_f = F^T H_f F (through Y = H_f F), g_f = F^T sum w r -- on the LAST wavefront: the slots' lanes start with the first
        const double *jl = frow + 12;
        if (lane < 36) {
            const int k = lane / 6, j = lane - 6 * k;
            Yl[lane] = j < 3 ? hacc[sym6(k, 0)] * jl[j] + hacc[sym6(k, 1)] * jl[3 + j] + hacc[sym6(k, 2)] * jl[6 + j] : hacc[sym6(k, j)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 28) {
            double out;
            if (lane < 21) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= lane) i++;
                const int j = lane - i * (i + 1) / 2;
                out = i < 3 ? jl[i] * Yl[j] + jl[3 + i] * Yl[6 + j] + jl[6 + i] * Yl[12 + j] : Yl[i * 6 + j];
            } else if (lane < 27) {
                const int i = lane - 21;
                out = i < 3 ? jl[i] * hacc[21] + jl[3 + i] * hacc[22] + jl[6 + i] * hacc[23] : hacc[21 + i];
            } else out = hacc[27];
            acc[lane] = out;
        }
        // ... and (V_f + mu I)^-1, h_f for the damping the next solve is expected to use, on this wavefront's last lane -- with more than one wavefront
        // per workgroup that is beside the slots' lanes, not after them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (a.mu_pred >= 0.0 && lane == 63) {
            double out[36];
            if (a.frames_fixed) {
#pragma unroll
                for (int i = 0; i < 36; i++) out[i] = 0.0;
            } else {
                double m[6][6];
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j < 6; j++) m[i][j] = acc[sym6(i, j)] + (i == j ? a.mu_pred : 0.0);
                if (!spd6_inverse(m, out)) atomicOr(a.flags, 1);
            }
            const bool dense_ = a.Yd != nullptr;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double hv = 0.0;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    a.Vinv[(size_t)f * 36 + i * 6 + j] = out[i * 6 + j];
                    hv += out[i * 6 + j] * acc[21 + j];
                    if (dense_) vil[i * 6 + j] = out[i * 6 + j];
                }
                a.hf[(size_t)f * 6 + i] = hv;
            }
        }
    }
    PA_STAMP(6);
    // one lane per slot: W = T^T H F, rows stored as they come
    for (int ts = tid; ts < kf; ts += BLOCK) {
        int e = e_slot;
        if (ts != tid) {   // (frames with more slots than lanes)
            e = a.fslot_ent[s0 + ts];
            const double2 *rowp = reinterpret_cast<const double2 *>(a.ent + (size_t)e * ENT_STRIDE);
#pragma unroll
            for (int i = 0; i < 7; i++) rv[i] = rowp[4 + i];
        }
        if (INTR && e >= a.k_ent0) {   // an intrinsics entity's slot: W = M F (rows fx, cx, fy, cy; the entity's two idle rows are zero)
            const double *mk = Hl + ts * SL;
            double wv[36];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double m6[6];
#pragma unroll
                for (int j = 0; j < 6; j++) m6[j] = mk[i * 6 + j];
#pragma unroll
                for (int j = 0; j < 3; j++) wv[6 * i + j] = m6[0] * ef.Jl[j] + m6[1] * ef.Jl[3 + j] + m6[2] * ef.Jl[6 + j];
                wv[6 * i + 3] = m6[3]; wv[6 * i + 4] = m6[4]; wv[6 * i + 5] = m6[5];
            }
#pragma unroll
            for (int q = 24; q < 36; q++) wv[q] = 0.0;
            if (a.W) {
                double2 *wp = reinterpret_cast<double2 *>(a.W + (size_t)(s0 + ts) * 36);
#pragma unroll
                for (int q = 0; q < 18; q++) wp[q] = make_double2(wv[2 * q], wv[2 * q + 1]);
            }
            if (a.Wf) {
                float4 *wf = reinterpret_cast<float4 *>(a.Wf + (size_t)s0 * 36) + ts;
#pragma unroll
                for (int q = 0; q < 9; q++) wf[(size_t)q * kf] = make_float4((float)wv[4 * q], (float)wv[4 * q + 1], (float)wv[4 * q + 2], (float)wv[4 * q + 3]);
            }
            continue;
        }
        const bool cam = e < a.C;
        const double tt[3] = {cam ? rv[0].y - ef.t[0] : rv[0].y, cam ? rv[1].x - ef.t[1] : rv[1].x, cam ? rv[1].y - ef.t[2] : rv[1].y};   // d = t_c - t_f, or t_m
        const double Jl[9] = {rv[2].x, rv[2].y, rv[3].x, rv[3].y, rv[4].x, rv[4].y, rv[5].x, rv[5].y, rv[6].x};
        double H[21], X[6][6];
        const double *slot = Hl + ts * SL;
#pragma unroll
        for (int i = 0; i < 21; i++) H[i] = slot[i];
        // camera: T^T = -[J_l^T, -J_l^T [d]x; 0, I]      marker: T^T = [J_l^T R_f^T, -J_l^T [t_m]x R_f^T; 0, R_f^T]
#pragma unroll
        for (int l = 0; l < 6; l++) {
            double zt[3], zb[3], x[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                zt[i] = cam ? H[sym6(i, l)] : ef.R[i] * H[sym6(0, l)] + ef.R[3 + i] * H[sym6(1, l)] + ef.R[6 + i] * H[sym6(2, l)];
                zb[i] = cam ? H[sym6(3 + i, l)] : ef.R[i] * H[sym6(3, l)] + ef.R[3 + i] * H[sym6(4, l)] + ef.R[6 + i] * H[sym6(5, l)];
            }
            cross3(tt, zb, x);
            const double v[3] = {zt[0] - x[0], zt[1] - x[1], zt[2] - x[2]};
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double wt = Jl[i] * v[0] + Jl[3 + i] * v[1] + Jl[6 + i] * v[2];
                X[i][l] = cam ? -wt : wt;
                X[3 + i][l] = cam ? -zb[i] : zb[i];
            }
        }
        double2 *wp = reinterpret_cast<double2 *>(a.W + (size_t)(s0 + ts) * 36);
        float4 *wf = a.Wf ? reinterpret_cast<float4 *>(a.Wf + (size_t)s0 * 36) + ts : nullptr;
#pragma unroll
        for (int i2 = 0; i2 < 3; i2++) {   // two rows = three float4 pieces of the fp32 copy
            float w12[12];
#pragma unroll
            for (int ii = 0; ii < 2; ii++) {
                const int i = 2 * i2 + ii;
                double wr[3];
#pragma unroll
                for (int j = 0; j < 3; j++) wr[j] = X[i][0] * ef.Jl[j] + X[i][1] * ef.Jl[3 + j] + X[i][2] * ef.Jl[6 + j];
                if (a.W) {
                    wp[3 * i] = make_double2(wr[0], wr[1]);
                    wp[3 * i + 1] = make_double2(wr[2], X[i][3]);
                    wp[3 * i + 2] = make_double2(X[i][4], X[i][5]);
                }
                w12[6 * ii] = (float)wr[0]; w12[6 * ii + 1] = (float)wr[1]; w12[6 * ii + 2] = (float)wr[2];
                w12[6 * ii + 3] = (float)X[i][3]; w12[6 * ii + 4] = (float)X[i][4]; w12[6 * ii + 5] = (float)X[i][5];
            }
            if (wf) {
#pragma unroll
                for (int q = 0; q < 3; q++) wf[(size_t)(3 * i2 + q) * kf] = make_float4(w12[4 * q], w12[4 * q + 1], w12[4 * q + 2], w12[4 * q + 3]);
            }
        }
    }
    PA_STAMP(7);
    __syncthreads();
    PA_STAMP(8);
    for (int q = tid; q < 36; q += BLOCK) a.V[(size_t)f * 36 + q] = acc[sym6(q / 6, q % 6)];
    for (int q = tid; q < 6; q += BLOCK) a.gf[(size_t)f * 6 + q] = acc[21 + q];
    if (tid == 0) a.err_part[f] = acc[27];
    const bool dense = a.Yd != nullptr && a.mu_pred >= 0.0;
    PA_STAMP(9);
    if (dense) {   // the frame's panels of the MFMA Schur path: W as it is, Y = W (V_f + mu I)^-1; the pseudo entity 0 carries g_f in its row 0
        __syncthreads();
        double vi[36];
#pragma unroll
        for (int q = 0; q < 36; q++) vi[q] = vil[q];
        const size_t fbase = (size_t)f * a.Ad * 36;
        for (int ts = tid; ts < kf; ts += BLOCK) {   // (a lane reads back the block it has stored itself)
            const double2 *wp = reinterpret_cast<const double2 *>(a.W + (size_t)(s0 + ts) * 36);
            const size_t o = fbase + (size_t)a.slot_dense[s0 + ts] * 36;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double2 w0 = wp[3 * i], w1 = wp[3 * i + 1], w2 = wp[3 * i + 2];
                const double w[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
                double y[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 6; k++)
#pragma unroll
                    for (int j = 0; j < 6; j++) y[j] = fma(w[k], vi[k * 6 + j], y[j]);
                double2 *yp = reinterpret_cast<double2 *>(a.Yd + o + i * 6), *wd = reinterpret_cast<double2 *>(a.Wd + o + i * 6);
                yp[0] = make_double2(y[0], y[1]); yp[1] = make_double2(y[2], y[3]); yp[2] = make_double2(y[4], y[5]);
                wd[0] = w0; wd[1] = w1; wd[2] = w2;
            }
        }
        if (tid < 6) a.Wd[fbase + tid] = acc[21 + tid];
    }
}

// ------------------------------------------------------------------------------------------------
// Pass B: observations sorted by (camera, marker, frame); one wavefront per chunk of <= PASSB_CHUNK
// observations of a single (camera, marker) pair.  Each lane accumulates the 90 values of
// U_cc (21), U_mm (21), W_cm (36), g_c (6), g_m (6) over its observations in registers; one LDS wave sum;
// one fp64 atomic per value into the dense shared system (lower triangle, row-major, cameras first).
// ------------------------------------------------------------------------------------------------
struct PassBArgs {
    const ObsIdx *idx; const float *uv; const double *ent; const double *Kmat; const int32_t *chunk_start;
    int kstride, k_ent0;   // Kmat[kstride * camera + i]; first intrinsics entity (C + M) for the intrinsics variant
    int n_chunks, A, res_f32, n_pad;
    float huber;
    double h;
    double *U0, *g0;   // U0 = blk.S (zeroed), g0 = blk.g0
    double *part; int part_stride;   // deterministic mode: the chunk's sums go to part[chunk * part_stride + value] instead of atomics
};

// one wavefront per chunk; `scratch` = 2048 doubles of LDS per wavefront of the workgroup
// DET: the 90 sums of the chunk are stored as the chunk's record (k_passB_reduce adds the records up in a fixed order)
// LEAN: the corner loop is NOT unrolled -- the four corners' temporaries then take turns in the same registers (274 instead of 390: with
// __launch_bounds__(256, 2) 23 of them spill and TWO wavefronts share a SIMD).  For pass B as a launch of its own (many chunks per SIMD: config 5); inside
// the merged launch pass A's registers set the occupancy anyway and the unrolled loop's instruction-level parallelism is worth more
template <bool DET = false, bool LEAN = false>
__device__ __forceinline__ void passB_body(const PassBArgs &b, double *scratch, const int first_chunk) {
    const ObsIdx *__restrict__ idx = b.idx;
    const float *__restrict__ uv = b.uv;
    const double *__restrict__ ent = b.ent, *__restrict__ Kmat = b.Kmat;
    const int32_t *__restrict__ chunk_start = b.chunk_start;
    const int n_chunks = b.n_chunks, A = b.A, res_f32 = b.res_f32, n_pad = b.n_pad;
    const float huber = b.huber;
    const double h = b.h;
    double *__restrict__ U0 = b.U0, *__restrict__ g0 = b.g0;
    // the chunk index is wave-uniform and the compiler is told so: the head record, the camera's and the marker's {R, t, J_l} rows
    // and K then arrive through the scalar cache into SGPRs instead of 102 VGPRs (512 VGPRs with 34 spilled -> 390, none spilled)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = first_chunk + wave;
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
    for (int i = 0; i < 9; i++) K[i] = Kmat[b.kstride * head.cam + i];
    for (int o = o0 + lane; o < o1; o += 64) {
        const ObsIdx id = idx[o];
        const float4 uv0 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o];
        const float4 uv1 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o + 1];
        const float ou[8] = {uv0.x, uv0.y, uv0.z, uv0.w, uv1.x, uv1.y, uv1.z, uv1.w};
        Ent ef;
        load_ent(ent, A + id.frame, ef);
#pragma unroll LEAN ? 1 : 4
        for (int k = 0; k < 4; k++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, h, k, g);
            double r[2];
            float ox, oy;
            if (LEAN) { const float2 t = reinterpret_cast<const float2 *>(uv)[4 * (int64_t)o + k]; ox = t.x; oy = t.y; }   // (a register array indexed by a loop counter would live in scratch)
            else { ox = ou[2 * k]; oy = ou[2 * k + 1]; }
            corner_residual(ox, oy, g.u, g.v, res_f32, huber, r[0], r[1]);
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
        if (DET) { b.part[(size_t)chunk * b.part_stride + v] = s; return; }
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
// Pass B in wrench form.  A row's wrench in CAMERA coordinates about the camera's origin is w = (pc x beta, beta), beta = d(u,v)_r / d(pc) -- no entity
// enters it -- and both Jacobian blocks of the row are images of it:  G_c = w^T T_c,  T_c = -[R_c^T J_l(c) 0; 0 R_c^T]  (the same for the whole chunk),
// G_m = (X w)^T [J_l(m) 0; 0 I]  with the observation's transport  X = [Q [e]x Q; 0 Q],  Q = R_f^T R_c,  e = R_f^T (t_c - t_f) - t_m  (the camera-frame
// wrench seen from the marker's origin in object coordinates).  So an observation costs its Gram matrix H = sum w w^T (21 + 6 values from 8 rows instead of
// 90), H X^T and X H X^T column by column (~330 fp64 instructions), and the lane's 90 sums are kept in wrench coordinates; the chunk's 90 values get their
// entity matrices once, after the wave sum, on the lanes that issue the atomics.  ~1 000 instead of ~1 400 fp64 instructions per observation; same results
// up to rounding.
// ------------------------------------------------------------------------------------------------
template <bool DET = false>
__device__ __forceinline__ void passB_wrench_body(const PassBArgs &b, double *scratch, const int first_chunk) {
    __shared__ double pb_red[4][128];   // per wavefront: the chunk's 90 sums | [90, 126): the entity matrices F_c top / bottom, F_m top / bottom (3x3 each)
    const ObsIdx *__restrict__ idx = b.idx;
    const float *__restrict__ uv = b.uv;
    const double *__restrict__ ent = b.ent, *__restrict__ Kmat = b.Kmat;
    const int32_t *__restrict__ chunk_start = b.chunk_start;
    const int n_chunks = b.n_chunks, A = b.A, res_f32 = b.res_f32, n_pad = b.n_pad;
    const float huber = b.huber;
    const double h = b.h;
    double *__restrict__ U0 = b.U0, *__restrict__ g0 = b.g0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the chunk's rows arrive in SGPRs
    const int chunk = first_chunk + wave;
    if (chunk >= n_chunks) return;
    const int o0 = chunk_start[chunk], o1 = chunk_start[chunk + 1];
    double vals[90];   // H_c (21) | H_m (21) | H_cm (36) | b_c (6) | b_m (6), in wrench coordinates
#pragma unroll
    for (int i = 0; i < 90; i++) vals[i] = 0.0;
    const ObsIdx head = idx[o0];
    Ent ec, em;
    load_ent(ent, head.cam, ec);
    load_ent(ent, head.marker, em);
    double K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = Kmat[b.kstride * head.cam + i];
    for (int o = o0 + lane; o < o1; o += 64) {
        const ObsIdx id = idx[o];
        const float4 uv0 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o];
        const float4 uv1 = reinterpret_cast<const float4 *>(uv)[2 * (int64_t)o + 1];
        const float ou[8] = {uv0.x, uv0.y, uv0.z, uv0.w, uv1.x, uv1.y, uv1.z, uv1.w};
        EntRT ef;
        load_ent_rt(ent, A + id.frame, ef);
        double H[21], bw[6];
#pragma unroll
        for (int i = 0; i < 21; i++) H[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) bw[i] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, h, k, g);
            double r[2];
            corner_residual(ou[2 * k], ou[2 * k + 1], g.u, g.v, res_f32, huber, r[0], r[1]);
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                double w[6];
#pragma unroll
                for (int j = 0; j < 3; j++) w[3 + j] = (K[3 * rr + j] - (rr ? g.v : g.u) * K[6 + j]) * g.iw;
                cross3(g.pc, &w[3], &w[0]);
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    bw[i] += w[i] * r[rr];
#pragma unroll
                    for (int j = 0; j <= i; j++) H[i * (i + 1) / 2 + j] += w[i] * w[j];
                }
            }
        }
        // the observation's transport X = [Q E; 0 Q]
        double Q[3][3], E[3][3], e3[3];
        {
            const double d[3] = {ec.t[0] - ef.t[0], ec.t[1] - ef.t[1], ec.t[2] - ef.t[2]};
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int j = 0; j < 3; j++) Q[i][j] = ef.R[i] * ec.R[j] + ef.R[3 + i] * ec.R[3 + j] + ef.R[6 + i] * ec.R[6 + j];
                e3[i] = ef.R[i] * d[0] + ef.R[3 + i] * d[1] + ef.R[6 + i] * d[2] - em.t[i];
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double qc[3] = {Q[0][j], Q[1][j], Q[2][j]};
                double x[3];
                cross3(e3, qc, x);
                E[0][j] = x[0]; E[1][j] = x[1]; E[2][j] = x[2];
            }
        }
#pragma unroll
        for (int i = 0; i < 21; i++) vals[i] += H[i];
#pragma unroll
        for (int i = 0; i < 6; i++) vals[78 + i] += bw[i];
#pragma unroll
        for (int i = 0; i < 3; i++) {   // b_m += X b
            vals[84 + i] += Q[i][0] * bw[0] + Q[i][1] * bw[1] + Q[i][2] * bw[2] + E[i][0] * bw[3] + E[i][1] * bw[4] + E[i][2] * bw[5];
            vals[87 + i] += Q[i][0] * bw[3] + Q[i][1] * bw[4] + Q[i][2] * bw[5];
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {   // column j of H X^T (x = row j of X), then of X H X^T (rows i >= j)
            double p[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                if (j < 3) p[i] = H[sym6(i, 0)] * Q[j][0] + H[sym6(i, 1)] * Q[j][1] + H[sym6(i, 2)] * Q[j][2] + H[sym6(i, 3)] * E[j][0] + H[sym6(i, 4)] * E[j][1] + H[sym6(i, 5)] * E[j][2];
                else p[i] = H[sym6(i, 3)] * Q[j - 3][0] + H[sym6(i, 4)] * Q[j - 3][1] + H[sym6(i, 5)] * Q[j - 3][2];
                vals[42 + i * 6 + j] += p[i];
            }
#pragma unroll
            for (int i = j; i < 6; i++) {
                double q;
                if (i < 3) q = Q[i][0] * p[0] + Q[i][1] * p[1] + Q[i][2] * p[2] + E[i][0] * p[3] + E[i][1] * p[4] + E[i][2] * p[5];
                else q = Q[i - 3][0] * p[3] + Q[i - 3][1] * p[4] + Q[i - 3][2] * p[5];
                vals[21 + i * (i + 1) / 2 + j] += q;
            }
        }
    }
    double *red = pb_red[wave & 3];
    wave_sum_lds<90>(vals, scratch + wave * 2048, lane, [&](int v, double s) { red[v] = s; });
    if (lane == 0) {   // F_c = [R_c^T J_l(c) 0; 0 R_c^T] (the minus sign goes to g_c and W_cm), F_m = [J_l(m) 0; 0 I]; entry [k][i]: wrench component k, parameter i
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                red[90 + 3 * k + i] = ec.R[k] * ec.Jl[i] + ec.R[3 + k] * ec.Jl[3 + i] + ec.R[6 + k] * ec.Jl[6 + i];
                red[99 + 3 * k + i] = ec.R[3 * i + k];
                red[108 + 3 * k + i] = em.Jl[3 * k + i];
                red[117 + 3 * k + i] = k == i ? 1.0 : 0.0;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int rc = 6 * head.cam, rm = 6 * head.marker;  // first row of the camera / marker block
    for (int v = lane; v < 90; v += 64) {
        double out = 0.0;
        double *dst;
        if (v < 78) {   // out(i, j) = sum_{a, c < 3} FA[3 bi + a][i] M[3 bi + a][3 bj + c] FB[3 bj + c][j]: FA / FB block-diagonal
            int i, j, mbase, fa, fb;
            bool sym;
            if (v < 42) {
                const int p = v < 21 ? v : v - 21;
                i = 0;
                while ((i + 1) * (i + 2) / 2 <= p) i++;
                j = p - i * (i + 1) / 2;
                mbase = v < 21 ? 0 : 21; fa = fb = v < 21 ? 90 : 108; sym = true;
                const int base = v < 21 ? rc : rm;
                dst = U0 + (size_t)(base + i) * n_pad + base + j;
            } else {
                i = (v - 42) / 6; j = (v - 42) % 6;
                mbase = 42; fa = 90; fb = 108; sym = false;
                dst = U0 + (size_t)(rm + j) * n_pad + rc + i;   // W_cm[i][j] -> row of the marker (below the cameras), column of the camera
            }
            const int bi = i / 3, bj = j / 3;
            const double *FA = red + fa + 9 * bi + (i - 3 * bi), *FB = red + fb + 9 * bj + (j - 3 * bj);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                double t = 0.0;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int k = 3 * bi + a, l = 3 * bj + c;
                    t += red[mbase + (sym ? sym6(k, l) : k * 6 + l)] * FB[3 * c];
                }
                out += FA[3 * a] * t;
            }
            if (!sym) out = -out;
        } else {        // g_c = -F_c^T b_c,  g_m = F_m^T b_m
            const bool cam = v < 84;
            const int i = cam ? v - 78 : v - 84, bi = i / 3;
            const double *F = red + (cam ? 90 : 108) + 9 * bi + (i - 3 * bi), *bv = red + (cam ? 78 : 84) + 3 * bi;
            out = F[0] * bv[0] + F[3] * bv[1] + F[6] * bv[2];
            if (cam) out = -out;
            dst = g0 + (cam ? rc : rm) + i;
        }
        if (DET) b.part[(size_t)chunk * b.part_stride + v] = out;
        else atomicAdd(dst, out);
    }
}

// The two passes as kernels of their own, and as ONE launch: workgroups [0, F) are pass A's, the rest pass B's.  They
// are independent once the {R, t, J_l} table exists (k_backsub / k_unpack write it), so the trial evaluation of an LM step
// runs them side by side; both are latency-bound with one wavefront per SIMD and together still fit the chip at config 3.
// WR: the wrench form (the default without intrinsics entities); the row form stays for the intrinsics variants and as the A/B reference (AAR_PASSA_WRENCH=0)
template <int BLOCK, int CPL, bool WR>
__global__ void __launch_bounds__(BLOCK) k_passA(const PassAArgs a) {
    extern __shared__ double lds[];
    if (WR) passA_wrench_body<BLOCK, CPL>(a, lds, (int)blockIdx.x, (int)gridDim.x);
    else passA_body<BLOCK, CPL>(a, lds, (int)blockIdx.x, (int)gridDim.x);
}

template <bool WR>
__global__ void __launch_bounds__(256) k_passB(const PassBArgs b) {
    __shared__ double scratch[4 * 2048];
    if (WR) passB_wrench_body(b, scratch, (int)blockIdx.x * 4);
    else passB_body(b, scratch, (int)blockIdx.x * 4);
}
template <bool WR>
__global__ void __launch_bounds__(256) k_passB_det(const PassBArgs b) {
    __shared__ double scratch[4 * 2048];
    if (WR) passB_wrench_body<true>(b, scratch, (int)blockIdx.x * 4);
    else passB_body<true>(b, scratch, (int)blockIdx.x * 4);
}
// experiments (AAR_PASSB_LEAN=1 / 2): the corner loop not unrolled, with / without the register cap that lets two wavefronts share a SIMD
__global__ void __launch_bounds__(256, 2) k_passB_lean2(const PassBArgs b) {
    __shared__ double scratch[4 * 2048];
    passB_body<false, true>(b, scratch, (int)blockIdx.x * 4);
}
__global__ void __launch_bounds__(256) k_passB_lean1(const PassBArgs b) {
    __shared__ double scratch[4 * 2048];
    passB_body<false, true>(b, scratch, (int)blockIdx.x * 4);
}

// Pass B for the intrinsics entities (optimize_cam_intrinsics): same chunks, the 62 values of U_kk (4x4, packed lower), W_kc
// (4x6: intrinsics x the camera's own pose), W_km (4x6: intrinsics x marker) and g_k; rows of entity k_ent0 + camera.
template <bool DET = false>
__device__ __forceinline__ void passB_intr_body(const PassBArgs &b, double *scratch, const int first_chunk) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, see passB_body
    const int chunk = first_chunk + wave;
    if (chunk >= b.n_chunks) return;
    const int o0 = b.chunk_start[chunk], o1 = b.chunk_start[chunk + 1];
    double vals[62];
#pragma unroll
    for (int i = 0; i < 62; i++) vals[i] = 0.0;
    const ObsIdx head = b.idx[o0];
    Ent ec, em;
    load_ent(b.ent, head.cam, ec);
    load_ent(b.ent, head.marker, em);
    double K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = b.Kmat[b.kstride * head.cam + i];
    for (int o = o0 + lane; o < o1; o += 64) {
        const ObsIdx id = b.idx[o];
        const float4 uv0 = reinterpret_cast<const float4 *>(b.uv)[2 * (int64_t)o];
        const float4 uv1 = reinterpret_cast<const float4 *>(b.uv)[2 * (int64_t)o + 1];
        const float ou[8] = {uv0.x, uv0.y, uv0.z, uv0.w, uv1.x, uv1.y, uv1.z, uv1.w};
        Ent ef;
        load_ent(b.ent, b.A + id.frame, ef);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CornerGeom g;
            project_corner(ec, em, ef, K, b.h, k, g);
            double r[2];
            corner_residual(ou[2 * k], ou[2 * k + 1], g.u, g.v, b.res_f32, b.huber, r[0], r[1]);
            double Gc[2][6], Gm[2][6], Gf[2][6], Gk[2][4];
            corner_jacobian<true, true, false>(ec, em, ef, K, g, Gc, Gm, Gf);
            corner_jacobian_intr(g, Gk);
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    vals[58 + i] += Gk[rr][i] * r[rr];
#pragma unroll
                    for (int j = 0; j <= i; j++) vals[i * (i + 1) / 2 + j] += Gk[rr][i] * Gk[rr][j];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        vals[10 + i * 6 + j] += Gk[rr][i] * Gc[rr][j];
                        vals[34 + i * 6 + j] += Gk[rr][i] * Gm[rr][j];
                    }
                }
        }
    }
    const int rc = 6 * head.cam, rm = 6 * head.marker, rk = 6 * (b.k_ent0 + head.cam);
    double *__restrict__ U0 = b.U0, *__restrict__ g0 = b.g0;
    const int n_pad = b.n_pad;
    wave_sum_lds<62>(vals, scratch + wave * 2048, lane, [&](int v, double s) {
        if (DET) { b.part[(size_t)chunk * b.part_stride + 90 + v] = s; return; }
        if (v < 10) {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= v) i++;
            const int j = v - i * (i + 1) / 2;
            atomicAdd(U0 + (size_t)(rk + i) * n_pad + rk + j, s);
        } else if (v < 34) {   // the intrinsics rows lie below cameras and markers: (k, c) and (k, m) are lower-triangle blocks
            const int i = (v - 10) / 6, j = (v - 10) % 6;
            atomicAdd(U0 + (size_t)(rk + i) * n_pad + rc + j, s);
        } else if (v < 58) {
            const int i = (v - 34) / 6, j = (v - 34) % 6;
            atomicAdd(U0 + (size_t)(rk + i) * n_pad + rm + j, s);
        } else {
            atomicAdd(g0 + rk + (v - 58), s);
        }
    });
}

__global__ void __launch_bounds__(256) k_passB_intr(const PassBArgs b) {
    __shared__ double scratch[4 * 2048];
    passB_intr_body(b, scratch, (int)blockIdx.x * 4);
}
__global__ void __launch_bounds__(256) k_passB_intr_det(const PassBArgs b) {
    __shared__ double scratch[4 * 2048];
    passB_intr_body<true>(b, scratch, (int)blockIdx.x * 4);
}

// Deterministic mode, second half of pass B: one wavefront per reduction item (a camera, a marker, a (camera, marker) pair), one
// lane per value; the chunk records are added in ascending chunk order = ascending (camera, marker, frame) order of the
// observations, and the sum is STORED (the block set was cleared before).  Same destinations as the atomics of passB_body.
struct PassBReduceArgs {
    const int32_t *start, *chunk, *kind, *ea, *eb;
    const double *part;
    int stride, n_items, n_pad, intr, k_ent0;
    double *U0, *g0;
};
__device__ __forceinline__ void unpack_lower(int p, int &i, int &j) {
    i = 0;
    while ((i + 1) * (i + 2) / 2 <= p) i++;
    j = p - i * (i + 1) / 2;
}
__global__ void __launch_bounds__(256) k_passB_reduce(const PassBReduceArgs r) {
    const int item = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= r.n_items) return;
    const int c0 = r.start[item], c1 = r.start[item + 1], kind = r.kind[item];
    const int ea = r.ea[item], eb = r.eb[item];
    const int nv = kind == 0 ? (r.intr ? 65 : 27) : (kind == 1 ? 27 : (r.intr ? 60 : 36));
    for (int v = lane; v < nv; v += 64) {
        int src, i, j;
        double *dst;
        if (kind == 0) {   // camera ea: U_cc, g_c (+ with intrinsics U_kk, W_kc, g_k of its intrinsics entity)
            const int rc = 6 * ea, rk = 6 * (r.k_ent0 + ea);
            if (v < 21) { src = v; unpack_lower(v, i, j); dst = r.U0 + (size_t)(rc + i) * r.n_pad + rc + j; }
            else if (v < 27) { src = 78 + (v - 21); dst = r.g0 + rc + (v - 21); }
            else if (v < 37) { src = 90 + (v - 27); unpack_lower(v - 27, i, j); dst = r.U0 + (size_t)(rk + i) * r.n_pad + rk + j; }
            else if (v < 61) { src = 90 + 10 + (v - 37); i = (v - 37) / 6; j = (v - 37) % 6; dst = r.U0 + (size_t)(rk + i) * r.n_pad + rc + j; }
            else { src = 90 + 58 + (v - 61); dst = r.g0 + rk + (v - 61); }
        } else if (kind == 1) {   // marker entity ea: U_mm, g_m
            const int rm = 6 * ea;
            if (v < 21) { src = 21 + v; unpack_lower(v, i, j); dst = r.U0 + (size_t)(rm + i) * r.n_pad + rm + j; }
            else { src = 84 + (v - 21); dst = r.g0 + rm + (v - 21); }
        } else {   // pair (camera ea, marker entity eb): W_cm (+ W_km)
            const int rc = 6 * ea, rm = 6 * eb, rk = 6 * (r.k_ent0 + ea);
            if (v < 36) { src = 42 + v; i = v / 6; j = v % 6; dst = r.U0 + (size_t)(rm + j) * r.n_pad + rc + i; }
            else { src = 90 + 34 + (v - 36); i = (v - 36) / 6; j = (v - 36) % 6; dst = r.U0 + (size_t)(rk + i) * r.n_pad + rm + j; }
        }
        double s = 0.0;
        int k = c0;
        for (; k + 8 <= c1; k += 8) {   // eight records in flight; the additions keep their order
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = r.part[(size_t)r.chunk[k + u] * r.stride + src];
#pragma unroll
            for (int u = 0; u < 8; u++) s += t[u];
        }
        for (; k < c1; k++) s += r.part[(size_t)r.chunk[k] * r.stride + src];
        *dst = s;
    }
}

template <int BLOCK, int CPL, bool WR>
__global__ void __launch_bounds__(BLOCK) k_passA_intr(const PassAArgs a) {
    extern __shared__ double lds[];
    if (WR) passA_wrench_body<BLOCK, CPL, true>(a, lds, (int)blockIdx.x, (int)gridDim.x);
    else passA_body<BLOCK, CPL, true>(a, lds, (int)blockIdx.x, (int)gridDim.x);
}

template <int BLOCK, int CPL, bool WR, bool WRB = WR>
__global__ void __launch_bounds__(BLOCK) k_passAB(const PassAArgs a, const PassBArgs b) {
    extern __shared__ double lds[];   // pass A's layout; pass B uses the first (BLOCK / 64) * 2048 doubles
    if ((int)blockIdx.x < a.F) {
        if (WR) passA_wrench_body<BLOCK, CPL>(a, lds, (int)blockIdx.x, a.F);
        else passA_body<BLOCK, CPL>(a, lds, (int)blockIdx.x, a.F);
    } else if (WRB) passB_wrench_body(b, lds, ((int)blockIdx.x - a.F) * (BLOCK / 64));
    else passB_body(b, lds, ((int)blockIdx.x - a.F) * (BLOCK / 64));
}
// the same capped at 256 registers (two wavefronts per SIMD; pass B's corner loop not unrolled): long sequences, whose frame workgroups would otherwise take
// more than one round of the chip's wavefront slots (AAR_PASSAB_OCC2; profiles/r05_attempts.txt section 6)
template <int BLOCK, int CPL>
__global__ void __launch_bounds__(BLOCK, 2) k_passAB_o2(const PassAArgs a, const PassBArgs b) {
    extern __shared__ double lds[];
    if ((int)blockIdx.x < a.F) passA_wrench_body<BLOCK, CPL>(a, lds, (int)blockIdx.x, a.F);
    else passB_body<false, true>(b, lds, ((int)blockIdx.x - a.F) * (BLOCK / 64));
}
// the same with camera intrinsics optimised: pass A with the W_kf blocks, pass B, and pass B's intrinsics blocks (three launches before)
template <int BLOCK, int CPL, bool WR>
__global__ void __launch_bounds__(BLOCK) k_passAB_intr(const PassAArgs a, const PassBArgs b, int nb) {
    extern __shared__ double lds[];
    const int blk = (int)blockIdx.x;
    if (blk < a.F) {
        if (WR) passA_wrench_body<BLOCK, CPL, true>(a, lds, blk, a.F);
        else passA_body<BLOCK, CPL, true>(a, lds, blk, a.F);
    }
    else if (blk < a.F + nb) passB_body(b, lds, (blk - a.F) * (BLOCK / 64));
    else passB_intr_body(b, lds, (blk - a.F - nb) * (BLOCK / 64));
}

// ------------------------------------------------------------------------------------------------
// max over the diagonal of J^T J restricted to free parameters (mu_0 = tau * max, libs/sparselevmarq.h:369-377)
__global__ void __launch_bounds__(256) k_maxdiag(const double *__restrict__ U0, int n_pad, int A,
                                                 const int32_t *__restrict__ ent_fixed, const double *__restrict__ V,
                                                 int F, int frames_fixed, double *__restrict__ out) {
    double m = -1.7976931348623157e308;
    for (int i = threadIdx.x; i < 6 * A; i += blockDim.x)
        if (!ent_fixed[i / 6]) m = fmax(m, U0[(size_t)i * n_pad + i]);
    if (!frames_fixed)   // eight loads in flight per thread (one workgroup walks all 6 F diagonal entries)
        for (int i0 = threadIdx.x; i0 < 6 * F; i0 += 8 * (int)blockDim.x) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * (int)blockDim.x;
                v[u] = i < 6 * F ? V[(size_t)(i / 6) * 36 + (i % 6) * 7] : m;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) m = fmax(m, v[u]);
        }
    __shared__ double wm[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmax(fmax(wm[0], wm[1]), fmax(wm[2], wm[3]));
}

// ------------------------------------------------------------------------------------------------
void launch_unpack(const DeviceProblem &P, int which, hipStream_t st, int zero_blk) {
    const int n_ent = P.A + P.F;
    double *zp = nullptr, *zq = nullptr;
    int64_t zn = 0, zqn = 0;
    int blocks = (n_ent + 255) / 256;
    if (zero_blk >= 0) {   // S | rhs | g0 | tail are one allocation; the linear-model partials another
        zp = P.blk[zero_blk].S; zn = (int64_t)P.n_pad * P.n_pad + 2 * (int64_t)P.n_pad;
        zq = P.lin_part; zqn = 2 * (int64_t)(P.F + 1);
        blocks = (int)std::max<int64_t>(blocks, std::min<int64_t>(512, (zn + 2047) / 2048));
    }
    { HookScope _h(P, KID_UNPACK); hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(256), 0, st, P.z[which], P.ent[which], n_ent,
                                                      P.C + P.M, P.intr ? P.C + P.M + P.C : P.C + P.M, zp, zn, zq, zqn); }
}

void launch_residual(const DeviceProblem &P, int which, double *r_out, hipStream_t st) {
    const int blocks = (int)((P.N + 255) / 256);
    if (blocks == 0) return;
    const KTable kt = k_table(P, which);
    { HookScope _h(P, KID_RESIDUAL); hipLaunchKernelGGL(k_residual, dim3(blocks), dim3(256), 0, st, P.a_idx, P.a_uv, P.ent[which], kt.base, kt.stride, P.N, P.A,
                       P.half_size, P.res_f32, P.huber, r_out, P.err_part); }
}

int residual_blocks(const DeviceProblem &P) { return (int)((P.N + 255) / 256); }

size_t passA_wrench_lds_bytes(int max_kf, bool intr) { return (passA_h_doubles(max_kf, intr ? HLS_INTR : HLS) + 64 + 24 + 72) * sizeof(double); }

size_t passA_lds_bytes(int max_kf, int block) {
    return (passA_w_doubles(max_kf) + 64 + (size_t)(max_kf + 1) * ENT_LDS + (block / 64) * passA_sum_chunk(block) * 64) * sizeof(double);
}

static PassAArgs passA_args(const DeviceProblem &P, int which, double mu_pred, int zero_blk) {
    PassAArgs a;
    a.idx = P.a_idx; a.uv = P.a_uv; a.ent = P.ent[which];
    { const KTable kt = k_table(P, which); a.Kmat = kt.base; a.kstride = kt.stride; }
    a.frame_obs_start = P.frame_obs_start; a.fslot_start = P.fslot_start; a.fslot_ent = P.fslot_ent; a.frame_stride = P.frame_stride;
    a.stamps = nullptr; a.k_ent0 = P.intr ? P.C + P.M : P.A;
#ifdef AAR_PASSA_STAMPS
    {
        static unsigned long long *st = nullptr;
        if (!st && getenv("AAR_STAMPS_A")) {
            (void)hipMalloc(&st, 512 * 16 * 8); (void)hipMemset(st, 0, 512 * 16 * 8);
            static unsigned long long *keep = st;
            atexit([] { std::vector<unsigned long long> h(512 * 16); (void)d2h(h.data(), keep, 512 * 16 * 8, nullptr);
                        FILE *fp = fopen(getenv("AAR_STAMPS_A"), "w"); for (int i = 0; i < 512; i++) { for (int j = 0; j < 16; j++) fprintf(fp, "%llu ", h[i * 16 + j]); fprintf(fp, "\n"); } fclose(fp); });
        }
        a.stamps = st;
    }
#endif
    a.A = P.A; a.F = P.F; a.C = P.C; a.res_f32 = P.res_f32; a.max_kf = P.max_kf; a.frames_fixed = P.frames_fixed;
    a.huber = P.huber;
    a.h = P.half_size; a.mu_pred = mu_pred;
    const DeviceProblem::Blocks &b = P.blk[which];
    a.V = b.V; a.gf = b.gf; a.W = b.W; a.Wf = b.Wf; a.Vinv = b.Vinv; a.hf = b.hf; a.err_part = P.err_part;
    if (b.Wf && !P.want_w64) a.W = nullptr;   // PCG with the fp32 operator: nobody reads the fp64 blocks (175 MB per pass at config 5) unless the dense-output API asks
    a.zero0 = a.zero1 = a.zero2 = nullptr; a.zero0_n = a.zero1_n = a.zero2_n = 0;
    if (zero_blk >= 0) {
        const DeviceProblem::Blocks &zb = P.blk[zero_blk];
        a.zero0 = zb.S; a.zero0_n = (int64_t)P.n_pad * P.n_pad;
        a.zero1 = zb.rhs; a.zero1_n = P.n_pad;
        a.zero2 = zb.g0; a.zero2_n = P.n_pad;
    }
    a.flags = P.flags;
    a.Wd = a.Yd = nullptr; a.slot_dense = nullptr; a.Ad = 0;
    if (P.n_smwork > 0 && P.dense_from_passA) { a.Wd = P.Wd; a.Yd = P.Yd; a.slot_dense = P.slot_dense; a.Ad = P.Ad; }
    return a;
}

static PassBArgs passB_args(const DeviceProblem &P, int which) {
    PassBArgs b;
    b.idx = P.b_idx; b.uv = P.b_uv; b.ent = P.ent[which]; b.chunk_start = P.chunk_start;
    { const KTable kt = k_table(P, which); b.Kmat = kt.base; b.kstride = kt.stride; }
    b.k_ent0 = P.C + P.M;
    b.n_chunks = P.n_chunks; b.A = P.A; b.res_f32 = P.res_f32; b.n_pad = P.n_pad; b.huber = P.huber; b.h = P.half_size;
    b.U0 = P.blk[which].S; b.g0 = P.blk[which].g0;
    b.part = P.pb_part; b.part_stride = P.pb_stride;
    return b;
}

// with_b: pass B's chunks ride in the same launch (the caller must not launch pass B again)
template <int B, int CPL>
static void launch_passA_t(const DeviceProblem &P, const PassAArgs &a, const PassBArgs *pbargs, hipStream_t st) {
    size_t lds = P.tune.passA_wrench ? passA_wrench_lds_bytes(P.max_kf, P.intr != 0) : passA_lds_bytes(P.max_kf, B);
    if (pbargs) lds = std::max(lds, (size_t)(B / 64) * 2048 * sizeof(double));   // pass B's wave-sum scratch, when its chunks ride along
    static size_t granted = 48 * 1024, granted_ab = 48 * 1024;
    HookScope _h(P, KID_PASSA);
    if (P.intr && pbargs) {
        static size_t granted_abi = 48 * 1024;
        const int nb = (P.n_chunks + B / 64 - 1) / (B / 64);
        if (P.tune.passA_wrench) {
            static size_t granted_abiw = 48 * 1024;
            allow_dynamic_lds(reinterpret_cast<const void *>(k_passAB_intr<B, CPL, true>), lds, granted_abiw);
            hipLaunchKernelGGL((k_passAB_intr<B, CPL, true>), dim3(P.F + 2 * nb), dim3(B), lds, st, a, *pbargs, nb);
        } else {
            allow_dynamic_lds(reinterpret_cast<const void *>(k_passAB_intr<B, CPL, false>), lds, granted_abi);
            hipLaunchKernelGGL((k_passAB_intr<B, CPL, false>), dim3(P.F + 2 * nb), dim3(B), lds, st, a, *pbargs, nb);
        }
    } else if (P.intr) {
        static size_t granted_i = 48 * 1024;
        if (P.tune.passA_wrench) {
            static size_t granted_iw = 48 * 1024;
            allow_dynamic_lds(reinterpret_cast<const void *>(k_passA_intr<B, CPL, true>), lds, granted_iw);
            hipLaunchKernelGGL((k_passA_intr<B, CPL, true>), dim3(P.F), dim3(B), lds, st, a);
        } else {
            allow_dynamic_lds(reinterpret_cast<const void *>(k_passA_intr<B, CPL, false>), lds, granted_i);
            hipLaunchKernelGGL((k_passA_intr<B, CPL, false>), dim3(P.F), dim3(B), lds, st, a);
        }
    } else if (pbargs && P.tune.passA_wrench && !P.tune.passB_wrench_merged &&
               (P.tune.passAB_occ2 >= 0 ? P.tune.passAB_occ2 != 0 : P.F + (P.n_chunks + B / 64 - 1) / (B / 64) > P.n_cus * 4 / (B / 64))) {
        // more workgroups than the chip holds at one wavefront per SIMD (388 registers): at two the frames of a long sequence run in one round instead of two
        // (config 4, 2000 frames: 36.3 -> 32.6 us; config 3's 500 frames fit anyway and keep the unrolled pass B)
        static size_t granted_o2 = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passAB_o2<B, CPL>), lds, granted_o2);
        hipLaunchKernelGGL((k_passAB_o2<B, CPL>), dim3(P.F + (P.n_chunks + B / 64 - 1) / (B / 64)), dim3(B), lds, st, a, *pbargs);
    } else if (pbargs && P.tune.passA_wrench && !P.tune.passB_wrench_merged) {   // (the default: pass A in wrench form, pass B's chunks in row form)
        static size_t granted_abr = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passAB<B, CPL, true, false>), lds, granted_abr);
        hipLaunchKernelGGL((k_passAB<B, CPL, true, false>), dim3(P.F + (P.n_chunks + B / 64 - 1) / (B / 64)), dim3(B), lds, st, a, *pbargs);
    } else if (pbargs && P.tune.passA_wrench) {
        static size_t granted_abw = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passAB<B, CPL, true>), lds, granted_abw);
        hipLaunchKernelGGL((k_passAB<B, CPL, true>), dim3(P.F + (P.n_chunks + B / 64 - 1) / (B / 64)), dim3(B), lds, st, a, *pbargs);
    } else if (pbargs) {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passAB<B, CPL, false>), lds, granted_ab);
        hipLaunchKernelGGL((k_passAB<B, CPL, false>), dim3(P.F + (P.n_chunks + B / 64 - 1) / (B / 64)), dim3(B), lds, st, a, *pbargs);
    } else if (P.tune.passA_wrench) {
        static size_t granted_w = 48 * 1024;
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passA<B, CPL, true>), lds, granted_w);
        hipLaunchKernelGGL((k_passA<B, CPL, true>), dim3(P.F), dim3(B), lds, st, a);
    } else {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_passA<B, CPL, false>), lds, granted);
        hipLaunchKernelGGL((k_passA<B, CPL, false>), dim3(P.F), dim3(B), lds, st, a);
    }
}

static void launch_passA_any(const DeviceProblem &P, const PassAArgs &a, const PassBArgs *pbargs, hipStream_t st) {
    // one wavefront per frame up to ~96 observations per frame (four / two lanes per observation while they fit in it),
    // two wavefronts above
    const double avg = (double)P.N / (double)P.F;
    if (P.deterministic && avg > 96) return launch_passA_t<64, 4>(P, a, pbargs, st);   // ONE wavefront per frame: its LDS additions come in program order
    const int var = P.tune.passA_variant;   // experiments (AAR_PASSA_VARIANT): 1281 / 1282 / 1284 / 2564
    if (var == 641) return launch_passA_t<64, 1>(P, a, pbargs, st);
    if (var == 642) return launch_passA_t<64, 2>(P, a, pbargs, st);
    if (var == 644) return launch_passA_t<64, 4>(P, a, pbargs, st);
    if (var == 1281) return launch_passA_t<128, 1>(P, a, pbargs, st);
    if (var == 1282) return launch_passA_t<128, 2>(P, a, pbargs, st);
    if (var == 1284) return launch_passA_t<128, 4>(P, a, pbargs, st);
    if (var == 2564) return launch_passA_t<256, 4>(P, a, pbargs, st);
    if (avg <= 14) launch_passA_t<64, 1>(P, a, pbargs, st);
    else if (avg <= 30) launch_passA_t<64, 2>(P, a, pbargs, st);
    else if (avg <= 96) launch_passA_t<64, 4>(P, a, pbargs, st);
    else launch_passA_t<128, 4>(P, a, pbargs, st);
}

void launch_passA(const DeviceProblem &P, int which, double mu_pred, int zero_blk, hipStream_t st) {
    if (P.F == 0) return;
    launch_passA_any(P, passA_args(P, which, mu_pred, zero_blk), nullptr, st);
}

void launch_passB(const DeviceProblem &P, int which, hipStream_t st) {
    if (P.n_chunks == 0) return;
    if (P.deterministic) {   // per-chunk records, then their fixed-order sums (stored: the block set is clear)
        HookScope _h(P, KID_PASSB);
        const PassBArgs b = passB_args(P, which);
        if (P.tune.passA_wrench) hipLaunchKernelGGL(k_passB_det<true>, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, b);
        else hipLaunchKernelGGL(k_passB_det<false>, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, b);
        if (P.intr) hipLaunchKernelGGL(k_passB_intr_det, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, b);
        PassBReduceArgs r;
        r.start = P.pbr_start; r.chunk = P.pbr_chunk; r.kind = P.pbr_kind; r.ea = P.pbr_a; r.eb = P.pbr_b; r.part = P.pb_part;
        r.stride = P.pb_stride; r.n_items = P.n_pbr; r.n_pad = P.n_pad; r.intr = P.intr; r.k_ent0 = P.C + P.M; r.U0 = b.U0; r.g0 = b.g0;
        hipLaunchKernelGGL(k_passB_reduce, dim3((P.n_pbr + 3) / 4), dim3(256), 0, st, r);
        return;
    }
    {
        HookScope _h(P, KID_PASSB);
        if (P.tune.passB_lean == 1) hipLaunchKernelGGL(k_passB_lean2, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, passB_args(P, which));
        else if (P.tune.passB_lean == 2) hipLaunchKernelGGL(k_passB_lean1, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, passB_args(P, which));
        else if (P.tune.passA_wrench) hipLaunchKernelGGL(k_passB<true>, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, passB_args(P, which));
        else hipLaunchKernelGGL(k_passB<false>, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, passB_args(P, which));
    }
    if (P.intr) { HookScope _h(P, KID_PASSB); hipLaunchKernelGGL(k_passB_intr, dim3((P.n_chunks + 3) / 4), dim3(256), 0, st, passB_args(P, which)); }
}

bool launch_passAB(const DeviceProblem &P, int which, double mu_pred, int zero_blk, hipStream_t st) {
    if (P.F == 0 || P.n_chunks == 0 || P.deterministic) return false;   // nothing to merge (or the deterministic variants): the caller launches what there is
    // Side by side pays while the two passes together are a few wavefronts per SIMD (configs 2-4: -45 % / -11 % of their
    // summed time at configs 3 / 4); once either fills the chip on its own (config 5: +7 %, pass B's workgroups then carry
    // pass A's LDS allocation) they go one after the other
    const int64_t waves = (int64_t)P.F * ((double)P.N / (double)P.F <= 96 ? 1 : 4) + (P.intr ? 2 : 1) * (int64_t)P.n_chunks;
    if (waves > 4096) return false;
    const PassBArgs b = passB_args(P, which);
    launch_passA_any(P, passA_args(P, which, mu_pred, zero_blk), &b, st);
    return true;
}

void launch_maxdiag(const DeviceProblem &P, int which, hipStream_t st) {
    { HookScope _h(P, KID_MAXDIAG); hipLaunchKernelGGL(k_maxdiag, dim3(1), dim3(256), 0, st, P.blk[which].S, P.n_pad, P.A, P.ent_fixed, P.blk[which].V, P.F,
                       P.frames_fixed, P.scal + 4); }
}

}  // namespace aar

// ================================================================================================
// track(): MultiCamMapper::track (libs/multicam_mapper.cpp:430-443) for a batch of frames.  Cameras and markers are
// fixed; every frame refines its own 6-DoF object pose with the same LM rules (SparseLevMarq::solve(z, f),
// libs/sparselevmarq.h:223-228,440-472) on error_function_tracking (:678-729: residuals kept in double, optional Huber
// weights with a fixed delta).  One wavefront runs the WHOLE loop of one frame on the device: lanes stride over the
// frame's observations, the 6x6 normal equations are reduced with wave shuffles, every lane solves the damped system
// redundantly, no host round trip.  The Jacobian is the analytic 2x6 frame block instead of calcDerivates_omp's central
// differences (:165-196).
// ================================================================================================
namespace aar {


struct TrackArgs {
    const ObsIdx *idx; const float *uv; const double *ent; const double *Kmat; const int32_t *frame_obs_start;
    int kstride;
    int A, F; float huber; double h;
    int max_iters; double min_error, min_step_error_diff, min_average_step_error_diff, tau;
    double *z;            // [6(A+F)]: frame poses in/out
    int32_t *iters_out;   // [F]
    double *err_out;      // [F]
};

// residual sum (and optionally V (21 packed), g (6)) of one frame at pose zf, summed over the wave
template <bool WITH_J>
__device__ __forceinline__ double track_eval(const TrackArgs &a, int f, const double zf[6], int lane, double V[21], double g[6]) {
    double row[ENT_STRIDE];
    make_ent_row(zf, row);
    Ent ef;
#pragma unroll
    for (int i = 0; i < 9; i++) { ef.R[i] = row[i]; ef.Jl[i] = row[12 + i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) ef.t[i] = row[9 + i];
    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = 0.0;
    const int o0 = a.frame_obs_start[f], o1 = a.frame_obs_start[f + 1];
    for (int o = o0 + lane; o < o1; o += 64) {
        const ObsIdx id = a.idx[o];
        const float4 uv0 = reinterpret_cast<const float4 *>(a.uv)[2 * (int64_t)o];
        const float4 uv1 = reinterpret_cast<const float4 *>(a.uv)[2 * (int64_t)o + 1];
        const float ou[8] = {uv0.x, uv0.y, uv0.z, uv0.w, uv1.x, uv1.y, uv1.z, uv1.w};
        Ent ec, em;
        load_ent(a.ent, id.cam, ec);
        load_ent(a.ent, id.marker, em);
        double K[9];
#pragma unroll
        for (int i = 0; i < 9; i++) K[i] = a.Kmat[a.kstride * id.cam + i];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CornerGeom gm;
            project_corner(ec, em, ef, K, a.h, k, gm);
            double r[2];
            corner_residual(ou[2 * k], ou[2 * k + 1], gm.u, gm.v, 0, -1.f, r[0], r[1]);  // double residuals (:712-713), unweighted
            // Huber: track() differentiates the WEIGHTED error function numerically (calcDerivates), so the weight's own
            // derivative belongs to the Jacobian here (unlike solve(), whose Jacobian ignores the weights):
            //   r_w = w r,  w = sqrt(rho)/s,  s = |r|,  rho = 2 delta s - delta^2  (outliers; w = 1 otherwise)
            //   d r_w = w dr + r (dw/ds) (r . dr)/s
            double w = 1.0, cw = 0.0;
            if (a.huber >= 0.f) {
                const double e = r[0] * r[0] + r[1] * r[1];
                const float dsq = a.huber * a.huber, d2 = 2 * a.huber;
                if (e != 0.0 && e > (double)dsq) {
                    const double sn = sqrt(e), rho = (double)d2 * sn - (double)dsq, sr = sqrt(rho);
                    w = sr / sn;
                    cw = ((double)a.huber / sr - sr / sn) / e;  // (dw/ds) / s
                }
            }
            acc[27] += w * w * (r[0] * r[0] + r[1] * r[1]);
            if (WITH_J) {
                double Gc[2][6], Gm[2][6], Gf[2][6];
                corner_jacobian<false, false, true>(ec, em, ef, K, gm, Gc, Gm, Gf);
                if (cw != 0.0 || w != 1.0) {
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        const double rg = r[0] * Gf[0][i] + r[1] * Gf[1][i];
                        Gf[0][i] = w * Gf[0][i] + cw * r[0] * rg;
                        Gf[1][i] = w * Gf[1][i] + cw * r[1] * rg;
                    }
                }
                r[0] *= w;
                r[1] *= w;
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        acc[21 + i] += Gf[rr][i] * r[rr];
#pragma unroll
                        for (int j = 0; j <= i; j++) acc[i * (i + 1) / 2 + j] += Gf[rr][i] * Gf[rr][j];
                    }
            }
        }
    }
    if (WITH_J) {
#pragma unroll
        for (int i = 0; i < 21; i++) V[i] = wave_sum(acc[i]);
#pragma unroll
        for (int i = 0; i < 6; i++) g[i] = wave_sum(acc[21 + i]);
    }
    return wave_sum(acc[27]);
}

__global__ void __launch_bounds__(256) k_track(const TrackArgs a) {
    const int lane = threadIdx.x & 63, f = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (f >= a.F) return;
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; i++) z[i] = a.z[6 * (size_t)(a.A + f) + i];
    const double rows = 8.0 * (double)(a.frame_obs_start[f + 1] - a.frame_obs_start[f]);
    double V[21], g[6];
    // init (:238-249): the first evaluation also yields the first step's normal equations
    double currErr = track_eval<true>(a, f, z, lane, V, g), prevErr = currErr;
    double mu = -1.0, v = 2.0;
    int mustExit = 0, iters = 0;
    for (int it = 0; it < a.max_iters && !mustExit && rows > 0; it++) {
        if (it > 0) (void)track_eval<true>(a, f, z, lane, V, g);  // J, Jt*J, B at curr_z (:353-367)
        if (mu < 0) {                                                 // :369-377
            double mx = V[0];
#pragma unroll
            for (int i = 1; i < 6; i++) mx = fmax(mx, V[i * (i + 1) / 2 + i]);
            mu = mx * a.tau;
        }
        double gain = 0.0;
        int ntries = 0;
        bool accepted = false;
        do {
            double m[6][6], inv[36], d[6], zt[6];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) m[i][j] = V[sym6(i, j)] + (i == j ? mu : 0.0);
            (void)spd6_inverse(m, inv);
            double d2 = 0.0, dg = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < 6; j++) s += inv[i * 6 + j] * g[j];
                d[i] = s;
                zt[i] = z[i] + s;
                d2 += s * s;
                dg += s * g[i];
            }
            double dummyV[21], dummyg[6];
            const double err = track_eval<false>(a, f, zt, lane, dummyV, dummyg);
            const double Lq = 0.5 * (mu * d2 - dg);                  // :406
            gain = (err - prevErr) / Lq;
            if (gain > 0 && (err - prevErr) < 0) {                   // :409-415
                const double t = 2 * gain - 1;
                mu = mu * fmax(0.33, 1.0 - t * t * t);
                v = 2.0;
                currErr = err;
#pragma unroll
                for (int i = 0; i < 6; i++) z[i] = zt[i];
                accepted = true;
            } else {
                mu = mu * v;
                v = v * 5;
            }
        } while (gain <= 0 && ntries++ < 5 && !accepted);
        if (currErr < a.min_error) mustExit = 1;                      // :458-461
        if (fabs(prevErr - currErr) <= a.min_step_error_diff || fabs((prevErr - currErr) / rows) <= a.min_average_step_error_diff || !accepted)
            mustExit = 2;
        if (currErr > prevErr) mustExit = 3;
        iters++;
        prevErr = currErr;
    }
    if (lane < 6) a.z[6 * (size_t)(a.A + f) + lane] = z[lane];
    if (lane == 0) { a.iters_out[f] = iters; a.err_out[f] = currErr; }
}

void launch_track(const DeviceProblem &P, int which, int max_iters, double min_error, double min_step, double min_avg, double tau,
                  int32_t *iters_out, double *err_out, hipStream_t st) {
    if (P.F == 0) return;
    TrackArgs a;
    a.idx = P.a_idx; a.uv = P.a_uv; a.ent = P.ent[which]; a.frame_obs_start = P.frame_obs_start;
    { const KTable kt = k_table(P, which); a.Kmat = kt.base; a.kstride = kt.stride; }
    a.A = P.A; a.F = P.F; a.huber = P.huber; a.h = P.half_size;
    a.max_iters = max_iters; a.min_error = min_error; a.min_step_error_diff = min_step; a.min_average_step_error_diff = min_avg; a.tau = tau;
    a.z = P.z[which]; a.iters_out = iters_out; a.err_out = err_out;
    hipLaunchKernelGGL(k_track, dim3((P.F + 3) / 4), dim3(256), 0, st, a);
}

}  // namespace aar
