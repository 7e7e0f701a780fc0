// Shared by solve_kernels.hip (k_backsub, the riders of the last LDL^T tile) and spcg_kernels.hip (the riders of k_spcg): the frame
// back-substitution of a damped try and the fence-free hand-over it waits on.
#pragma once
#include "geom.hpp"
#include "kernels.h"

namespace aar {

// Hand-over of a small vector between workgroups of one launch WITHOUT cache-maintenance fences.  A release / acquire pair at agent
// scope costs a write-back of the XCD's L2 plus an invalidate: 4.3-5.4 us per hop from a 1024-thread workgroup, measured
// (scripts/probe/hop_probe.hip) -- more than a kernel boundary (2.9 us).  When the payload itself travels as relaxed agent-scope
// atomic stores / loads (sc1: written through and read past the XCD's L2) nothing needs to be flushed: the writer drains its own
// stores (s_waitcnt) and raises the flag, the reader polls the flag and then loads: 1.1 us per hop for up to 8 KB.
__device__ __forceinline__ void hop_publish(int32_t *flag, int value, bool leader) {   // whole wavefront; its payload stores were atomic (agent)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (leader) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool hop_wait(const int32_t *flag, int value) {   // one thread; false = gave up (the caller raises an error flag)
    long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 24)) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}

// ------------------------------------------------------------------------------------------------
// Frame back-substitution, one wavefront per frame: delta_f = (V_f+mu I)^-1 (g_f - sum_a W_af^T delta_a);
// z_trial = z_cur + delta; per-frame pieces of L = 0.5 delta^T (mu delta - B) (libs/sparselevmarq.h:406).
// The last workgroup updates the shared (camera / marker) parameters.
// Runs as a kernel of its own (k_backsub) or as extra workgroups of the LAST diagonal tile's launch (systems of up to three tiles):
// there the workgroups fetch everything that does not depend on delta_s -- the frame's W blocks, g_f, V_f^-1 -- while the tile is
// still being factored next door, wait for ONE flag (hop_wait: delta_s travels as agent-scope atomics) and only then gather
// delta_s: a 1.1 us hand-over instead of a kernel boundary and a cold fetch.
// ------------------------------------------------------------------------------------------------
struct BacksubArgs {
    const int32_t *fslot_start, *fslot_ent;
    const double *W, *Vinv, *gf, *g0, *delta_s, *zc;
    const float *Wf;            // PCG with the fp32 operator: pass A's fp32 copy of W (kernels.h, Blocks::Wf) instead of W -- what the solve itself has read; nullptr otherwise
    double *zt;
    int A, F, n_frame_blocks;
    double *lin_part, *ent_out;
    int k_ent0;
};

// blk = workgroup index among the back-substitution workgroups; wave-uniform `wave`, `nwaves` wavefronts per workgroup;
// flag != nullptr: delta_s is not in memory before flag[0] == epoch
__device__ __forceinline__ void backsub_body(const BacksubArgs &b, int blk, double *red, const int32_t *flag, int epoch, int32_t *err_flags) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;   // wave-uniform: the frame's slot range, g_f, V_f^-1 through the scalar cache
    auto dl = [&](int i) -> double {   // an entry of delta_s: past the L2 when it has just been written by another workgroup of this launch
        return flag ? __hip_atomic_load(b.delta_s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : b.delta_s[i];
    };
    auto wait = [&]() {
        if (!flag) return;
        if (tid == 0 && !hop_wait(flag, epoch)) atomicOr(err_flags, 4);
        __syncthreads();
    };
    if (blk == b.n_frame_blocks) {  // shared part
        wait();
        double d2 = 0.0, dg = 0.0;
        for (int i = tid; i < 6 * b.A; i += blockDim.x) {
            const double d = dl(i);
            b.zt[i] = b.zc[i] + d;
            d2 += d * d;
            dg += d * b.g0[i];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { d2 += __shfl_xor(d2, off); dg += __shfl_xor(dg, off); }
        if (lane == 0) { red[wave] = d2; red[16 + wave] = dg; }
        __syncthreads();
        if (tid == 0) {
            double s2 = 0.0, sg = 0.0;
            for (int w = 0; w < nwaves; w++) { s2 += red[w]; sg += red[16 + w]; }
            b.lin_part[2 * (size_t)b.F] = s2;
            b.lin_part[2 * (size_t)b.F + 1] = sg;
        }
        // {R, t, J_l} rows of the shared entities at the trial point: both observation passes of the trial evaluation read
        // them from this table (so they need not run one after the other)
        __threadfence_block();
        __syncthreads();
        for (int e = tid; e < b.A; e += blockDim.x) {   // entities from k_ent0 on are intrinsics entities: their row is the camera matrix
            if (e >= b.k_ent0) make_k_row(b.zt + 6 * (size_t)e, b.ent_out + (size_t)e * ENT_STRIDE);
            else make_ent_row(b.zt + 6 * (size_t)e, b.ent_out + (size_t)e * ENT_STRIDE);
        }
        return;
    }
    const int f = blk * nwaves + wave;
    const bool live = f < b.F;
    const int s0 = live ? b.fslot_start[f] : 0, s1 = live ? b.fslot_start[f + 1] : 0;
    // first pass of the slot list (all of it for frames with up to 64 cameras+markers): W block and entity BEFORE the wait
    const int sl = s0 + lane;
    const bool has = sl < s1;
    int a0 = 0;
    double2 w0[18];
    if (has) {
        a0 = b.fslot_ent[sl];
        if (b.Wf) {
            const float4 *wf = reinterpret_cast<const float4 *>(b.Wf + (size_t)s0 * 36) + lane;
            const int kf = s1 - s0;
#pragma unroll
            for (int q = 0; q < 9; q++) {
                const float4 v = wf[(size_t)q * kf];
                w0[2 * q] = make_double2(v.x, v.y); w0[2 * q + 1] = make_double2(v.z, v.w);
            }
        } else {
            const double2 *wb = reinterpret_cast<const double2 *>(b.W + (size_t)sl * 36);
#pragma unroll
            for (int q = 0; q < 18; q++) w0[q] = wb[q];
        }
    }
    double g[6], vrow[6], zc6 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) { g[i] = live ? b.gf[(size_t)f * 6 + i] : 0.0; vrow[i] = (live && lane < 6) ? b.Vinv[(size_t)f * 36 + lane * 6 + i] : 0.0; }
    if (live && lane < 6) zc6 = b.zc[(size_t)6 * (b.A + f) + lane];
    wait();
    if (!live) return;
    double c[6] = {0, 0, 0, 0, 0, 0};
    if (has) {
        double da[6];
#pragma unroll
        for (int i = 0; i < 6; i++) da[i] = dl(6 * a0 + i);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            c[0] += w0[3 * i].x * da[i]; c[1] += w0[3 * i].y * da[i]; c[2] += w0[3 * i + 1].x * da[i];
            c[3] += w0[3 * i + 1].y * da[i]; c[4] += w0[3 * i + 2].x * da[i]; c[5] += w0[3 * i + 2].y * da[i];
        }
    }
    for (int s = s0 + lane + 64; s < s1; s += 64) {
        const int a = b.fslot_ent[s];
        double2 wt[18];
        if (b.Wf) {
            const float4 *wf = reinterpret_cast<const float4 *>(b.Wf + (size_t)s0 * 36) + (s - s0);
            const int kf = s1 - s0;
#pragma unroll
            for (int q = 0; q < 9; q++) {
                const float4 v = wf[(size_t)q * kf];
                wt[2 * q] = make_double2(v.x, v.y); wt[2 * q + 1] = make_double2(v.z, v.w);
            }
        } else {
            const double2 *wb = reinterpret_cast<const double2 *>(b.W + (size_t)s * 36);
#pragma unroll
            for (int q = 0; q < 18; q++) wt[q] = wb[q];
        }
        double da[6];
#pragma unroll
        for (int i = 0; i < 6; i++) da[i] = dl(6 * a + i);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double2 x0 = wt[3 * i], x1 = wt[3 * i + 1], x2 = wt[3 * i + 2];
            c[0] += x0.x * da[i]; c[1] += x0.y * da[i]; c[2] += x1.x * da[i];
            c[3] += x1.y * da[i]; c[4] += x2.x * da[i]; c[5] += x2.y * da[i];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int i = 0; i < 6; i++) c[i] += __shfl_xor(c[i], off);
    double d = 0.0;
    if (lane < 6) {
#pragma unroll
        for (int k = 0; k < 6; k++) d += vrow[k] * (g[k] - c[k]);
        b.zt[(size_t)6 * (b.A + f) + lane] = zc6 + d;
    }
    {   // the frame's own {R, t, J_l} row at the trial point (lane 0; the six new parameters come from lanes 0..5)
        const double zn = (lane < 6) ? zc6 + d : 0.0;
        double zv[6];
#pragma unroll
        for (int i = 0; i < 6; i++) zv[i] = __shfl(zn, i);
        if (lane == 0) make_ent_row(zv, b.ent_out + (size_t)(b.A + f) * ENT_STRIDE);
    }
    double d2 = (lane < 6) ? d * d : 0.0;
    double dgv = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) dgv += (lane == i) ? d * g[i] : 0.0;
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { d2 += __shfl_xor(d2, off); dgv += __shfl_xor(dgv, off); }
    if (lane == 0) { b.lin_part[2 * (size_t)f] = d2; b.lin_part[2 * (size_t)f + 1] = dgv; }
}

// host side: the arguments of the back-substitution z[trial] = z[cur] + delta with `waves_per_block` wavefronts (= frames) per workgroup
BacksubArgs backsub_args(const DeviceProblem &P, int cur, int trial, int waves_per_block);

}  // namespace aar
