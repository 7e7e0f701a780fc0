// Host-callable launchers of the HIP kernels (defined in eval_kernels.hip / solve_kernels.hip).
// All pointers are device pointers unless noted.  Nothing here allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aar {

constexpr int CHOL_NB = 96;       // dense LDL^T tile (16 entity blocks of 6)
constexpr int PASSB_CHUNK = 1024;  // upper limit of AAR_PASSB_CHUNK (observations of one (camera, marker) run handled by one wavefront; the default rule stops at 512)
// CG on the explicit reduced system (spcg_kernels.hip): iteration cap (sizes the hand-over buffers: one per iteration plus the
// start-up and the final one), largest system (tiles of 96 unknowns: the six rows of an entity live in one wavefront's registers)
constexpr int SPCG_MAX_IT = 128, SPCG_BUFS = SPCG_MAX_IT + 2, SPCG_MAX_NT = 14;
// default cap by system size: up to four tiles the direct chain (<= 120 us) is cheaper than a CG solve of more than ~64 iterations; larger systems' chains cost 200-500 us
inline int spcg_default_cap(int nT) { return nT <= 4 ? 64 : SPCG_MAX_IT; }
// Default forcing terms of the inexact solvers (include/aar.h: aar_solver_options.pcg_eta; DESIGN.md section 12, profiles/r05_eta_pose_sweep.txt).  Chosen for the
// final POSES: with these the poses of a run agree with the direct solver's to ~2e-6 (rotation-matrix entries) / ~1e-6 m at configs 3-5 -- 100x closer than the
// direct path is to the reference-faithful CPU run (analytic against central-difference Jacobian), 1000x closer than the reference's own last LM step moves them.
// Measured, uniform eta -> largest pose difference against the direct run (config 3, SPCG): 0.02 -> 2e-4, 3e-3 -> 2e-5, 1e-3 -> 1e-5, 3e-4 -> 2e-6, 1e-4 -> 6e-7;
// (config 5, PCG): 0.1 -> 9e-5, 0.03 -> 2e-5, 0.01 -> 5e-6, 3e-3 -> 1e-6.  A loose-then-tight SEQUENCE buys nothing at equal cost: the early steps' errors along
// weakly determined directions are never corrected by the later ones (round 4's 0.1 -> 0.02 sequence: 6e-4), so the default is ONE forcing term.
constexpr int PCG_NY = 16;   // width of the PCG kernels' barrier tree (pcg_kernels.hip, grid_hop_tree): 256 workgroups on ONE address serialise at ~60 ns each: 15 us per hop
constexpr int PCG_NYV = 2;   // k_pcgf: partial y vectors (workgroup wg adds into vector wg % PCG_NYV; every reader adds them up)
constexpr int PCG_HOP_WORDS = (2 * PCG_NY + 1) * 16;   // k_pcgf's barrier tree: PCG_NY first-level counters, one second-level counter, PCG_NY flags (64 bytes apart)
constexpr double PCG_W32_MIN_ETA = 1e-4;   // k_pcgf reads the fp32 copy of W only at forcing terms from here up (pcg_kernels.hip, launch_pcg)
constexpr double SPCG_ABS_TOL_DEFAULT = 2e-5, PCG_ABS_TOL_DEFAULT = 5e-5;   // (measured: free for SPCG at configs 3-4; PCG at config 5: 5e-5 +3 % CG iterations, 2e-5 +15 %)
constexpr double SPCG_ETA_DEFAULT = 3e-4, SPCG_ETA_LOOSE_DEFAULT = 0.0, PCG_ETA_DEFAULT = 5e-3, PCG_ETA_LOOSE_DEFAULT = 0.0;
inline int spcg_stride(int n_pad) { return 8 * (n_pad / 6); }   // doubles per hand-over buffer: one 64-byte record per entity


#ifndef SPCG_PRE
#define SPCG_PRE 0   // s_sleep units (64 cycles) between a wavefront's publish and its first poll of a hand-over (measured: 0 is best, profiles/r04_spcg_probe.txt)
#endif

// Kernel ids for the optional per-launch timing hooks (aar_get_kernel_times)
enum KernelId { KID_UNPACK = 0, KID_RESIDUAL, KID_PASSA, KID_PASSB, KID_MAXDIAG, KID_FRAME_INV, KID_SCHUR, KID_LDL_DIAG,
                KID_LDL_TRSM, KID_LDL_UPDATE, KID_LDL_BACKSOLVE, KID_BACKSUB, KID_REDUCE, KID_LDL_PANEL, KID_PCG, KID_SPCG, KID_SPCG_PRE, KID_COUNT };

struct LaunchHook {  // called around every kernel launch when profiling is on
    void (*pre)(void *ctx, int kid) = nullptr;
    void (*post)(void *ctx, int kid) = nullptr;
    void *ctx = nullptr;
};

// Per-observation record of an ordering: {frame (local), camera, marker, frame-local W slots: camera | marker << 10 | intrinsics << 21}
struct ObsIdx {
    int32_t frame, cam, marker, slots;
};

struct DeviceProblem {
    // sizes
    int C = 0, M = 0, A = 0;          // cameras, markers, shared entities A = C + M (+ C intrinsics entities when intr)
    int intr = 0;                     // optimize_cam_intrinsics: entity C + M + c = (fx, cx, fy, cy, -, -) of camera c, its row = K
    int F = 0;                        // local frames
    int64_t N = 0;                    // local observations
    int n = 0, n_pad = 0, nT = 0;     // reduced system: n = 6A, padded to a multiple of CHOL_NB, tiles
    int total_slots = 0, max_kf = 0;  // sum / max of distinct shared entities per frame
    int n_chunks = 0, n_swork = 0;
    int res_f32 = 1;
    float huber = -1.f;               // Huber delta of the residual weights (< 0: plain residuals)
    double half_size = 0;             // (double)((float)marker_size / 2.f)
    // constant problem data
    double *K = nullptr;              // [C][9]
    ObsIdx *a_idx = nullptr;          // ordering A = reference order (frame, camera, detection order)
    float *a_uv = nullptr;            // [N][8]
    int32_t *frame_obs_start = nullptr;   // [F+1]
    int32_t *frame_stride = nullptr;      // [F] pass A (wrench form): lane i of a frame's workgroup starts at observation (i * stride) mod n, stride coprime with n
    int32_t *fslot_start = nullptr;       // [F+1] first W block of each frame
    int32_t *fslot_ent = nullptr;         // [total_slots] shared entity of each W block (ascending inside a frame)
    ObsIdx *b_idx = nullptr;          // ordering B = (camera, marker, frame)
    float *b_uv = nullptr;
    int32_t *chunk_start = nullptr;   // [n_chunks+1] into ordering B; a chunk never spans two (camera, marker) runs
    int32_t *ent_fixed = nullptr;     // [A] 1 = root or non-optimised group
    int frames_fixed = 0;
    // Schur work list: item w handles pairs [sw_begin[w], sw_end[w]) of entity sw_ent[w]
    int32_t *sw_ent = nullptr, *sw_begin = nullptr, *sw_end = nullptr;
    int4 *pair_rec = nullptr;             // (entity, frame) incidence, grouped by entity: {frame, W block, first W block of the frame, 0}
    // Schur work list of the MFMA kernel (many shared entities): item w = block (16 dense entities sm_ga[w]) x (32 dense entities
    // sm_gb[w]) of S over the frames sm_frames[sm_fb[w] .. sm_fe[w])
    int n_smwork = 0;
    int32_t *sm_ga = nullptr, *sm_gb = nullptr, *sm_fb = nullptr, *sm_fe = nullptr, *sm_frames = nullptr;
    // ... which reads DENSE per-frame panels: dense entity 0 = the frame's gradient g_f (a pseudo entity whose block row 0 is g_f: its
    // column of S is the Schur part of the right-hand side), 1.. = the shared entities that are seen at all, MOST FREQUENT FIRST
    // (ties ascending; k_schur_mfma stores a pair transposed where the real entity order is the other way round); Ad = their
    // number padded to a multiple of 32.  Blocks of absent (entity, frame) pairs are zero ONCE and for all (the visibility pattern
    // never changes), k_schur_fill only rewrites the present ones.
    int Ad = 0;
    int32_t *slot_dense = nullptr;        // [total_slots] dense entity of every W block
    int32_t *slot_frame = nullptr;        // [total_slots] frame of every W block
    int32_t *dense_ent = nullptr;         // [Ad] shared entity of a dense index (-1: the pseudo entity / padding)
    double *Wd = nullptr, *Yd = nullptr;  // [F][Ad][36] W_af and W_af (V_f + mu I)^-1
    int dense_from_passA = 1;             // pass A writes them itself whenever it inverts V_f for a predicted damping (AAR_DENSE_FROM_PASSA=0: always k_schur_fill)
    // AAR_DETERMINISTIC=1: every sum that the default path leaves to fp64 atomics (whose order changes from run to run) is taken
    // in a FIXED order instead -- pass B writes per-chunk partials that a second kernel adds up chunk-ascending, the Schur kernel
    // (always the output-stationary one, one wavefront per work item) writes per-item row panels that a second kernel adds up
    // frame-ascending, pass A runs one wavefront per frame.  Two runs of the same problem then give the same bits.
    int deterministic = 0;
    double *pb_part = nullptr;            // [n_chunks][pb_stride] per-chunk sums of pass B (90 values, + 62 with intrinsics)
    int pb_stride = 0, n_pbr = 0;
    // reduction items of pass B: item i adds up the chunks pbr_chunk[pbr_start[i] .. pbr_start[i+1]) (ascending) for camera pbr_a
    // (kind 0), marker entity pbr_a (kind 1) or the pair (camera pbr_a, marker entity pbr_b) (kind 2)
    int32_t *pbr_start = nullptr, *pbr_chunk = nullptr, *pbr_kind = nullptr, *pbr_a = nullptr, *pbr_b = nullptr;
    double *sp_part = nullptr;            // per Schur work item: its row panel [(a+1)*36] | its 6 rhs entries | 2 idle
    int64_t *sp_off = nullptr;            // [n_swork] first double of the item's record
    int32_t *se_start = nullptr, *se_items = nullptr;   // [A+1], [n_swork]: the work items of every entity, frame-ascending
    // AAR_SOLVER=pcg (opt-in, pcg_kernels.hip): the reduced system solved by preconditioned CG through the frame blocks
    int use_pcg = 0, pcg_grid = 0, pcg_max_it = 200;
    double pcg_eta = 0.1;                 // |r| <= eta |b| stops an inner solve of a LATE LM step (aar_solver_options.pcg_eta)
    // the inexact solvers' forcing SEQUENCE when AUTO chose them and the caller left the forcing term at its default: while the LM is still far from its stopping rule (the last
    // accepted step took more than pcg_eta_switch of the error away) the inner solve stops at pcg_eta_loose, afterwards at pcg_eta -- the steps that decide
    // the stopping rule are solved as tightly as before (profiles/r04_pcg_eta_sweep.txt).  pcg_eta_now is what the next launch uses.
    double pcg_eta_loose = 0.0, pcg_eta_switch = 0.01, pcg_eta_now = 0.1;
    // ... and an ABSOLUTE one beside the relative forcing term (both must hold): the error an inner solve leaves in the step, in the units of the pose vector (radians / metres).
    // The relative term alone lets a solve stop while the step is still large in absolute terms -- far starts, a tiny initial damping (tau = 1e-6: poses 4e-3 off): there the
    // absolute term keeps the CG going (or sends the try to the direct chain through the iteration cap).  Both solvers: r^T M^-1 r <= eps^2 mu (M = the block-Jacobi preconditioner).
    double pcg_abs_tol = SPCG_ABS_TOL_DEFAULT;
    int pcg_n_items = 0;                  // work items of the entity-side passes (pcg_kernels.hip): ranges of one entity's incidences in pair_rec
    int32_t *pcg_it_ent = nullptr, *pcg_it_begin = nullptr, *pcg_it_end = nullptr, *pcg_ent_item_start = nullptr;
    double *pcg_ws = nullptr;             // items' shares [n_items][28] | t [6F]
    int32_t *pcg_counter = nullptr;       // [0..1] grid-barrier counters (alternating), [2] iterations of the last solve, [3] running total
    mutable int pcg_parity = 0;
    int n_cus = 256;                      // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    mutable int want_w64 = 0;             // aar_eval_normal_equations: pass A writes the fp64 W blocks even where the solver only reads the fp32 copy (Blocks::Wf)
    int pcg_fused = 1;                    // AAR_PCG_FUSED=0: k_pcg (two passes over W and two hand-overs per iteration) instead of k_pcgf
    int pcg_rank0 = 1;                    // this rank adds the terms of the sharded PCG's set-up that must enter ONCE (the damping of the coarse operator)
    int pcg_resident = 1;                 // AAR_PCG_RESIDENT=0: k_pcgf streams every W block once per CG iteration (1: the first round of every wavefront's first frame stays in registers for the solve -- fp32 blocks; both rounds with fp64 blocks, whose kernel has 512 registers per lane)
    int pcg_e_every = 1;                  // AAR_PCG_E_EVERY: k_pcgf forms the coarse operator's Z^T S Z every this-many solves of an LM run and keeps it in between (the damping's mu Z^T Z is always today's); 1: every solve; set by size at problem creation (3 from 300 000 (entity, frame) pairs)
    mutable int pcg_e_age = 0;            // solves since E was formed (0: the next one forms it; reset where an LM run starts)
    int pcg_coarse = 1, pcg_coarse_from = 8;   // AAR_PCG_COARSE=0 / AAR_PCG_COARSE_FROM: k_pcgf's coarse space (the groups' rigid-motion modes) joins when the previous solve of the LM run took this many iterations
    int32_t *up_start = nullptr, *up_ent = nullptr;   // [A + 1], [..]: entity -> the OTHER entities whose block of U can be non-zero (seen together in an observation); the CG operator skips the rest
    double *pcg_yg = nullptr;             // k_pcgf: y [3][PCG_NYV][n_pad] (rotating; PCG_NYV partial vectors) | the set-up's sums [A][28]
    int32_t *pcg_hop = nullptr;           // k_pcgf: [2][PCG_HOP_WORDS] barrier tree (pcg_kernels.hip, grid_hop_tree), by launch parity
    // solver spcg (spcg_kernels.hip): CG on the explicit Schur complement, one wavefront per shared entity
    int use_spcg = 0, spcg_max_it = 64;
    int spcg_spread = 8;                  // AAR_SPCG_SPREAD=1: every workgroup of the grid works (the wavefronts then sit on all XCDs and hand over through memory)
    int spcg_coarse = 1;                  // AAR_SPCG_COARSE=0: block-Jacobi only (the A/B reference of the two-level preconditioner)
    int spcg_coarse_from = 12;            // AAR_SPCG_COARSE_FROM: the coarse space joins once a solve of the LM run has taken this many CG iterations (0: from the first solve)
    mutable int spcg_coarse_on = 0;       // the next k_spcg launch carries the coarse space (damped_try decides; sticky within an LM run)
    double *spcg_pre = nullptr;           // workspace of k_spcg_pre (spcg_pre_doubles): augmented rows | right-hand side | inverse blocks | Z | (A Z)^T | shares | counter
    mutable int spcg_pre_epoch = 0;       // launches of k_spcg_pre (its arrival counter is monotonic)
    int spcg_root_c = -1, spcg_root_m = -1, spcg_n_free = 0;   // the fixed entity of each group whose slot carries the group's rigid-motion unknowns (-1: none); free entities
    int spcg_test_drop = -1;              // test hook (AAR_SPCG_TEST_DROP=entity): that entity's wavefront never shows up -> every hand-over times out -> flag 4 -> direct chain
    double *spcg_ws = nullptr;            // [2][SPCG_BUFS][spcg_stride(n_pad)] hand-over slots (sentinel-filled when idle)
    int32_t *spcg_iters = nullptr;        // [0] iterations of the last solve, [1] running total, [2] solves, [3] solves that hit the cap (flag 8)
    mutable int spcg_parity = 0;
    int32_t *spcg_done = nullptr;         // [0] arrivals of CG workgroups, [1] flag: the frame back-substitution riding in k_spcg's launch waits for it
    mutable int spcg_epoch = 0;
    // ... with frames sharded over ranks: this rank's set-up share [A][28] (all-reduced), Minv [A][36], x | r | p | scalars [3 n + 8],
    // this rank's partial y [n] (all-reduced per iteration), mapped host record {done, iterations, -, sequence}
    double *pcgd_setup = nullptr, *pcgd_minv = nullptr, *pcgd_state = nullptr, *pcgd_y = nullptr, *pcgd_host = nullptr;
    // tuning switches (environment, read when THE PROBLEM is created -- aar_problem_create_ex -- so that two problems of one process may differ;
    // defaults are the measured optima)
    struct Tuning {
        int fused_panel = 3;       // AAR_FUSED_PANEL: block columns with at most this many tiles below the diagonal take k_ldl_panel
        int bs_rides = 1;          // AAR_BS_RIDES=0: the back-substitution of 2-3 tile systems as its own chained launch
        int backsub_rides = 0;     // AAR_BACKSUB_RIDES=1: the frame back-substitution rides in the last tile's launch
        int lookahead = 1;         // AAR_LDL_LOOKAHEAD=0: tall block columns launch k_ldl_update
        int passA_variant = 0;     // AAR_PASSA_VARIANT: 641 / 642 / 644 / 1281 / 1282 / 1284 / 2564 force a pass A workgroup shape (threads, corners per lane)
        int passAB_occ2 = -1;       // AAR_PASSAB_OCC2=0 / 1: the merged observation passes capped at 256 registers (two wavefronts per SIMD); -1: when their workgroups exceed one round of the chip's slots
        int spcg_backsub_rides = 0; // AAR_SPCG_BACKSUB_RIDES=1 (experiment, slower: profiles/r04_attempts.txt): the frame back-substitution rides in k_spcg's launch on the XCDs the CG leaves idle
        int passA_wrench = 1;      // AAR_PASSA_WRENCH=0: pass A in its row form (three Jacobian blocks per row, 100 accumulators per lane) -- the A/B reference of the wrench form
        int passB_wrench_merged = 0;   // AAR_PASSB_WRENCH_MERGED=1: pass B in wrench form also inside the merged launch of small problems (there a lane has one observation and the
                                       // stage after the wave sum costs more than the rows save: -1.5 % LM it/s at config 3); as a launch of its own pass B is always in pass A's form
        int passB_lean = 0;        // AAR_PASSB_LEAN (experiment): 1 = pass B's corner loop not unrolled + two wavefronts per SIMD (28 spilled registers), 2 = not unrolled only
        int pack_system = -1;      // AAR_PACK_SYSTEM=0/1: the reduced system travels as it lies / as the packed triangle (default: by size)
        int init_headstart = 1;    // AAR_INIT_HEADSTART=0: the first step's frame inverses and Schur complement wait for the host to have read mu_0
    } tune;
    // state: everything that depends on a pose vector exists twice (index 0/1 = the two pose buffers), so that the
    // blocks of a trial point can be built while those of the current point are still needed for a mu retry
    double *z[2] = {nullptr, nullptr};    // [6A + 6F] pose vectors
    double *ent[2] = {nullptr, nullptr};  // [(A+F)][ENT_STRIDE]
    struct Blocks {
        double *V = nullptr, *gf = nullptr;   // [F][36], [F][6]
        double *W = nullptr;                  // [total_slots][36]   W_af (rows: entity params, cols: frame params)
        float *Wf = nullptr;                  // PCG (k_pcgf, fp32 operator): the same blocks in fp32, per frame piece-major -- float4 piece q (entries 4q .. 4q+3 of the
                                              // row-major block) of the frame's slot j at float4 index (fslot_start[f] * 9 + q k_f + j) -- written by pass A beside W
        double *Vinv = nullptr, *hf = nullptr;// (V_f + mu I)^-1, (V_f + mu I)^-1 g_f
        double *S = nullptr;                  // [n_pad*n_pad] row-major lower triangle: shared blocks (pass B), then minus
                                              // the Schur terms, then (in place) its LDL^T factor
        double *rhs = nullptr;                // [n_pad] Schur part of the right-hand side, then z = D^-1 L^-1 b
        double *g0 = nullptr;                 // [n_pad] shared part of B = -J^T r (kept for the gain denominator)
        double *tail = nullptr;               // [8] right behind g0: the step's scalars of THIS rank, so that on the multi-GPU path
                                              // they travel in the same all-reduce as the next step's S | rhs | g0
    } blk[2];
    double *Dfac = nullptr;               // [nT][NB*NB] factored diagonal tiles (unit L below, D on the diagonal)
    double *Linv16 = nullptr;             // [nT][6][16*16] inverses of the 16x16 diagonal sub-blocks of every L_ss
    double *Lp = nullptr;                 // [nT][n_pad][NB] block columns of L written by the fused panel kernel (stages with <= 2 tiles below the diagonal)
    double *zf = nullptr;                 // [n_pad] forward-substituted right-hand side of those stages
    double *delta_s = nullptr;            // [n_pad]
    int32_t *bs_flags = nullptr;          // [nT] k_ldl_backsolve: flag[s] == bs_epoch <=> x_s of the current launch is in memory
    mutable int bs_epoch = 0;
    double *err_part = nullptr;           // [max(F, residual_blocks)] partial sums of squared residuals
    double *lin_part = nullptr;           // [F+1][2] per-frame ( |delta_f|^2 , delta_f . g_f ), last = shared part
    double *scal = nullptr;               // [8] reduced scalars
    // host-visible (pinned, mapped) result record the reduction kernel publishes: [0..7] scalars, [8] flags, [9] sequence
    double *host_result = nullptr;        // device pointer to the mapped host record (10 x 8 bytes)
    int32_t *flags = nullptr;             // [4] device error flags (0: non-positive pivot)
    double *r_out = nullptr;              // optional [8N]
    LaunchHook hook;
};

// opt in to more than the default dynamic LDS per workgroup (gfx950: 160 KiB per CU); remembered per kernel
inline void allow_dynamic_lds(const void *kernel, size_t bytes, size_t &granted) {
    if (bytes > granted) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        granted = bytes;
    }
}

// camera matrices: the constant table, or -- intrinsics being optimised -- the rows of the intrinsics entities of pose buffer `which`
struct KTable { const double *base; int stride; };
constexpr int ENT_STRIDE_H = 24;   // = geom.hpp's ENT_STRIDE
inline KTable k_table(const DeviceProblem &P, int which) {
    if (P.intr) return KTable{P.ent[which] + (size_t)(P.C + P.M) * ENT_STRIDE_H, ENT_STRIDE_H};
    return KTable{P.K, 9};
}
constexpr int SLOT_C_BITS = 10, SLOT_M_BITS = 11;   // ObsIdx::slots
inline int pack_slots(int sc, int sm, int sk) { return sc | (sm << SLOT_C_BITS) | (sk << (SLOT_C_BITS + SLOT_M_BITS)); }

extern volatile int g_last_kernel_id;   // the kernel id of the most recent launch of this process (diagnostics: AAR_ABORT_BACKTRACE prints it)
struct HookScope {  // RAII: pre/post around one launch
    const DeviceProblem &P;
    int kid;
    HookScope(const DeviceProblem &p, int k) : P(p), kid(k) { g_last_kernel_id = k; if (P.hook.pre) P.hook.pre(P.hook.ctx, kid); }
    ~HookScope() { if (P.hook.post) P.hook.post(P.hook.ctx, kid); }
};

// z -> ent (start of a solve, rebuilds, the residual-vector API); zero_blk >= 0: the same launch clears that block set's S | rhs | g0 and lin_part
void launch_unpack(const DeviceProblem &P, int which, hipStream_t st, int zero_blk = -1);
void launch_residual(const DeviceProblem &P, int which, double *r_out, hipStream_t st);
// pass A at z[which]: entity table, V, g_f, W, per-frame sum r^2 (err_part[f]); mu_pred >= 0 also gives Vinv, h_f for that
// damping; zero_blk >= 0 clears S, rhs, g0 of that block set (they are dead / about to be rebuilt)
void launch_passA(const DeviceProblem &P, int which, double mu_pred, int zero_blk, hipStream_t st);
void launch_passB(const DeviceProblem &P, int which, hipStream_t st);              // accumulates into blk[which].S, .g0 (must be zero); with intrinsics: their shared blocks too
// both passes in one launch (they only share the entity table ent[which], which must be complete); false = nothing launched
bool launch_passAB(const DeviceProblem &P, int which, double mu_pred, int zero_blk, hipStream_t st);
size_t passA_lds_bytes(int max_kf, int block);   // dynamic LDS of the frame-block kernel in row form (block = 64 or 256 threads)
size_t passA_wrench_lds_bytes(int max_kf, bool intr);   // ... in wrench form (the default)
void launch_maxdiag(const DeviceProblem &P, int which, hipStream_t st);            // scal[4] = max free diagonal
void launch_frame_inv(const DeviceProblem &P, int which, double mu, hipStream_t st, const double *mu_dev = nullptr, double mu_scale = 0.0);   // mu_dev: mu = mu_scale * mu_dev[0], read on the device
// S -= sign * W (V+mu I)^-1 W^T (sign -1 takes it back).  ride_seq != 0: the reduction of the step's scalars rides in the same
// launch (true is returned if it did)
// panels_ready (MFMA path): Wd / Yd already hold this block set's panels for the damping in Vinv (pass A wrote them): no k_schur_fill
bool launch_schur(const DeviceProblem &P, int which, double sign, hipStream_t st, unsigned long long ride_seq = 0, int ride_n_err = 0,
                  double *ride_scal = nullptr, bool panels_ready = false);   // ride_scal: the rider leaves the scalars there and does not publish
// damping + LDL^T + both substitutions -> delta_s; trial >= 0: launch_backsub(which, trial) may ride in the last launch (true: it did)
bool launch_chol(const DeviceProblem &P, int which, double mu, hipStream_t st, int trial = -1);
void launch_backsub(const DeviceProblem &P, int cur, int trial, hipStream_t st);   // z[trial] = z[cur] + delta, lin_part
void launch_pcg(const DeviceProblem &P, int which, double mu, hipStream_t st);     // AAR_SOLVER=pcg: delta_s by PCG through the frame blocks (needs Vinv, hf for mu)
size_t pcg_lds_bytes(int A, bool coarse = false);   // coarse: with the tables of k_pcgf's coarse space
int pcg_max_grid(int A, int cus);   // largest co-resident grid of the persistent PCG kernels
// solver spcg: delta_s by CG on the explicit reduced system S of block set `which` (the Schur complement for mu must have been taken; S is not modified)
bool launch_spcg(const DeviceProblem &P, int which, double mu, hipStream_t st, int trial = -1);   // trial >= 0: launch_backsub(which, trial) may ride (true: it did)
int spcg_resident_per_cu(int nT, bool coarse);             // occupancy query: wavefronts of k_spcg<nT> one CU holds (0: unknown)
inline bool spcg_coarse_now(const DeviceProblem &P) { return P.spcg_coarse && (P.spcg_root_c >= 0 || P.spcg_root_m >= 0); }   // this problem's k_spcg can carry the coarse space
bool spcg_fits(int nT);                                    // the system's rows fit the wavefronts' registers
size_t spcg_ws_doubles(int n_pad);
size_t spcg_pre_doubles(int n_pad);
void spcg_ws_reset(const DeviceProblem &P, hipStream_t st);   // every hand-over slot back to the sentinel (at creation, after a timed-out launch)
// the same with a communicator: set-up share -> [all-reduce] -> launches k = 0, 1, .. with an all-reduce of pcgd_y between two of them
void launch_pcgd_setup(const DeviceProblem &P, int which, double mu, hipStream_t st);
void launch_pcgd_iter(const DeviceProblem &P, int which, double mu, int k, bool last, unsigned long long publish_seq, hipStream_t st);
double *pcgd_y_of_launch(const DeviceProblem &P, int k);   // this rank's partial y of launch k: what the host all-reduces before launch k + 1
void launch_reduce_scalars(const DeviceProblem &P, int n_err, bool fold_shared, unsigned long long publish_seq, hipStream_t st,
                           double *scal_out = nullptr, int maxdiag_blk = -1);  // scal[0..2], scal[5..6] (into scal_out instead of P.scal if given);
                                                                               // maxdiag_blk >= 0: also scal[4] = max free diagonal of that block set (k_maxdiag's job)
// scal / flags -> host record; flags_reduced: the flags are decoded from src[3] (every rank's flags, all-reduced) instead of P.flags
void launch_publish(const DeviceProblem &P, unsigned long long publish_seq, hipStream_t st, const double *src = nullptr, bool flags_reduced = false);
int residual_blocks(const DeviceProblem &P);   // entries of err_part written by launch_residual
// track(): every frame's own 6-DoF LM, whole loop on the device; needs ent[which] rows of the shared entities (launch_unpack)
void launch_track(const DeviceProblem &P, int which, int max_iters, double min_error, double min_step, double min_avg, double tau,
                  int32_t *iters_out, double *err_out, hipStream_t st);

// The error flags of a rank (bits 0..3) as one double that survives a SUM all-reduce over up to 4095 ranks: bit b set on k ranks adds k 4096^b.
// Every rank decodes the same value, so every rank takes the same branch (a rank that failed alone would otherwise leave the
// others waiting in their next collective).
__device__ __forceinline__ double encode_flags(int f) {
    return (double)(f & 1) + 4096.0 * (double)((f >> 1) & 1) + 16777216.0 * (double)((f >> 2) & 1) + 68719476736.0 * (double)((f >> 3) & 1);
}
__device__ __forceinline__ int decode_flags(double v) {
    const long long q = (long long)v;
    return ((q % 4096) ? 1 : 0) | (((q / 4096) % 4096) ? 2 : 0) | (((q / 16777216) % 4096) ? 4 : 0) | ((q / 68719476736LL) ? 8 : 0);
}

// flags_reduced: the flags come from scal[3] (all ranks' flags, summed by the all-reduce) instead of this rank's flag words
__device__ __forceinline__ void publish_host(const double *__restrict__ scal, const int32_t *__restrict__ flags,
                                             double *__restrict__ host, unsigned long long seq, int flags_reduced = 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) host[i] = scal[i];
    reinterpret_cast<long long *>(host)[8] = flags_reduced ? (long long)decode_flags(scal[3]) : (long long)(flags[0] | flags[1] | flags[2] | flags[3]);
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host) + 9, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}


}  // namespace aar
