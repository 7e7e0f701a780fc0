// C-ABI compute entry points (include/aar.h): problem upload, index structures, the LM loop of
// ucoslam::SparseLevMarq<double> (libs/sparselevmarq.h:238-249,349-430,440-472) driven from the host
// with every arithmetic stage on the GPU, and the RCCL exchange for frame-sharded problems.
//
// There is deliberately no CPU compute path in this file: without a HIP device every compute entry
// point returns AAR_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <unistd.h>
#include <signal.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "../host/internal.h"
#include "geom.hpp"
#include "hostcopy.h"
#include "kernels.h"

using namespace aar;

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return set_error(AAR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// RCCL, resolved at run time so that libaar.so loads on machines without it (and shares whichever
// librccl.so.1 the process already holds, e.g. the one PyTorch brought in).
// ------------------------------------------------------------------------------------------------
namespace {
struct NcclId { char internal[128]; };
typedef void *NcclComm;
struct NcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommCount)(NcclComm, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
NcclApi g_nccl;
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0, NCCL_MAX = 2;

int load_nccl() {
    if (g_nccl.ok) return AAR_OK;
    // One node, one process per GPU over xGMI: the bootstrap sockets can stay on loopback and there is no InfiniBand to
    // probe.  On hosts whose name does not resolve the default interface discovery has been seen to take minutes.
    // Both are only defaults: a caller's own environment wins.
    setenv("NCCL_SOCKET_IFNAME", "lo", 0);
    setenv("NCCL_IB_DISABLE", "1", 0);
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_nccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_nccl.lib) break;
    }
    if (!g_nccl.lib) return set_error(AAR_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
    g_nccl.GetUniqueId = (int (*)(NcclId *))dlsym(g_nccl.lib, "ncclGetUniqueId");
    g_nccl.CommInitRank = (int (*)(NcclComm *, int, NcclId, int))dlsym(g_nccl.lib, "ncclCommInitRank");
    g_nccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, NcclComm, hipStream_t))dlsym(g_nccl.lib, "ncclAllReduce");
    g_nccl.CommDestroy = (int (*)(NcclComm))dlsym(g_nccl.lib, "ncclCommDestroy");
    g_nccl.CommCount = (int (*)(NcclComm, int *))dlsym(g_nccl.lib, "ncclCommCount");
    g_nccl.GetErrorString = (const char *(*)(int))dlsym(g_nccl.lib, "ncclGetErrorString");
    if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.AllReduce || !g_nccl.CommDestroy)
        return set_error(AAR_ERR_COMM, "librccl lacks the expected nccl* symbols");
    g_nccl.ok = true;
    return AAR_OK;
}
}  // namespace

// In-process stand-in for a communicator (bring-up / tests): `world` host threads of ONE process, each with its own
// aar_problem on the SAME GPU, exchange through this group instead of RCCL.  Every rule of the sharded path (frame ranges,
// the all-reduces of S | rhs, of the step's scalars, of the initial diagonal, the final gather) runs unchanged; only the
// transport differs.  Lets the multi-rank logic be checked on a 1-GPU box.
struct aar_local_group {
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long gen = 0;
    std::vector<const double *> ptrs;
    bool broken = false;
    // false: a rank never arrived (it failed before its collective) -- the group is marked broken and every waiter returns,
    // as a communicator whose peer died would; a test must fail, not hang
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned long long g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || broken; }) || broken) {
            broken = true;
            cv.notify_all();
            return false;
        }
        return true;
    }
};

struct aar_comm {
    NcclComm comm = nullptr;
    int world = 1, rank = 0, device = 0;
    aar_local_group *local = nullptr;   // non-null: in-process group instead of RCCL
    double *tmp = nullptr;              // local transport: reduction scratch on the device
    size_t tmp_count = 0;
    int64_t allreduce_calls = 0, allreduce_bytes = 0, last_system_bytes = 0;   // aar_comm_get_stats
};

namespace {
__global__ void k_local_reduce(double *__restrict__ out, const double *const *__restrict__ in, int world, size_t count, int is_max) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double v = in[0][i];
    for (int r = 1; r < world; r++) v = is_max ? fmax(v, in[r][i]) : v + in[r][i];   // fixed rank order: every rank gets the same bits
    out[i] = v;
}

// The reduced system travels as the PACKED lower triangle (only that half is ever produced or read): row i of S contributes its
// i + 1 leading entries, then the `extra` doubles that sit right behind S (rhs | g0 | the step's scalars).  grid.y = row
// (row n_pad = the extra part), dir 0 = pack, 1 = unpack.
__global__ void __launch_bounds__(256) k_pack_system(double *__restrict__ S, int n_pad, int extra, double *__restrict__ packed, int dir,
                                                     double *__restrict__ host, unsigned long long publish_seq, const int32_t *__restrict__ flags) {
    const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    const size_t tri = (size_t)n_pad * (n_pad + 1) / 2;
    if (dir == 1 && publish_seq && i == n_pad && j == 0) {
        // the step's scalars go to the host from HERE, beside the unpacking (the host only reads the record; the next kernel it
        // queues is stream-ordered behind this one): tail[0..3] reduced over the ranks, tail[4..7] this rank's own
        const double *red = packed + tri + 2 * (size_t)n_pad, *own = S + (size_t)n_pad * n_pad + 2 * (size_t)n_pad;
        double sc[8];
#pragma unroll
        for (int q = 0; q < 8; q++) sc[q] = (q < extra - 2 * n_pad) ? red[q] : own[q];
        publish_host(sc, flags, host, publish_seq, /*flags_reduced=*/1);
    }
    if (i < n_pad) {
        if (j > i) return;
        double *a = S + (size_t)i * n_pad + j, *b = packed + (size_t)i * (i + 1) / 2 + j;
        if (dir == 0) *b = *a; else *a = *b;
    } else if (j < extra) {
        double *a = S + (size_t)n_pad * n_pad + j, *b = packed + tri + j;
        if (dir == 0) *b = *a; else *a = *b;
    }
}
}  // namespace

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        int _r = (expr);                                                                                        \
        if (_r != 0)                                                                                            \
            return set_error(AAR_ERR_COMM, "%s failed: %s", #expr, g_nccl.GetErrorString ? g_nccl.GetErrorString(_r) : "?"); \
    } while (0)

// ------------------------------------------------------------------------------------------------
struct aar_problem {
    DeviceProblem P;
    PoseLayout L;             // global layout (all frames)
    int f_begin = 0, f_end = 0;  // global frame range owned by this rank
    int64_t o_begin = 0, N_global = 0;
    aar_comm *comm = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    double *h_scal = nullptr;   // pinned, mapped [10]: 8 scalars | flags | sequence number (written by the device)
    int32_t h_flags[4] = {0, 0, 0, 0};
    unsigned long long seq = 0;
    double *d_frames_all = nullptr;  // [6 F_global] gather buffer (multi-GPU)
    double *d_diag = nullptr;        // [n_pad]
    double *d_pack = nullptr;        // multi-GPU: packed lower triangle of S | rhs | g0 | scalars, the all-reduce payload
    double *d_status = nullptr;      // multi-GPU: one double for collective status decisions
    PinnedBuf h_z;              // pose staging [6A + 6F_loc]: page-locked memory of the problem's own (hostcopy.h), so uploads need no sync
    double *h_gather = nullptr; size_t h_gather_n = 0;   // multi-rank: pinned landing zone of the gathered frame poses (download_z)
    hipEvent_t up_ev = nullptr;  // the last upload from h_z (it must have left before h_z is packed again)
    bool mu_seed_valid = false;  // max diag(J^T J) of the start point, published together with its sum r^2 (aar_lm_init)
    double mu_seed = 0;
    // host copies of the index structure (normal-equation assembly for tests)
    std::vector<int32_t> h_fslot_start, h_fslot_ent;
    // LM state (SparseLevMarq members, libs/sparselevmarq.h:129-136)
    aar_lm_params prm;
    int cur = 0;                       // pose buffer / block set of curr_z
    double mu = -1, v = 2, currErr = 0, prevErr = 0;
    double rel_drop = -1;              // share of the error the last accepted LM step took away (-1: none yet); the PCG forcing sequence looks at it
    bool lm_ready = false;
    bool with_huber = false;
    float hubber_delta = 2.5f;         // MultiCamMapper::hubberDelta (libs/multicam_mapper.h:41)
    // SparseLevMarq::_step_callback / _stopFunction (libs/sparselevmarq.h:135-136)
    aar_lm_step_callback step_cb = nullptr;
    void *step_ctx = nullptr;
    bool step_want_z = false;
    aar_lm_stop_function stop_fn = nullptr;
    void *stop_ctx = nullptr;
    std::vector<double> cb_x, cb_z;    // staging of curr_z for the callbacks
    float huber_of_blocks = -1.f;      // Huber delta the residual behind blk[cur]'s B = -J^T r was weighted with (x64 of libs/sparselevmarq.h:367)
    bool blocks_valid = false;         // blk[cur] holds J^T J blocks and B at z[cur], S not yet eliminated
    double vinv_mu = -1;               // damping for which blk[cur].Vinv / hf are valid (< 0: none)
    double schur_mu = -1;              // damping whose Schur terms are already subtracted from blk[cur].S / rhs (< 0: none)
    int panels_blk = -1;               // MFMA Schur path: the block set whose dense panels Wd / Yd currently hold (-1: none) ...
    double panels_mu = -1;             // ... and the damping of the inverses behind Yd
    bool s_reduced = false;            // multi-GPU: blk[cur].S | rhs | g0 already hold the all-reduced system for schur_mu
    bool trial_reduced = false;        // ... the same for the trial's block set, until the step is accepted or rejected
    bool fused_comm = true;            // the step's scalars and the next step's system share ONE all-reduce (AAR_FUSED_COMM=0: two)
    // multi-GPU: the factorisation of the NEXT step's (already reduced) system is queued right behind the fused all-reduce, before the
    // host has seen this step's scalars -- the usual outcome (accepted, predicted damping) then finds it done, and the host's
    // turn-around hides behind it as it hides behind the Schur kernel on one GPU; any other outcome rebuilds the system anyway
    bool spec_chol = true;             // AAR_SPEC_CHOL=0: off
    int solver = AAR_SOLVER_DIRECT;    // what the problem runs with (aar_solver_options.solver, AUTO resolved)
    int env_overrides = 0;             // AAR_ENV_* bits: option fields an environment variable changed (aar_solver_stats.env_overrides)
    bool force_direct = false;         // solver spcg: THIS try takes the direct chain (the CG solve of the same system hit its cap / timed out)
    // ... and so do the next tries of this solve (the damping only falls along accepted steps: the systems get harder, not easier): 8 after the first
    // fall-back, twice as many after each further one (a 505-step -with-huber run moves its damping both ways: CG gets another chance now and then);
    // aar_lm_init clears both
    int spcg_skip = 0, spcg_backoff = 0;
    // CG iterations of the last SPCG solve of this LM run, as published with the step's scalars (-1: none yet).  Along an LM run the damping only falls and the systems only get
    // harder: a solve that came within 20 % of the iteration cap is taken as the announcement that the next one will not fit -- that try goes to the direct chain at once,
    // instead of through a failed CG attempt (64 .. 128 wasted iterations, the trial evaluation on a garbage step, a rebuild of the blocks)
    int last_cg_its = -1;
    int near_cap_tries = 0;            // tries the announcement above has sent to the direct chain: after 8 of them it is forgotten and the CG gets its chance again
    int64_t spcg_fallbacks = 0;
    double *h_pcg = nullptr;           // solver pcg with a communicator: pinned, mapped {done, iterations, -, sequence} the iteration launches publish
    unsigned long long pcg_seq = 0;
    int spec_chol_blk = -1;            // block set whose S a speculative factorisation has consumed (-1: none pending)
    double spec_chol_mu = -1;
    bool spec_chol_by_cg = false;      // ... whether it was a CG solve (a speculative solve is only consumed by a try that takes the same solver)
    double spec_chol_eta = 0;          // ... and the forcing term it was solved to (inexact solvers: a speculative solve is only consumed by a try that wants the same)
    hipStream_t stream2 = nullptr;     // pass B of the trial point runs here, beside the speculative Schur complement
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap = false;              // AAR_OVERLAP=1: pass B of the trial point on a second stream beside the Schur complement
    bool merge_passes = true;          // passes A and B of an evaluation in one launch (AAR_MERGE_PASSES=0: two launches)
    int64_t trial_points = 0, launches = 0;
    aar_stage_times times;
    bool stage_timers = false;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // per-kernel profiling (aar_set_kernel_profiling)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_kid;       // kernel id of pending pair i (events 2i, 2i+1)
    size_t ev_used = 0;
    double k_seconds[KID_COUNT] = {0};
    int64_t k_launches[KID_COUNT] = {0};
};

namespace {

template <class T>
int dev_alloc(aar_problem *pb, T **ptr, size_t count) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    HIP_TRY(hipMemsetAsync(p, 0, bytes, pb->stream));
    pb->allocs.push_back(p);
    *ptr = static_cast<T *>(p);
    return AAR_OK;
}

// host <-> device through the library's page-locked staging (hostcopy.h): the runtime never sees a pageable pointer
int copy_h2d(aar_problem *pb, void *dst, const void *src, size_t bytes) {
    const char *what = "";
    const int e = h2d(dst, src, bytes, pb->stream, &what);
    return e ? set_error(AAR_ERR_HIP, "%s failed: %s", what, hipGetErrorString((hipError_t)e)) : AAR_OK;
}
int copy_d2h(aar_problem *pb, void *dst, const void *src, size_t bytes) {
    const char *what = "";
    const int e = d2h(dst, src, bytes, pb->stream, &what);
    return e ? set_error(AAR_ERR_HIP, "%s failed: %s", what, hipGetErrorString((hipError_t)e)) : AAR_OK;
}

template <class T>
int dev_upload(aar_problem *pb, T **ptr, const std::vector<T> &h) {
    int rc = dev_alloc(pb, ptr, h.size());
    if (rc) return rc;
    return h.empty() ? AAR_OK : copy_h2d(pb, *ptr, h.data(), h.size() * sizeof(T));
}

int ensure_device(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(AAR_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return set_error(AAR_ERR_INVALID, "device_id %d out of range (%d devices)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    return AAR_OK;
}

// x_full (reference packing, roots skipped) -> device pose vector [A entities | local frames]
void pack_z(const aar_problem *pb, const double *x_full, double *z) {
    const PoseLayout &L = pb->L;
    const int A = pb->P.A, F = pb->P.F;
    memset(z, 0, (size_t)6 * (A + F) * sizeof(double));
    for (int c = 0; c < L.C; c++)
        if (c != L.rc) memcpy(&z[6 * (size_t)c], x_full + L.full_cam0() + 6LL * L.cam_slot(c), 6 * sizeof(double));
    for (int m = 0; m < L.M; m++)
        if (m != L.rm) memcpy(&z[6 * (size_t)(L.C + m)], x_full + L.full_mk0() + 6LL * L.mk_slot(m), 6 * sizeof(double));
    if (F) memcpy(&z[6 * (size_t)A], x_full + L.full_fr0() + 6LL * pb->f_begin, (size_t)6 * F * sizeof(double));
    if (L.oi)   // intrinsics entity of camera c: (fx, cx, fy, cy, -, -); the distortion entries d0..d4 never reach the projection
        for (int c = 0; c < L.C; c++) memcpy(&z[6 * (size_t)(L.C + L.M + c)], x_full + L.full_intr0() + 9LL * c, 4 * sizeof(double));
}

int upload_z(aar_problem *pb, const double *x_full, int which) {
    if (pb->up_ev) HIP_TRY(hipEventSynchronize(pb->up_ev));   // the previous upload has left the staging buffer (it normally has long ago)
    const size_t cnt = (size_t)6 * (pb->P.A + pb->P.F);
    if (pb->h_z.reserve(cnt)) return set_error(AAR_ERR_HIP, "hipHostMalloc(pose staging) failed");
    pack_z(pb, x_full, pb->h_z.data());
    if (!pb->up_ev && hipEventCreateWithFlags(&pb->up_ev, hipEventDisableTiming) != hipSuccess) { pb->up_ev = nullptr; (void)hipGetLastError(); }
    if (cnt) HIP_TRY(hipMemcpyAsync(pb->P.z[which], pb->h_z.data(), cnt * sizeof(double), hipMemcpyHostToDevice, pb->stream));   // (page-locked source)
    if (pb->up_ev) HIP_TRY(hipEventRecord(pb->up_ev, pb->stream));
    else HIP_TRY(hipStreamSynchronize(pb->stream));
    return AAR_OK;
}

int allreduce(aar_problem *pb, double *buf, size_t count, int op) {
    if (!pb->comm) return AAR_OK;
    pb->comm->allreduce_calls++;
    pb->comm->allreduce_bytes += (int64_t)(count * sizeof(double));
    if (aar_local_group *g = pb->comm->local) {   // in-process transport (see aar_local_group)
        aar_comm *c = pb->comm;
        if (c->tmp_count < count + (size_t)g->world) {
            if (c->tmp) (void)hipFree(c->tmp);
            c->tmp_count = count + (size_t)g->world;
            HIP_TRY(hipMalloc((void **)&c->tmp, c->tmp_count * sizeof(double)));
        }
        HIP_TRY(hipStreamSynchronize(pb->stream));            // my contribution is complete
        g->ptrs[c->rank] = buf;
        if (!g->barrier()) return set_error(AAR_ERR_COMM, "local group: a rank did not reach the collective");   // ... and so is everybody else's
        const double **d_ptrs = reinterpret_cast<const double **>(c->tmp + count);
        { int rc = copy_h2d(pb, (void *)d_ptrs, g->ptrs.data(), sizeof(double *) * g->world); if (rc) return rc; }
        hipLaunchKernelGGL(k_local_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, pb->stream, c->tmp, d_ptrs, g->world, count,
                           op == NCCL_MAX ? 1 : 0);
        HIP_TRY(hipStreamSynchronize(pb->stream));
        if (!g->barrier()) return set_error(AAR_ERR_COMM, "local group: a rank did not reach the collective");   // everyone has read everyone's buffer: it may be overwritten now
        HIP_TRY(hipMemcpyAsync(buf, c->tmp, count * sizeof(double), hipMemcpyDeviceToDevice, pb->stream));
        return AAR_OK;
    }
    NCCL_TRY(g_nccl.AllReduce(buf, buf, count, NCCL_FLOAT64, op, pb->comm->comm, pb->stream));
    return AAR_OK;
}

// All-reduce of the reduced system of block set `which`: packed lower triangle of S | rhs | g0 (| the first n_tail doubles of
// the tail: the step's scalars of the fused collective).  n_pad (n_pad + 1) / 2 + 2 n_pad + n_tail doubles instead of the
// n_pad^2 + ... of the square.
int allreduce_system(aar_problem *pb, int which, int n_tail, unsigned long long publish_seq = 0) {
    DeviceProblem &P = pb->P;
    // Small systems travel as they lie -- the square S (its upper triangle is never written: zeros) | rhs | g0 | tail are ONE allocation --
    // without the two packing launches: below ~1 MB an all-reduce over xGMI is latency, not bytes, and each launch on this dependent
    // chain costs 3-4 us.  From 384 unknowns on (2.4 MB square) the packed triangle's halved payload wins.  AAR_PACK_SYSTEM=0 / 1 forces.
    const bool pack = P.tune.pack_system >= 0 ? P.tune.pack_system != 0 : P.n_pad > 384;
    if (!pack) {
        const size_t count = (size_t)P.n_pad * P.n_pad + 2 * (size_t)P.n_pad + (size_t)n_tail;
        int rc = allreduce(pb, P.blk[which].S, count, NCCL_SUM);
        if (rc) return rc;
        pb->comm->last_system_bytes = (int64_t)(count * sizeof(double));
        // tail[0..3] are now the rank sums, tail[4..7] still this rank's own: exactly the record the host reads
        if (publish_seq) { launch_publish(P, publish_seq, pb->stream, P.blk[which].tail, /*flags_reduced=*/true); pb->launches += 1; }
        return AAR_OK;
    }
    const int extra = 2 * P.n_pad + n_tail;
    const size_t count = (size_t)P.n_pad * (P.n_pad + 1) / 2 + (size_t)extra;
    const dim3 grid((unsigned)((std::max(P.n_pad, extra) + 255) / 256), (unsigned)P.n_pad + 1);
    hipLaunchKernelGGL(k_pack_system, grid, dim3(256), 0, pb->stream, P.blk[which].S, P.n_pad, extra, pb->d_pack, 0, nullptr, 0ull, nullptr);
    int rc = allreduce(pb, pb->d_pack, count, NCCL_SUM);
    if (rc) return rc;
    pb->comm->last_system_bytes = (int64_t)(count * sizeof(double));
    hipLaunchKernelGGL(k_pack_system, grid, dim3(256), 0, pb->stream, P.blk[which].S, P.n_pad, extra, pb->d_pack, 1, P.host_result, publish_seq, P.flags);
    pb->launches += 2;
    return AAR_OK;
}

// The same status on every rank: the most severe (most negative) code any rank holds.  A rank that fails alone would leave
// the others waiting in their next collective forever (RCCL has no timeout), so rank-local failures that decide control flow
// are agreed on first.
int collective_status(aar_problem *pb, int local_rc, int *agreed) {
    *agreed = local_rc;
    if (!pb->comm) return AAR_OK;
    const double v = (double)(-local_rc);
    int rc = copy_h2d(pb, pb->d_status, &v, sizeof v);
    if (rc) return rc;
    if ((rc = allreduce(pb, pb->d_status, 1, NCCL_MAX))) return rc;
    double w = 0;
    if ((rc = copy_d2h(pb, &w, pb->d_status, sizeof w))) return rc;
    *agreed = -(int)w;
    return AAR_OK;
}

// device pose vector -> x_full; fixed groups keep the caller's values
// side_stream: the copy goes through the problem's second stream and only THAT is waited for.  z[which] must be complete already (the host has
// seen the scalars of the step that wrote it); what is still running on the main stream -- the speculative Schur complement of a step that
// turned out to be the last, ~20 us -- then overlaps with the caller's own work instead of being waited for.  Single GPU only.
int download_z(aar_problem *pb, int which, double *x_full, bool side_stream = false) {
    const PoseLayout &L = pb->L;
    const int A = pb->P.A, F = pb->P.F;
    if (pb->up_ev) HIP_TRY(hipEventSynchronize(pb->up_ev));   // (the staging buffer is shared with the uploads)
    const size_t zcnt = (size_t)6 * (A + F);
    if (pb->h_z.reserve(zcnt)) return set_error(AAR_ERR_HIP, "hipHostMalloc(pose staging) failed");
    const double *z = pb->h_z.data();
    hipStream_t st = (side_stream && !pb->comm && pb->stream2) ? pb->stream2 : pb->stream;
    if (zcnt) HIP_TRY(hipMemcpyAsync(pb->h_z.data(), pb->P.z[which], zcnt * sizeof(double), hipMemcpyDeviceToHost, st));   // (page-locked destination)
    HIP_TRY(hipStreamSynchronize(st));
    if (L.oc)
        for (int c = 0; c < L.C; c++)
            if (c != L.rc) memcpy(x_full + L.full_cam0() + 6LL * L.cam_slot(c), &z[6 * (size_t)c], 6 * sizeof(double));
    if (L.om)
        for (int m = 0; m < L.M; m++)
            if (m != L.rm) memcpy(x_full + L.full_mk0() + 6LL * L.mk_slot(m), &z[6 * (size_t)(L.C + m)], 6 * sizeof(double));
    if (L.oi)
        for (int c = 0; c < L.C; c++) memcpy(x_full + L.full_intr0() + 9LL * c, &z[6 * (size_t)(L.C + L.M + c)], 4 * sizeof(double));
    if (L.of) {
        if (!pb->comm) {
            if (F) memcpy(x_full + L.full_fr0(), &z[6 * (size_t)A], (size_t)6 * F * sizeof(double));
        } else {
            // every rank contributes its own frames to a zeroed vector; the sum is the gather
            const size_t cnt = (size_t)6 * L.F;
            HIP_TRY(hipMemsetAsync(pb->d_frames_all, 0, cnt * sizeof(double), pb->stream));
            if (F)
                HIP_TRY(hipMemcpyAsync(pb->d_frames_all + 6 * (size_t)pb->f_begin, pb->P.z[which] + 6 * (size_t)A,
                                       (size_t)6 * F * sizeof(double), hipMemcpyDeviceToDevice, pb->stream));
            int rc = allreduce(pb, pb->d_frames_all, cnt, NCCL_SUM);
            if (rc) return rc;
            if (pb->h_gather_n < cnt) {   // (the caller's vector is pageable: the copy lands in pinned memory of the problem's own, allocated once)
                if (pb->h_gather) (void)hipHostFree(pb->h_gather);
                pb->h_gather = nullptr; pb->h_gather_n = 0;
                HIP_TRY(hipHostMalloc((void **)&pb->h_gather, cnt * sizeof(double), hipHostMallocDefault));
                pb->h_gather_n = cnt;
            }
            HIP_TRY(hipMemcpyAsync(pb->h_gather, pb->d_frames_all, cnt * sizeof(double), hipMemcpyDeviceToHost, pb->stream));
            HIP_TRY(hipStreamSynchronize(pb->stream));
            memcpy(x_full + L.full_fr0(), pb->h_gather, cnt * sizeof(double));
        }
    }
    return AAR_OK;
}

void prof_pre(void *ctx, int kid) {
    aar_problem *pb = static_cast<aar_problem *>(ctx);
    if (pb->ev_used + 2 > pb->ev_pool.size()) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        pb->ev_pool.push_back(a);
        pb->ev_pool.push_back(b);
    }
    pb->ev_kid.push_back(kid);
    (void)hipEventRecord(pb->ev_pool[pb->ev_used], pb->stream);
}
void prof_post(void *ctx, int) {
    aar_problem *pb = static_cast<aar_problem *>(ctx);
    (void)hipEventRecord(pb->ev_pool[pb->ev_used + 1], pb->stream);
    pb->ev_used += 2;
}
// after a stream synchronisation: fold the pending event pairs into the per-kernel totals
void prof_harvest(aar_problem *pb) {
    if (!pb->profiling) return;
    for (size_t i = 0; i < pb->ev_kid.size(); i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pb->ev_pool[2 * i], pb->ev_pool[2 * i + 1]) == hipSuccess) {
            pb->k_seconds[pb->ev_kid[i]] += ms * 1e-3;
            pb->k_launches[pb->ev_kid[i]]++;
        }
    }
    pb->ev_kid.clear();
    pb->ev_used = 0;
}

struct StageTimer {  // optional per-stage device timing (AAR_STAGE_TIMERS=1); costs two event records + a sync per stage
    aar_problem *pb;
    double *slot;
    StageTimer(aar_problem *p, double *s) : pb(p), slot(s) {
        if (pb->stage_timers) (void)hipEventRecord(pb->ev[0], pb->stream);
    }
    ~StageTimer() {
        if (pb->stage_timers) {
            (void)hipEventRecord(pb->ev[1], pb->stream);
            (void)hipEventSynchronize(pb->ev[1]);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, pb->ev[0], pb->ev[1]);
            *slot += ms * 1e-3;
        }
    }
};

int check_async(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(AAR_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return AAR_OK;
}

// both block sets' shared system and the linear-model partials in ONE launch (aar_lm_init: seven hipMemsetAsync = seven fill
// kernels otherwise, ~25 us of a solve's fixed cost)
struct ZeroArgs { double *p[7]; long long n[7]; };
__global__ void __launch_bounds__(256) k_zero_many(const ZeroArgs z) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
#pragma unroll
    for (int b = 0; b < 7; b++)
        for (long long i = gid; i < z.n[b]; i += stride) z.p[b][i] = 0.0;
}

int zero_for_init(aar_problem *pb) {
    DeviceProblem &P = pb->P;
    ZeroArgs z;
    const long long n2 = (long long)P.n_pad * P.n_pad;
    z.p[0] = P.blk[0].S; z.n[0] = n2; z.p[1] = P.blk[1].S; z.n[1] = n2;
    z.p[2] = P.blk[0].rhs; z.n[2] = P.n_pad; z.p[3] = P.blk[1].rhs; z.n[3] = P.n_pad;
    z.p[4] = P.blk[0].g0; z.n[4] = P.n_pad; z.p[5] = P.blk[1].g0; z.n[5] = P.n_pad;
    z.p[6] = P.lin_part; z.n[6] = 2LL * (P.F + 1);
    const long long blocks = std::min<long long>(2048, (2 * n2 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_zero_many, dim3((unsigned)blocks), dim3(256), 0, pb->stream, z);
    pb->launches += 1;
    return check_async("k_zero_many");
}

int zero_block_set(aar_problem *pb, int which) {
    DeviceProblem &P = pb->P;
    HIP_TRY(hipMemsetAsync(P.blk[which].S, 0, (size_t)P.n_pad * P.n_pad * sizeof(double), pb->stream));
    HIP_TRY(hipMemsetAsync(P.blk[which].rhs, 0, (size_t)P.n_pad * sizeof(double), pb->stream));
    HIP_TRY(hipMemsetAsync(P.blk[which].g0, 0, (size_t)P.n_pad * sizeof(double), pb->stream));
    return AAR_OK;
}

// MFMA Schur path: do Wd / Yd already hold block set `which` for damping mu (pass A wrote them, or an earlier k_schur_fill)?
bool panels_ok(const aar_problem *pb, int which, double mu) { return pb->P.n_smwork > 0 && pb->panels_blk == which && pb->panels_mu == mu; }
// a Schur launch for (which, mu) leaves the panels behind for exactly that pair
void panels_now(aar_problem *pb, int which, double mu) { if (pb->P.n_smwork > 0) { pb->panels_blk = which; pb->panels_mu = mu; } }

// J^T J blocks and B at z[which] into blk[which] (whose S, rhs, g0 must be zero): the "J", "transpose", "Jt*J", "B"
// stages of libs/sparselevmarq.h:353-367.  Pass A also leaves the per-frame sums of r^2 in err_part and, for
// mu_pred >= 0, (V_f + mu_pred I)^-1; zero_blk >= 0 clears that block set on the way.
// spec_schur: also subtract the Schur terms for mu_pred right away (they only need pass A's output), on the main stream,
// while pass B accumulates the shared blocks into the same S on a second stream (both only add into S: they commute).
int eval_blocks(aar_problem *pb, int which, double mu_pred, int zero_blk, bool spec_schur = false, bool ents_ready = false) {
    DeviceProblem &P = pb->P;
    if (!ents_ready) launch_unpack(P, which, pb->stream);   // {R, t, J_l} rows of z[which]; after a damped try k_backsub has written them
    if (P.F == 0 && zero_blk >= 0) {  // a rank without frames launches no pass A: clear the dead block set here
        int rc = zero_block_set(pb, zero_blk);
        if (rc) return rc;
    }
    // pass A rewrites W of this block set: its old panels are stale; it writes new ones itself when it inverts V_f (mu_pred >= 0)
    if (pb->panels_blk == which) pb->panels_blk = -1;
    if (P.n_smwork > 0 && P.dense_from_passA && mu_pred >= 0.0 && P.F > 0) panels_now(pb, which, mu_pred);
    const bool ready = panels_ok(pb, which, mu_pred);
    {
        StageTimer t(pb, &pb->times.jacobian_normal_eq);
        if (pb->merge_passes && launch_passAB(P, which, mu_pred, zero_blk, pb->stream)) {
            if (spec_schur) launch_schur(P, which, 1.0, pb->stream, 0, 0, nullptr, ready);
            pb->launches += spec_schur ? 2 : 1;
            return check_async("normal-equation kernels");
        }
        launch_passA(P, which, mu_pred, zero_blk, pb->stream);
        if (spec_schur && pb->overlap && !pb->profiling && !pb->stage_timers) {
            HIP_TRY(hipEventRecord(pb->ev_fork, pb->stream));
            HIP_TRY(hipStreamWaitEvent(pb->stream2, pb->ev_fork, 0));
            launch_passB(P, which, pb->stream2);
            HIP_TRY(hipEventRecord(pb->ev_join, pb->stream2));
            launch_schur(P, which, 1.0, pb->stream, 0, 0, nullptr, ready);
            HIP_TRY(hipStreamWaitEvent(pb->stream, pb->ev_join, 0));
        } else {
            launch_passB(P, which, pb->stream);
            if (spec_schur) launch_schur(P, which, 1.0, pb->stream, 0, 0, nullptr, ready);
        }
    }
    if (spec_schur) panels_now(pb, which, mu_pred);
    pb->launches += spec_schur ? 3 : 2;
    return check_async("normal-equation kernels");
}

// Wait for the device to publish record `seq` into the mapped host buffer.  Spinning on the sequence word avoids the
// two blit kernels and the interrupt latency of memcpy + hipStreamSynchronize on the per-try critical path.
int wait_result(aar_problem *pb) {
    volatile unsigned long long *sq = reinterpret_cast<volatile unsigned long long *>(pb->h_scal) + 9;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*sq != pb->seq) {
        if ((++spins & 0x3fff) == 0) {
            if (hipStreamQuery(pb->stream) == hipSuccess && *sq != pb->seq) {
                // the stream drained but the record is stale: an asynchronous failure
                hipError_t e = hipGetLastError();
                return set_error(AAR_ERR_HIP, "device did not publish its result: %s", hipGetErrorString(e));
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0)
                return set_error(AAR_ERR_HIP, "timed out waiting for the device result");
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    pb->h_flags[0] = (int32_t)reinterpret_cast<volatile long long *>(pb->h_scal)[8];
    if (pb->profiling || pb->stage_timers) {
        HIP_TRY(hipStreamSynchronize(pb->stream));
        prof_harvest(pb);
    }
    return AAR_OK;
}

// queue the reduction of the step's scalars and their publication to the host record (no wait)
int launch_scalars(aar_problem *pb, int n_err, int maxdiag_blk = -1) {
    DeviceProblem &P = pb->P;
    pb->seq++;
    {
        StageTimer t(pb, &pb->times.control);
        // (PCG mode with ranks: g0 is this rank's partial sum -- it was never all-reduced --, so delta_s . g0 joins the rank sum)
        launch_reduce_scalars(P, n_err, P.use_pcg && pb->comm, pb->comm ? 0ull : pb->seq, pb->stream, nullptr, maxdiag_blk);
        pb->launches += 1;
    }
    if (pb->comm) {
        StageTimer t(pb, &pb->times.allreduce);
        int rc = allreduce(pb, P.scal, 4, NCCL_SUM);  // [sum r^2, sum |delta_f|^2, sum delta.g, every rank's error flags]
        if (rc) return rc;
        launch_publish(P, pb->seq, pb->stream, nullptr, /*flags_reduced=*/true);
    }
    return check_async("kernel launch");
}

int read_scalars(aar_problem *pb, int n_err, int maxdiag_blk = -1) {
    int rc = launch_scalars(pb, n_err, maxdiag_blk);
    if (rc) return rc;
    return wait_result(pb);
}

// One damped solve from blk[cur] (valid, un-eliminated) and the evaluation of the trial point:
//   Schur complement -> [all-reduce] -> LDL^T -> back-substitution -> z[1-cur] = z[cur] + delta
//   -> pass A / pass B at the trial point into blk[1-cur] (speculative: they ARE the next step's Jacobian pass when the
//      trial is accepted, and the trial's sum r^2 comes out of pass A) -> scalars to the host.
// blk[cur].S/rhs are consumed; pass A clears them together with blk[cur].g0 on the way.
constexpr int TRY_NOT_POSITIVE_DEFINITE = 1;   // damped_try: not an error code of the C ABI (those are negative)
constexpr int TRY_CG_FAILED = 2;               // solver spcg: the CG solve hit its iteration cap or its hand-over timed out; S is intact, the caller redoes the try with the direct chain

int damped_try(aar_problem *pb, double mu, bool evaluate_trial) {
    DeviceProblem &P = pb->P;
    const int cur = pb->cur, tr = 1 - cur;
    bool chol_done = false;
    const bool cg_near_cap = pb->last_cg_its >= 0 && 10 * pb->last_cg_its >= 8 * std::min(P.spcg_max_it, SPCG_MAX_IT);
    const bool by_cg = P.use_spcg && !pb->force_direct && pb->spcg_skip == 0 && !cg_near_cap;
    // the coarse space of the CG (k_spcg_pre + the roots' wavefronts) costs ~8 us per solve and saves iterations only once block-Jacobi needs many: it joins when a
    // solve of this run has taken spcg_coarse_from iterations and stays (the damping only falls from there on)
    if (!P.spcg_coarse_on && spcg_coarse_now(P) && (P.spcg_coarse_from <= 0 || pb->last_cg_its >= P.spcg_coarse_from)) P.spcg_coarse_on = 1;
    if (pb->spec_chol_blk >= 0) {   // a factorisation was queued ahead of the host's decision (multi-GPU, see aar_problem::spec_chol)
        if (pb->spec_chol_blk == cur && pb->spec_chol_mu == mu && pb->schur_mu == mu && pb->s_reduced && pb->spec_chol_by_cg == by_cg &&
            (!by_cg || pb->spec_chol_eta == P.pcg_eta_now)) {
            chol_done = true;       // the step was accepted with the predicted damping: this try's factorisation is already running
        } else {
            // rejected (the trial's block set is rebuilt from scratch) or accepted with another damping (s_reduced: the system is
            // rebuilt from the blocks below): what the speculative factorisation left in S, and in the pivot flags, means nothing
            HIP_TRY(hipMemsetAsync(P.flags, 0, 4 * sizeof(int32_t), pb->stream));
        }
        pb->spec_chol_blk = -1;
    }
    if (P.use_pcg) {   // opt-in: no Schur complement, no factorisation -- PCG through the frame blocks (pcg_kernels.hip)
        if (pb->vinv_mu != mu) {
            launch_frame_inv(P, cur, mu, pb->stream);
            pb->vinv_mu = mu;
            pb->launches += 1;
        }
        if (!pb->comm) {
            StageTimer t(pb, &pb->times.chol);
            launch_pcg(P, cur, mu, pb->stream);
            pb->launches += 1;
        } else {
            // frames sharded over ranks: the set-up shares and every iteration's partial y are all-reduced between launches; every
            // fourth launch publishes {done, iterations} so that the host stops queueing (all ranks read the same: same control flow)
            { StageTimer t(pb, &pb->times.chol); launch_pcgd_setup(P, cur, mu, pb->stream); }
            { StageTimer t(pb, &pb->times.allreduce); int rc = allreduce(pb, P.pcgd_setup, (size_t)P.A * 28 + ((P.pcg_fused && !P.deterministic && P.pcg_coarse) ? 144 : 0), NCCL_SUM); if (rc) return rc; }   // (+ the coarse operator's shares)
            pb->launches += 1;
            for (int k = 0;; k++) {
                const bool last = k >= P.pcg_max_it + 1;
                const bool poll = last || (k % 4) == 3;
                if (poll) pb->pcg_seq++;
                { StageTimer t(pb, &pb->times.chol); launch_pcgd_iter(P, cur, mu, k, last, poll ? pb->pcg_seq : 0ull, pb->stream); }
                pb->launches += 1;
                if (poll) {
                    int rc = check_async("pcg launch");
                    if (rc) return rc;
                    volatile unsigned long long *sq = reinterpret_cast<volatile unsigned long long *>(pb->h_pcg) + 3;
                    const auto t0 = std::chrono::steady_clock::now();
                    unsigned spins = 0;
                    int poll_rc = AAR_OK;
                    while (*sq != pb->pcg_seq) {
                        if ((++spins & 0x3fff) == 0) {
                            if (hipStreamQuery(pb->stream) == hipSuccess && *sq != pb->pcg_seq) {   // the stream drained but the record is stale: an asynchronous failure
                                poll_rc = set_error(AAR_ERR_HIP, "device did not publish the PCG progress record: %s", hipGetErrorString(hipGetLastError()));
                                break;
                            }
                            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                                poll_rc = set_error(AAR_ERR_HIP, "timed out waiting for the PCG progress record");
                                break;
                            }
                        }
                        __builtin_ia32_pause();
                    }
                    std::atomic_thread_fence(std::memory_order_acquire);
                    // (a rank whose device has faulted or hangs cannot tell the others: its stream carries nothing any more, and an agreement per poll --
                    //  one more collective -- would cost every healthy step ~50 us; the peers then sit in their next all-reduce until the launcher's
                    //  watchdog ends the job, as with any rank that dies mid-collective)
                    if (poll_rc) return poll_rc;
                    if (pb->h_pcg[0] != 0.0 || last) break;
                }
                { StageTimer t(pb, &pb->times.allreduce); int rc = allreduce(pb, pcgd_y_of_launch(P, k), (size_t)6 * P.A, NCCL_SUM); if (rc) return rc; }
            }
        }
        chol_done = true;
    }
    if (!P.use_pcg && pb->schur_mu != mu) {  // not already done speculatively by the try that produced this point
        StageTimer t(pb, &pb->times.schur);
        if (pb->s_reduced) {
            // multi-GPU, fused collective: S | rhs | g0 of this point already hold the ALL-REDUCED system for the predicted
            // damping, so the terms cannot be taken back rank by rank.  Rebuild this rank's shared blocks from its observations
            // (pass B alone: the frame blocks V, W, g_f of pass A are untouched) -- a mispredicted damping is rare.
            int rc = zero_block_set(pb, cur);
            if (rc) return rc;
            launch_passB(P, cur, pb->stream);
            pb->launches += 1;
            pb->s_reduced = false;
        } else if (pb->schur_mu >= 0) {
            // the speculative Schur complement was taken with another damping than the step now needs (gain < 0.94): take it
            // back with the inverses it used (still in Vinv), keeping the blocks -- and the residual they were built from --
            // exactly those of the accepted trial, as the reference's x64 / J are
            launch_schur(P, cur, -1.0, pb->stream, 0, 0, nullptr, panels_ok(pb, cur, pb->schur_mu));
            panels_now(pb, cur, pb->schur_mu);
            pb->launches += 1;
        }
        if (pb->vinv_mu != mu) {
            launch_frame_inv(P, cur, mu, pb->stream);
            pb->vinv_mu = mu;
            pb->launches += 1;
        }
        launch_schur(P, cur, 1.0, pb->stream, 0, 0, nullptr, panels_ok(pb, cur, mu));
        panels_now(pb, cur, mu);
        pb->launches += 1;
    }
    pb->schur_mu = -1;
    if (pb->comm && !pb->s_reduced && !P.use_pcg) {
        StageTimer t(pb, &pb->times.allreduce);
        int rc = allreduce_system(pb, cur, 0);   // S (lower triangle) | rhs | g0
        if (rc) return rc;
    }
    pb->s_reduced = false;      // the factorisation below consumes the system
    pb->trial_reduced = false;
    bool backsub_rode = false;
    if (P.use_spcg && !pb->force_direct && pb->spcg_skip > 0) pb->spcg_skip--;   // CG on the explicit reduced system instead of the LDL^T chain (S stays as it is)
    if (!chol_done) {
        StageTimer t(pb, &pb->times.chol);
        // (stage timers keep the frame back-substitution in its own launch, so that it has a time of its own)
        if (by_cg) backsub_rode = launch_spcg(P, cur, mu, pb->stream, pb->stage_timers ? -1 : tr);
        else backsub_rode = launch_chol(P, cur, mu, pb->stream, pb->stage_timers ? -1 : tr);
    }
    if (!backsub_rode) {
        StageTimer t(pb, &pb->times.backsub);
        launch_backsub(P, cur, tr, pb->stream);
    }
    pb->launches += 2 + 3 * P.nT;
    pb->blocks_valid = false;  // S of the current point has been eliminated in place
    if (evaluate_trial) {
        // predicted damping of the next step: every accepted step of the reference's rule with gain >= 0.94 gives 0.33 mu
        int rc = eval_blocks(pb, tr, mu * 0.33, cur, /*spec_schur=*/false, /*ents_ready=*/true);
        if (rc) return rc;
        pb->trial_points++;
    } else {
        HIP_TRY(hipMemsetAsync(P.err_part, 0, sizeof(double) * (size_t)std::max(P.F, 1), pb->stream));
    }
    // The scalars go to the host BEFORE the speculative Schur complement of the trial point is queued: the host takes its
    // accept / reject decision and queues the next factorisation while that kernel runs, instead of after it.
    // On one GPU the reduction does not even get a launch of its own: it rides as one more workgroup of that kernel.
    int rc = AAR_OK;
    bool rode = false;
    if (P.use_pcg) {   // nothing is eliminated ahead of the next step in this mode: only the step's scalars travel
        if ((rc = launch_scalars(pb, P.F))) return rc;
    } else if (evaluate_trial && !pb->comm) {
        StageTimer t(pb, &pb->times.schur);
        rode = launch_schur(P, tr, 1.0, pb->stream, pb->seq + 1, P.F, nullptr, panels_ok(pb, tr, mu * 0.33));
        panels_now(pb, tr, mu * 0.33);
        pb->launches += 1;
        if (rode) pb->seq++;
        else if ((rc = launch_scalars(pb, P.F))) return rc;   // (nothing rode: a rank without frames launches no Schur kernel)
    } else if (evaluate_trial && pb->fused_comm) {
        // Multi-GPU: this rank's scalars ride in the speculative Schur launch as on one GPU, but into the 8 doubles behind
        // g0 of the trial's block set, and ONE all-reduce then carries the step's scalars AND the next step's S | rhs | g0:
        // an accepted step with the predicted damping (the usual case) costs one collective, not two.
        {
            StageTimer t(pb, &pb->times.schur);
            rode = launch_schur(P, tr, 1.0, pb->stream, 0, P.F, P.blk[tr].tail, panels_ok(pb, tr, mu * 0.33));
            panels_now(pb, tr, mu * 0.33);
            pb->launches += 1;
            if (!rode) { launch_reduce_scalars(P, P.F, false, 0ull, pb->stream, P.blk[tr].tail); pb->launches += 1; }
        }
        pb->seq++;
        {
            StageTimer t(pb, &pb->times.allreduce);
            // tail[0..2] = sum r^2, sum |delta_f|^2, sum delta_f.g_f are rank sums, tail[3] carries every rank's error flags (so that
            // all ranks take the same branch); tail[5..6] (shared-parameter pieces) are computed from replicated data on every
            // rank and stay out of the reduction
            if ((rc = allreduce_system(pb, tr, 4, pb->seq))) return rc;   // (the unpacking kernel publishes the reduced scalars)
        }
        pb->trial_reduced = true;
        if (pb->spec_chol && !pb->stage_timers && !pb->profiling) {
            StageTimer t(pb, &pb->times.chol);
            // (the solver the NEXT try is expected to take: this try's own CG count is not known yet, the previous one's is)
            const bool spec_cg = P.use_spcg && pb->spcg_skip == 0 && !cg_near_cap;
            if (spec_cg) launch_spcg(P, tr, mu * 0.33, pb->stream);
            else (void)launch_chol(P, tr, mu * 0.33, pb->stream);
            pb->spec_chol_by_cg = spec_cg;
            pb->spec_chol_blk = tr;
            pb->spec_chol_mu = mu * 0.33;
            pb->spec_chol_eta = P.pcg_eta_now;   // (a forcing SEQUENCE may switch to the tight term for the next try: that try then solves again)
            pb->launches += 3 * P.nT;
        }
    } else {
        if ((rc = launch_scalars(pb, P.F))) return rc;
        if (evaluate_trial) {
            StageTimer t(pb, &pb->times.schur);
            launch_schur(P, tr, 1.0, pb->stream, 0, 0, nullptr, panels_ok(pb, tr, mu * 0.33));
            panels_now(pb, tr, mu * 0.33);
            pb->launches += 1;
        }
    }
    if ((rc = check_async("kernel launch"))) return rc;
    if ((rc = wait_result(pb))) return rc;
    if (by_cg && pb->h_scal[7] >= 0.0) pb->last_cg_its = (int)pb->h_scal[7];
    // every rank must take the same decisions from last_cg_its, and the flags ARE joined over the ranks: a solve that gave up anywhere counts as one at the cap everywhere
    // (a time-out records SPCG_BUFS on the rank it happened on only)
    if (by_cg && (pb->h_flags[0] & (8 | 4))) pb->last_cg_its = std::min(P.spcg_max_it, SPCG_MAX_IT);
    if (cg_near_cap && P.use_spcg && !pb->force_direct && pb->spcg_skip == 0 && ++pb->near_cap_tries >= 8) { pb->last_cg_its = -1; pb->near_cap_tries = 0; }   // (the latch decays)
    if (pb->h_flags[0]) {
        (void)hipMemsetAsync(P.flags, 0, 4 * sizeof(int32_t), pb->stream);
        if (by_cg && (pb->h_flags[0] & (8 | 4))) {
            // the CG solve gave up (8: iteration cap; 4: a wavefront of its grid never showed up within ~1 s -- the device is shared): whatever
            // followed it in this try is meaningless, but S was not touched.  Not an error: the caller redoes the try with the direct chain.
            if (pb->h_flags[0] & 4) spcg_ws_reset(P, pb->stream);   // (a timed-out launch leaves slots of both buffer sets in an unknown state)
            pb->spcg_fallbacks++;
            pb->spcg_backoff = pb->spcg_backoff ? std::min(2 * pb->spcg_backoff, 1024) : 8;
            pb->spcg_skip = pb->spcg_backoff;
            return TRY_CG_FAILED;
        }
        set_error(AAR_ERR_NUMERIC, "device flags %d at mu=%g (1: a frame block is not positive definite, 2: non-positive pivot of the reduced system, 4: back-substitution chain timed out)", pb->h_flags[0], mu);
        // Only the reduced system lost positive definiteness (far from the optimum its Schur complement can, in floating point):
        // the LM loop takes that as a failed try and raises the damping; every rank sees the same replicated pivots.
        // (a zero or negative pivot fills the trial point with NaNs, which the next pass A reports as flag 1 as well: still a failed try)
        // (a back-substitution chain that timed out, flag 4, is never a failed try: its delta is garbage whatever the pivots were)
        return ((pb->h_flags[0] & 2) && !(pb->h_flags[0] & 4)) ? TRY_NOT_POSITIVE_DEFINITE : AAR_ERR_NUMERIC;
    }
    return AAR_OK;
}

// Rebuild the blocks of the current point after a rejected try consumed them (rare): everything the trial wrote into
// blk[1-cur] is garbage as well.
int rebuild_current(aar_problem *pb) {
    pb->mu_seed_valid = false;   // (the seed belongs to the blocks aar_lm_init built)
    int rc = zero_block_set(pb, pb->cur);
    if (rc) return rc;
    if ((rc = zero_block_set(pb, 1 - pb->cur))) return rc;
    // B = -J^T x64 of the reference is computed ONCE per step(), from the residual of the last accepted evaluation
    // (libs/sparselevmarq.h:367), i.e. with the Huber delta in force THEN -- the step callback may have moved it since
    const float huber_now = pb->P.huber;
    if (pb->with_huber && pb->huber_of_blocks > 0.f) pb->P.huber = pb->huber_of_blocks;
    rc = eval_blocks(pb, pb->cur, -1.0, -1);
    pb->P.huber = huber_now;
    if (rc) return rc;
    pb->blocks_valid = true;
    pb->vinv_mu = -1;
    pb->schur_mu = -1;
    pb->s_reduced = pb->trial_reduced = false;
    return AAR_OK;
}

// damped_try, and -- solver spcg -- the same try once more with the direct chain when the CG solve gave up (iteration cap, hand-over
// time-out): the blocks of the current point are rebuilt (the abandoned try's trial evaluation cleared them), the damping stays
int damped_try_fb(aar_problem *pb, double mu, bool evaluate_trial) {
    int rc = damped_try(pb, mu, evaluate_trial);
    if (rc != TRY_CG_FAILED) return rc;
    if (evaluate_trial) pb->trial_points--;   // (the abandoned try's trial evaluation is not a residual evaluation of the LM loop)
    pb->trial_reduced = false;
    if ((rc = rebuild_current(pb))) return rc;
    pb->force_direct = true;
    rc = damped_try(pb, mu, evaluate_trial);
    pb->force_direct = false;
    return rc;
}

// x_full -> z of the reference for the problem's Config (mats2eVec order: cameras | markers | frames), and back
void extract_z(const PoseLayout &L, const double *x_full, double *z) {
    int64_t k = 0;
    if (L.oc) for (int64_t i = 0; i < 6LL * (L.C - 1); i++) z[k++] = x_full[L.full_cam0() + i];
    if (L.om) for (int64_t i = 0; i < 6LL * (L.M - 1); i++) z[k++] = x_full[L.full_mk0() + i];
    if (L.of) for (int64_t i = 0; i < 6LL * L.F; i++) z[k++] = x_full[L.full_fr0() + i];
    if (L.oi) for (int64_t i = 0; i < 9LL * L.C; i++) z[k++] = x_full[L.full_intr0() + i];
}
void merge_z(const PoseLayout &L, const double *z, double *x_full) {
    int64_t k = 0;
    if (L.oc) for (int64_t i = 0; i < 6LL * (L.C - 1); i++) x_full[L.full_cam0() + i] = z[k++];
    if (L.om) for (int64_t i = 0; i < 6LL * (L.M - 1); i++) x_full[L.full_mk0() + i] = z[k++];
    if (L.of) for (int64_t i = 0; i < 6LL * L.F; i++) x_full[L.full_fr0() + i] = z[k++];
    if (L.oi) for (int64_t i = 0; i < 9LL * L.C; i++) x_full[L.full_intr0() + i] = z[k++];
}

// curr_z on the host for a callback (a device -> host copy per call: only made when a callback asked for it)
int current_z_for_callbacks(aar_problem *pb, const double *x_start) {
    pb->cb_x.assign(x_start, x_start + pb->L.full_len());
    int rc = download_z(pb, pb->cur, pb->cb_x.data());
    if (rc) return rc;
    pb->cb_z.resize((size_t)std::max<int64_t>(pb->L.z_len(), 1));
    extract_z(pb->L, pb->cb_x.data(), pb->cb_z.data());
    return AAR_OK;
}

int initial_mu(aar_problem *pb, double tau, double *mu) {
    DeviceProblem &P = pb->P;
    const int cur = pb->cur;
    if (pb->comm) {
        // the diagonal of the shared blocks is a sum over ranks; the frame blocks are rank-local
        HIP_TRY(hipMemcpy2DAsync(pb->d_diag, sizeof(double), P.blk[cur].S, (size_t)(P.n_pad + 1) * sizeof(double), sizeof(double),
                                 P.n_pad, hipMemcpyDeviceToDevice, pb->stream));
        int rc = allreduce(pb, pb->d_diag, P.n_pad, NCCL_SUM);
        if (rc) return rc;
        // the trial block set's S is zero and unused at this point: borrow its diagonal for the summed values
        HIP_TRY(hipMemcpy2DAsync(P.blk[1 - cur].S, (size_t)(P.n_pad + 1) * sizeof(double), pb->d_diag, sizeof(double), sizeof(double),
                                 P.n_pad, hipMemcpyDeviceToDevice, pb->stream));
        DeviceProblem Q = P;
        Q.blk[cur].S = P.blk[1 - cur].S;
        launch_maxdiag(Q, cur, pb->stream);
        HIP_TRY(hipMemsetAsync(P.blk[1 - cur].S, 0, (size_t)P.n_pad * P.n_pad * sizeof(double), pb->stream));
        rc = allreduce(pb, P.scal + 4, 1, NCCL_MAX);
        if (rc) return rc;
    } else {
        if (pb->schur_mu >= 0) {   // (the init head start has already reduced S for its damping: the diagonal of J^T J is read from the blocks, so give the terms back first)
            launch_schur(P, cur, -1.0, pb->stream, 0, 0, nullptr, panels_ok(pb, cur, pb->schur_mu));
            panels_now(pb, cur, pb->schur_mu);
            pb->schur_mu = -1;
            pb->launches += 1;
        }
        launch_maxdiag(P, cur, pb->stream);
    }
    pb->launches += 1;
    pb->seq++;
    launch_publish(P, pb->seq, pb->stream);
    int rc2 = wait_result(pb);
    if (rc2) return rc2;
    *mu = pb->h_scal[4] * tau;
    return AAR_OK;
}

}  // namespace

extern "C" {

int aar_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int aar_device_synchronize(void) {
    HIP_TRY(hipDeviceSynchronize());
    return AAR_OK;
}

int aar_comm_make_id(char id[AAR_COMM_ID_BYTES]) {
    int rc = load_nccl();
    if (rc) return rc;
    NcclId u;
    NCCL_TRY(g_nccl.GetUniqueId(&u));
    memcpy(id, u.internal, AAR_COMM_ID_BYTES);
    return AAR_OK;
}

int aar_comm_create(const char id[AAR_COMM_ID_BYTES], int32_t world_size, int32_t rank, int32_t device_id, aar_comm **out) {
    if (!id || !out || world_size < 1 || rank < 0 || rank >= world_size) return set_error(AAR_ERR_INVALID, "aar_comm_create: bad arguments");
    int rc = load_nccl();
    if (rc) return rc;
    rc = ensure_device(device_id);
    if (rc) return rc;
    NcclId u;
    memcpy(u.internal, id, AAR_COMM_ID_BYTES);
    aar_comm *c = new aar_comm();
    c->world = world_size; c->rank = rank; c->device = device_id;
    int r = g_nccl.CommInitRank(&c->comm, world_size, u, rank);
    if (r != 0) {
        delete c;
        return set_error(AAR_ERR_COMM, "ncclCommInitRank failed: %s", g_nccl.GetErrorString ? g_nccl.GetErrorString(r) : "?");
    }
    *out = c;
    return AAR_OK;
}

int aar_comm_get_stats(const aar_comm *c, aar_comm_stats *out) {
    if (!c || !out) return set_error(AAR_ERR_INVALID, "aar_comm_get_stats: null argument");
    memset(out, 0, sizeof *out);
    out->world_size = c->world;
    out->rank = c->rank;
    out->ranks_seen = c->world;
    if (c->comm && g_nccl.CommCount) {   // what RCCL itself reports for this communicator
        int n = 0;
        NCCL_TRY(g_nccl.CommCount(c->comm, &n));
        out->ranks_seen = n;
    }
    out->allreduce_calls = c->allreduce_calls;
    out->allreduce_bytes = c->allreduce_bytes;
    out->system_allreduce_bytes = c->last_system_bytes;
    return AAR_OK;
}

void aar_comm_destroy(aar_comm *c) {
    if (!c) return;
    if (c->comm && g_nccl.ok) g_nccl.CommDestroy(c->comm);
    if (c->tmp) (void)hipFree(c->tmp);
    delete c;
}

int aar_local_group_create(int32_t world_size, aar_local_group **out) {
    if (!out || world_size < 1 || world_size > 64) return set_error(AAR_ERR_INVALID, "aar_local_group_create: bad arguments");
    aar_local_group *g = new aar_local_group();
    g->world = world_size;
    g->ptrs.assign(world_size, nullptr);
    *out = g;
    return AAR_OK;
}

void aar_local_group_destroy(aar_local_group *g) { delete g; }

int aar_comm_create_local(aar_local_group *group, int32_t rank, int32_t device_id, aar_comm **out) {
    if (!group || !out || rank < 0 || rank >= group->world) return set_error(AAR_ERR_INVALID, "aar_comm_create_local: bad arguments");
    int rc = ensure_device(device_id);
    if (rc) return rc;
    aar_comm *c = new aar_comm();
    c->world = group->world; c->rank = rank; c->device = device_id; c->local = group;
    *out = c;
    return AAR_OK;
}

void aar_lm_default_params(aar_lm_params *p) {  // libs/multicam_mapper.cpp:326-330 over libs/sparselevmarq.h:41-49
    p->max_iters = 10000;
    p->min_error = 1e-5;
    p->min_step_error_diff = 0;
    p->min_average_step_error_diff = 1e-4;
    p->tau = 1;
    p->verbose = 0;
}

void aar_problem_destroy(aar_problem *pb) {
    if (!pb) return;
    (void)hipSetDevice(pb->device);
    if (pb->stream) (void)hipStreamSynchronize(pb->stream);
    for (void *p : pb->allocs) (void)hipFree(p);
    if (pb->h_scal) (void)hipHostFree(pb->h_scal);
    if (pb->h_pcg) (void)hipHostFree(pb->h_pcg);
    if (pb->h_gather) (void)hipHostFree(pb->h_gather);
    if (pb->up_ev) (void)hipEventDestroy(pb->up_ev);
    if (pb->ev[0]) (void)hipEventDestroy(pb->ev[0]);
    if (pb->ev[1]) (void)hipEventDestroy(pb->ev[1]);
    for (hipEvent_t e : pb->ev_pool) (void)hipEventDestroy(e);
    if (pb->ev_fork) (void)hipEventDestroy(pb->ev_fork);
    if (pb->ev_join) (void)hipEventDestroy(pb->ev_join);
    if (pb->stream2) { (void)hipStreamSynchronize(pb->stream2); (void)hipStreamDestroy(pb->stream2); }
    if (pb->stream) (void)hipStreamDestroy(pb->stream);
    delete pb;
}

void aar_solver_default_options(aar_solver_options *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->struct_size = (uint32_t)sizeof *o;
    o->solver = AAR_SOLVER_AUTO;
}

int aar_problem_create(const aar_problem_desc *d, aar_problem **out) { return aar_problem_create_ex(d, nullptr, out); }

// AAR_ABORT_BACKTRACE=<file> (diagnostics): the native stack of whoever raises SIGABRT in this process (a runtime library giving up) is appended to the file before the default action
static int abort_bt_fd = 2;
static void abort_backtrace(int sig) {
    void *bt[64];
    { char msg[64]; const int k = aar::g_last_kernel_id; int n = 0; const char *t = "last kernel id "; while (t[n]) { msg[n] = t[n]; n++; } if (k < 0) msg[n++] = '-'; else { if (k >= 10) msg[n++] = '0' + k / 10; msg[n++] = '0' + k % 10; } msg[n++] = '\n'; (void)!write(abort_bt_fd, msg, n); }
    const int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, abort_bt_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}

int aar_problem_create_ex(const aar_problem_desc *d, const aar_solver_options *opts, aar_problem **out) {
    { static bool once = false; if (!once && getenv("AAR_ABORT_BACKTRACE")) { once = true; const int fd = open(getenv("AAR_ABORT_BACKTRACE"), O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) abort_bt_fd = fd; signal(SIGABRT, abort_backtrace); } }
    if (!d || !out) return set_error(AAR_ERR_INVALID, "aar_problem_create: null argument");
    aar_solver_options so;
    aar_solver_default_options(&so);
    if (opts) {   // a caller built against an older header passes a shorter struct: the fields it does not know keep their defaults
        if (opts->struct_size < 2 * sizeof(uint32_t)) return set_error(AAR_ERR_INVALID, "aar_solver_options.struct_size is not set (use aar_solver_default_options)");
        memcpy(&so, opts, std::min<size_t>(opts->struct_size, sizeof so));
        so.struct_size = (uint32_t)sizeof so;
    }
    // environment overrides (tuning / bisecting only: the options struct is the interface): they apply to fields the caller left at their defaults
    // -- an explicitly chosen solver, forcing term or cap always wins -- and are reported (aar_solver_stats.env_overrides)
    int env_over = 0;
    if (const char *e = getenv("AAR_SOLVER")) {
        if (so.solver == AAR_SOLVER_AUTO) {
            int v = -1;
            if (!strcmp(e, "direct")) v = AAR_SOLVER_DIRECT;
            else if (!strcmp(e, "pcg")) v = AAR_SOLVER_PCG;
            else if (!strcmp(e, "spcg")) v = AAR_SOLVER_SPCG;
            if (v >= 0) { so.solver = v; env_over |= AAR_ENV_SOLVER; }
        }
    }
    if (const char *e = getenv("AAR_DETERMINISTIC")) if (!so.deterministic && atoi(e) != 0) { so.deterministic = 1; env_over |= AAR_ENV_DETERMINISTIC; }
    if (const char *e = getenv("AAR_PCG_ETA")) if (so.pcg_eta == 0 && atof(e) > 0) { so.pcg_eta = atof(e); env_over |= AAR_ENV_PCG_ETA; }
    if (const char *e = getenv("AAR_PCG_MAX_IT")) if (so.pcg_max_it == 0 && atoi(e) > 0) { so.pcg_max_it = atoi(e); env_over |= AAR_ENV_PCG_MAX_IT; }
    if (so.solver < AAR_SOLVER_DIRECT || so.solver > AAR_SOLVER_AUTO) return set_error(AAR_ERR_INVALID, "aar_solver_options.solver %d is not one of AAR_SOLVER_*", so.solver);
    if (so.pcg_eta < 0 || so.pcg_max_it < 0 || so.pcg_eta_loose < 0 || so.pcg_eta_switch < 0 || so.pcg_abs_tol < 0) return set_error(AAR_ERR_INVALID, "aar_solver_options: negative pcg_eta / pcg_eta_loose / pcg_eta_switch / pcg_abs_tol / pcg_max_it");
    const int C = d->num_cams, M = d->num_markers, Fg = d->num_frames;
    const int64_t Ng = d->num_obs;
    if (C < 1 || M < 1 || Fg < 0 || Ng < 0) return set_error(AAR_ERR_INVALID, "aar_problem_create: bad sizes");
    if (d->root_cam < 0 || d->root_cam >= C || d->root_marker < 0 || d->root_marker >= M)
        return set_error(AAR_ERR_INVALID, "aar_problem_create: root index out of range");
    if (!d->cam_mats || (Ng > 0 && (!d->obs_frame || !d->obs_cam || !d->obs_marker || !d->obs_uv)))
        return set_error(AAR_ERR_INVALID, "aar_problem_create: null array");
    if (Ng >= (1LL << 31) || (int64_t)C + M >= 32768) return set_error(AAR_ERR_UNSUPPORTED, "problem too large for 32-bit indexing");
    std::vector<int64_t> per_frame(Fg, 0);
    // (entity, frame) incidences of the WHOLE data set, counted the same way on every rank: what AUTO's choice between the two CG solvers rests on must not
    // depend on which frames a rank happens to own (ranks that chose differently would wait in different collectives forever)
    int64_t global_slots = 0;
    {
        std::vector<int32_t> seen_c(C, -1), seen_m(M, -1);
        for (int64_t o = 0; o < Ng; o++) {
            const int f = d->obs_frame[o], c = d->obs_cam[o], m = d->obs_marker[o];
            if (f < 0 || f >= Fg || c < 0 || c >= C || m < 0 || m >= M) return set_error(AAR_ERR_INVALID, "observation %lld has an index out of range", (long long)o);
            if (o > 0 && f < d->obs_frame[o - 1]) return set_error(AAR_ERR_INVALID, "observations must be ordered by frame (reference residual order)");
            per_frame[f]++;
            if (seen_c[c] != f) { seen_c[c] = f; global_slots += d->optimize_cam_intrinsics ? 2 : 1; }
            if (seen_m[m] != f) { seen_m[m] = f; global_slots += 1; }
        }
    }
    int rc = ensure_device(d->device_id);
    if (rc) return rc;

    aar_problem *pb = new aar_problem();
    pb->device = d->device_id;
    pb->comm = d->comm;
    aar_lm_default_params(&pb->prm);
    memset(&pb->times, 0, sizeof pb->times);
    const char *tenv = getenv("AAR_STAGE_TIMERS");
    pb->stage_timers = tenv && tenv[0] == '1';
    auto fail = [&](int code) { aar_problem_destroy(pb); return code; };
    if (hipStreamCreateWithFlags(&pb->stream, hipStreamNonBlocking) != hipSuccess) return fail(set_error(AAR_ERR_HIP, "hipStreamCreate failed"));
    (void)hipEventCreate(&pb->ev[0]);
    (void)hipEventCreate(&pb->ev[1]);
    if (hipStreamCreateWithFlags(&pb->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&pb->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&pb->ev_join, hipEventDisableTiming) != hipSuccess)
        return fail(set_error(AAR_ERR_HIP, "second stream / events could not be created"));
    { const char *e = getenv("AAR_OVERLAP"); pb->overlap = (e && e[0] == '1'); }  // measured slower than one stream at config 3: off by default
    { const char *e = getenv("AAR_MERGE_PASSES"); if (e) pb->merge_passes = (e[0] != '0'); }
    { const char *e = getenv("AAR_FUSED_COMM"); if (e) pb->fused_comm = (e[0] != '0'); }
    { const char *e = getenv("AAR_SPEC_CHOL"); if (e) pb->spec_chol = (e[0] != '0'); }

    PoseLayout &L = pb->L;
    L.C = C; L.M = M; L.F = Fg; L.rc = d->root_cam; L.rm = d->root_marker;
    L.oc = d->optimize_cam_poses != 0; L.om = d->optimize_marker_poses != 0; L.of = d->optimize_object_poses != 0;
    L.oi = d->optimize_cam_intrinsics != 0;
    pb->N_global = Ng;

    // ---- shard by frame range ----
    const int world = pb->comm ? pb->comm->world : 1, rank = pb->comm ? pb->comm->rank : 0;
    pb->P.pcg_rank0 = rank == 0 ? 1 : 0;
    std::vector<int32_t> begin(world + 1, 0);
    if ((rc = aar_plan_shards(Fg, per_frame.data(), world, begin.data()))) return fail(rc);
    pb->f_begin = begin[rank];
    pb->f_end = begin[rank + 1];
    int64_t ob = 0;
    for (int f = 0; f < pb->f_begin; f++) ob += per_frame[f];
    int64_t N = 0;
    for (int f = pb->f_begin; f < pb->f_end; f++) N += per_frame[f];
    pb->o_begin = ob;

    DeviceProblem &P = pb->P;
    // optimize_cam_intrinsics: one more shared entity per camera (fx, cx, fy, cy and two idle slots), after cameras and markers
    P.intr = L.oi ? 1 : 0;
    P.C = C; P.M = M; P.A = C + M + (L.oi ? C : 0); P.F = pb->f_end - pb->f_begin; P.N = N;
    P.n = 6 * P.A;
    P.nT = (P.n + CHOL_NB - 1) / CHOL_NB;
    P.n_pad = P.nT * CHOL_NB;
    // solver spcg, coarse space (spcg_kernels.hip, k_spcg_pre): a group with free entities lends its root's slot to the group's rigid-motion unknowns
    P.spcg_root_c = (L.oc && C > 1) ? L.rc : -1;
    P.spcg_root_m = (L.om && M > 1) ? C + L.rm : -1;
    P.spcg_n_free = (L.oc ? C - 1 : 0) + (L.om ? M - 1 : 0) + (L.oi ? C : 0);
    P.res_f32 = d->residual_mode == AAR_RES_F64 ? 0 : 1;
    pb->with_huber = d->with_huber != 0;
    P.huber = pb->with_huber ? pb->hubber_delta : -1.f;
    P.half_size = (double)((float)d->marker_size / 2.f);
    P.frames_fixed = L.of ? 0 : 1;
    const int A = P.A, F = P.F;

    int local_rc = AAR_OK;   // rank-local limits: decided collectively below
    // ---- ordering A (reference order) + per-frame slot lists ----
    std::vector<int32_t> frame_obs_start(F + 1, 0), fslot_start(F + 1, 0), fslot_ent;
    std::vector<ObsIdx> a_idx(N);
    std::vector<float> a_uv((size_t)8 * N);
    if (N) memcpy(a_uv.data(), d->obs_uv + 8 * ob, sizeof(float) * 8 * N);
    {
        int64_t o = 0;
        std::vector<int32_t> slot_of(A, -1);
        for (int f = 0; f < F; f++) {
            frame_obs_start[f] = (int32_t)o;
            const int64_t cnt = per_frame[pb->f_begin + f];
            std::vector<int32_t> ents;
            for (int64_t k = 0; k < cnt; k++) {
                const int64_t g = ob + o + k;
                ents.push_back(d->obs_cam[g]);
                ents.push_back(C + d->obs_marker[g]);
                if (L.oi) ents.push_back(C + M + d->obs_cam[g]);
            }
            std::sort(ents.begin(), ents.end());
            ents.erase(std::unique(ents.begin(), ents.end()), ents.end());
            if (ents.size() >= (1u << SLOT_C_BITS) && !local_rc) local_rc = set_error(AAR_ERR_UNSUPPORTED, "frame %d touches %zu entities (limit %d)", f, ents.size(), (1 << SLOT_C_BITS) - 1);
            for (size_t s = 0; s < ents.size(); s++) slot_of[ents[s]] = (int32_t)s;
            for (int64_t k = 0; k < cnt; k++) {
                const int64_t g = ob + o + k;
                ObsIdx id;
                id.frame = f; id.cam = d->obs_cam[g]; id.marker = C + d->obs_marker[g];
                id.slots = pack_slots(slot_of[id.cam], slot_of[id.marker], L.oi ? slot_of[C + M + id.cam] : 0);
                a_idx[o + k] = id;
            }
            fslot_start[f] = (int32_t)fslot_ent.size();
            fslot_ent.insert(fslot_ent.end(), ents.begin(), ents.end());
            P.max_kf = std::max<int>(P.max_kf, (int)ents.size());
            o += cnt;
        }
        frame_obs_start[F] = (int32_t)o;
        fslot_start[F] = (int32_t)fslot_ent.size();
    }
    P.total_slots = (int)fslot_ent.size();
    std::vector<int32_t> frame_stride(F, 1);   // the observations of a frame are camera-major: a stride of ~ n / 8, coprime with n, puts different cameras into neighbouring lanes
    for (int f = 0; f < F; f++) {
        const int n = frame_obs_start[f + 1] - frame_obs_start[f];
        if (n <= 16) continue;
        int st = n / 8 + 1;
        while (std::gcd(st, n) != 1) st++;
        frame_stride[f] = st;
    }
    {   // tuning switches of this problem (kernels.h, DeviceProblem::Tuning)
        auto env_int = [](const char *name, int &v) { if (const char *e = getenv(name)) v = atoi(e); };
        env_int("AAR_FUSED_PANEL", P.tune.fused_panel); env_int("AAR_BS_RIDES", P.tune.bs_rides); env_int("AAR_BACKSUB_RIDES", P.tune.backsub_rides);
        env_int("AAR_LDL_LOOKAHEAD", P.tune.lookahead); env_int("AAR_PASSA_VARIANT", P.tune.passA_variant); env_int("AAR_PACK_SYSTEM", P.tune.pack_system);
        env_int("AAR_INIT_HEADSTART", P.tune.init_headstart); env_int("AAR_PASSB_LEAN", P.tune.passB_lean); env_int("AAR_PASSA_WRENCH", P.tune.passA_wrench); env_int("AAR_PASSB_WRENCH_MERGED", P.tune.passB_wrench_merged); env_int("AAR_SPCG_BACKSUB_RIDES", P.tune.spcg_backsub_rides); env_int("AAR_PASSAB_OCC2", P.tune.passAB_occ2);
    }
    // the frame-block kernel keeps a frame's slots in LDS: sized by the form that is actually launched (wrench form: 21 + 4 doubles per slot; row form: 62)
    const size_t ldsA = P.tune.passA_wrench ? passA_wrench_lds_bytes(P.max_kf, L.oi)
                                            : std::max(passA_lds_bytes(P.max_kf, 256), passA_lds_bytes(P.max_kf, 64));
    if (ldsA > 160 * 1024 && !local_rc) local_rc = set_error(AAR_ERR_UNSUPPORTED, "a frame touches %d cameras+markers: %zu bytes of LDS in the frame-block kernel (%s form), the CU has 160 KiB", P.max_kf, ldsA, P.tune.passA_wrench ? "wrench" : "row");
    // Which Schur kernel (solve_kernels.hip): the MFMA kernel works on dense per-frame panels and is the default from 96 shared
    // entities, where the output-stationary kernel's re-reads of W dominate (config 5); AAR_SCHUR_MFMA=0 / 1 forces it (tests
    // force it on small problems).  Its panels cost 2 F Ad 288 bytes -- tens of GB for long sequences with many rarely seen
    // entities -- so they must fit a budget (half of the free device memory; AAR_SCHUR_PANEL_MB overrides), else the
    // output-stationary kernel, which needs none of this, takes over.  Only THAT kernel keeps a row panel of all entities in LDS.
    P.deterministic = so.deterministic ? 1 : 0;
    { const char *e = getenv("AAR_DENSE_FROM_PASSA"); if (e) P.dense_from_passA = atoi(e) != 0 ? 1 : 0; }
    {   // which solver (aar_solver_options; AUTO: DESIGN.md section 12)
        hipDeviceProp_t prop;
        const int cus = (hipGetDeviceProperties(&prop, pb->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 64;
        P.n_cus = cus;
        const bool pcg_ok = pcg_lds_bytes(A) <= 150 * 1024;
        // one wavefront per entity, every one of them resident AT ONCE (they hand over to each other): asked of the runtime's occupancy calculator for this
        // kernel's registers and LDS, as pcg_max_grid asks for the PCG grid.  (Another process on the device can still take the slots: k_spcg's time-out.)
        if (const char *t = getenv("AAR_SPCG_COARSE")) P.spcg_coarse = atoi(t) != 0;
        if (const char *t = getenv("AAR_SPCG_COARSE_FROM")) P.spcg_coarse_from = atoi(t);
        if (const char *t = getenv("AAR_PCG_COARSE")) P.pcg_coarse = atoi(t) != 0;
        if (const char *t = getenv("AAR_PCG_COARSE_FROM")) P.pcg_coarse_from = atoi(t);
        // the coarse operator's pass costs like ~1.5 CG iterations and a kept operator ~0.1 - 0.7 more iterations per solve: keeping it pays on long sequences only
        // (profiles/r06_attempts.txt section 3: +3.7 % at 160 entities x 4 000 frames, +1.4 % at config 5, -1 .. -3 % on 500-frame problems)
        P.pcg_e_every = P.total_slots >= 300000 ? 3 : 1;
        if (const char *t = getenv("AAR_PCG_E_EVERY")) P.pcg_e_every = std::max(1, atoi(t));
        if (const char *t = getenv("AAR_PCG_RESIDENT")) P.pcg_resident = atoi(t) != 0;
        if (pcg_lds_bytes(A, true) > 150 * 1024) P.pcg_coarse = 0;   // (the coarse space's tables do not fit beside the vectors of this many entities: block-Jacobi only)
        const int spcg_per_cu = spcg_fits(P.nT) ? spcg_resident_per_cu(P.nT, spcg_coarse_now(P)) : 0;
        const bool spcg_ok = spcg_fits(P.nT) && P.n_pad / 6 <= std::max(1, spcg_per_cu) * cus;
        int solver = so.solver;
        if (solver == AAR_SOLVER_AUTO) {
            // Measured on MI355X at the default forcing terms (profiles/r05_auto_crossover.txt; scripts/dev/auto_crossover.py):
            //  * one tile of unknowns (up to 16 cameras + markers): the direct chain is a single 25-us launch -- about what 10 CG iterations cost;
            //  * CG on the EXPLICIT Schur complement (SPCG) beats both the LDL^T chain and the CG through the frame blocks wherever it fits (up to 224 shared
            //    entities), 48 .. 216 entities x 500 frames: 1.6x .. 1.4x the direct chain, 3.2x .. 1.1x PCG;
            //  * PCG never forms the complement: its step costs (CG iterations) x (a pass over the frames' W blocks -- fp32 since round 5), SPCG's one Schur complement
            //    (work ~ slots x slots-per-frame) + CG iterations of ~1.7 us.  PCG overtakes on long sequences of frames that each see many entities: measured
            //    crossovers (PCG with fp32 blocks and the barrier tree) at ~45 k (entity, frame) incidences for 122 per frame (216 entities), ~85 k for 92 (160
            //    entities), ~95 k for 64 (112 entities) -- fitted by  incidences x (incidences per frame - 30) >= 4e6.
            // The rule is applied to a rank's SHARE of the whole data set (shards are balanced by observation count), from numbers every rank holds.
            const double kf_avg = Fg > 0 ? (double)global_slots / (double)Fg : 0.0;
            const bool pcg_pays = A >= 96 && pcg_ok && (double)global_slots / (double)world * (kf_avg - 30.0) >= 4e6;
            if (P.nT < 2) solver = AAR_SOLVER_DIRECT;
            else if (spcg_ok && !pcg_pays) solver = AAR_SOLVER_SPCG;
            else if (pcg_ok) solver = AAR_SOLVER_PCG;
            else solver = AAR_SOLVER_DIRECT;
        }
        pb->solver = solver;
        pb->env_overrides = env_over;
        P.use_pcg = solver == AAR_SOLVER_PCG ? 1 : 0;
        P.use_spcg = solver == AAR_SOLVER_SPCG ? 1 : 0;
        if (const char *t = getenv("AAR_SPCG_SPREAD")) P.spcg_spread = std::max(1, atoi(t));
        // (one XCD has an eighth of the CUs: more entities than that many wavefront slots would not all be resident there)
        if (P.n_pad / 6 > std::min(4, std::max(1, spcg_per_cu)) * (cus / 8)) P.spcg_spread = 1;
        // Forcing term of the inexact solvers (include/aar.h; defaults and why: kernels.h).  The CG through the frame blocks measures |r| / |b|; the CG on the
        // explicit system measures in the preconditioner's norm (r^T M^-1 r, which its recurrences carry anyway).  pcg_eta_loose > pcg_eta: a forcing sequence (opt-in).
        P.pcg_eta = P.use_spcg ? SPCG_ETA_DEFAULT : PCG_ETA_DEFAULT;
        if (so.pcg_eta > 0) P.pcg_eta = so.pcg_eta;
        P.pcg_eta_loose = 0.0;
        if (so.pcg_eta_loose > 0) P.pcg_eta_loose = so.pcg_eta_loose;
        else if (so.pcg_eta == 0 && (P.use_pcg || P.use_spcg)) P.pcg_eta_loose = P.use_pcg ? PCG_ETA_LOOSE_DEFAULT : SPCG_ETA_LOOSE_DEFAULT;   // (0: none)
        if (const char *e = getenv("AAR_PCG_ETA_LOOSE")) if (so.pcg_eta_loose == 0) P.pcg_eta_loose = atof(e);
        if (P.pcg_eta_loose <= P.pcg_eta) P.pcg_eta_loose = 0.0;   // (no sequence: one forcing term throughout)
        P.pcg_abs_tol = P.use_pcg ? PCG_ABS_TOL_DEFAULT : SPCG_ABS_TOL_DEFAULT;
        if (so.pcg_abs_tol > 0) P.pcg_abs_tol = so.pcg_abs_tol;
        else if (const char *e = getenv("AAR_PCG_ABS_TOL")) P.pcg_abs_tol = atof(e);
        if (so.pcg_eta_switch > 0) P.pcg_eta_switch = so.pcg_eta_switch;
        else if (const char *e = getenv("AAR_PCG_ETA_SWITCH")) P.pcg_eta_switch = atof(e);
        P.pcg_eta_now = P.pcg_eta;
        P.spcg_max_it = spcg_default_cap(P.nT);
        if (so.pcg_max_it > 0) { P.pcg_max_it = so.pcg_max_it; P.spcg_max_it = std::min(so.pcg_max_it, SPCG_MAX_IT); }
        if (P.use_pcg && !pcg_ok && !local_rc) local_rc = set_error(AAR_ERR_UNSUPPORTED, "AAR_SOLVER_PCG keeps the CG vectors and the preconditioner of %d unknowns in LDS: too many shared entities", 6 * A);
        if (P.use_spcg && !spcg_ok && !local_rc) local_rc = set_error(AAR_ERR_UNSUPPORTED, "AAR_SOLVER_SPCG keeps the reduced system in the registers of one wavefront per shared entity: %d unknowns are too many (limit %d, and at most %d entities)", 6 * A, 96 * SPCG_MAX_NT, std::max(1, spcg_per_cu) * cus);
    }
    bool schur_mfma = A >= 96 && F > 0;
    if (const char *e = getenv("AAR_SCHUR_MFMA")) schur_mfma = atoi(e) != 0 && F > 0;
    if (P.deterministic) schur_mfma = false;   // fixed-order sums exist for the output-stationary kernel only (kernels.h)
    if (P.use_pcg) schur_mfma = true;          // (no Schur complement is ever formed in that mode: no LDS row panel to fit; the dense panels are not allocated either)
    if (schur_mfma) {
        const size_t panel_bytes = (size_t)2 * F * ((A + 1 + 31) / 32 * 32) * 288;
        size_t free_b = 0, total_b = 0;
        size_t budget = (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? free_b / 2 : (size_t)64 << 30;
        if (const char *e = getenv("AAR_SCHUR_PANEL_MB")) budget = (size_t)std::max(0, atoi(e)) << 20;
        if (panel_bytes > budget) schur_mfma = false;
    }
    if (!schur_mfma && (size_t)A * 36 * 8 + 2048 > 160 * 1024 && !local_rc)
        local_rc = set_error(AAR_ERR_UNSUPPORTED, "%d cameras+markers exceed the row panel the output-stationary Schur kernel holds in LDS (and the dense panels of the MFMA kernel do not fit the memory budget)", A);
    {   // the limits above depend on the rank's own frames: agree on the outcome before anybody returns (see collective_status)
        if (pb->comm && (rc = dev_alloc(pb, &pb->d_status, 1))) return fail(rc);
        int agreed = local_rc;
        if ((rc = collective_status(pb, local_rc, &agreed))) return fail(rc);
        if (local_rc) return fail(local_rc);
        if (agreed) return fail(set_error(agreed, "another rank could not create its shard of the problem (status %d)", agreed));
    }

    // ---- ordering B: (camera, marker, frame) runs cut into wave-sized chunks ----
    std::vector<int32_t> perm(N);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) {
        const ObsIdx &p = a_idx[x], &q = a_idx[y];
        if (p.cam != q.cam) return p.cam < q.cam;
        if (p.marker != q.marker) return p.marker < q.marker;
        return p.frame < q.frame;
    });
    std::vector<ObsIdx> b_idx(N);
    std::vector<float> b_uv((size_t)8 * N);
    for (int64_t i = 0; i < N; i++) {
        b_idx[i] = a_idx[perm[i]];
        memcpy(&b_uv[8 * i], &a_uv[8 * (size_t)perm[i]], 8 * sizeof(float));
    }
    // (a wavefront pays a wave sum of 90 values and 90 atomics per chunk whatever its length: config 5's runs of ~390 observations as ONE chunk each, 3 200 wavefronts,
    //  run 11 % faster than cut at 256 -- profiles/r04_attempts.txt section 17; small problems keep 64 and the parallelism)
    int chunk_len = (int)((N / 2048 + 63) / 64 * 64);
    chunk_len = std::min(std::max(chunk_len, 64), 512);
    if (const char *e = getenv("AAR_PASSB_CHUNK")) chunk_len = std::min(std::max(atoi(e) / 64 * 64, 64), PASSB_CHUNK);   // tuning knob (observations of a run per wavefront)
    std::vector<int32_t> chunk_start;
    for (int64_t i = 0; i < N;) {
        int64_t e = i;
        while (e < N && b_idx[e].cam == b_idx[i].cam && b_idx[e].marker == b_idx[i].marker) e++;
        for (int64_t s = i; s < e; s += chunk_len) chunk_start.push_back((int32_t)s);
        i = e;
    }
    P.n_chunks = (int)chunk_start.size();
    chunk_start.push_back((int32_t)N);

    // ---- which off-diagonal blocks of U exist: entities that share an observation (camera x marker; with intrinsics entities also those of the camera) ----
    std::vector<int32_t> up_start(A + 1, 0), up_ent;
    {
        std::vector<std::vector<int32_t>> nb(A);
        for (int ch = 0; ch < P.n_chunks; ch++) {
            const ObsIdx &h = b_idx[chunk_start[ch]];
            if (ch > 0 && b_idx[chunk_start[ch - 1]].cam == h.cam && b_idx[chunk_start[ch - 1]].marker == h.marker) continue;
            int ents[3] = {h.cam, h.marker, L.oi ? C + M + h.cam : -1};
            for (int x = 0; x < 3; x++)
                for (int y = 0; y < 3; y++)
                    if (x != y && ents[x] >= 0 && ents[y] >= 0) nb[ents[x]].push_back(ents[y]);
        }
        for (int a = 0; a < A; a++) {
            std::sort(nb[a].begin(), nb[a].end());
            nb[a].erase(std::unique(nb[a].begin(), nb[a].end()), nb[a].end());
            up_start[a] = (int32_t)up_ent.size();
            up_ent.insert(up_ent.end(), nb[a].begin(), nb[a].end());
        }
        up_start[A] = (int32_t)up_ent.size();
        if (up_ent.empty()) up_ent.push_back(0);
    }

    // ---- (entity, frame) incidence for the Schur complement ----
    std::vector<std::vector<std::pair<int32_t, int32_t>>> inc(A);
    for (int f = 0; f < F; f++)
        for (int s = fslot_start[f]; s < fslot_start[f + 1]; s++) inc[fslot_ent[s]].push_back({f, s});
    std::vector<int4> pair_rec;
    std::vector<int32_t> sw_ent, sw_begin, sw_end;
    const int64_t total_pairs = P.total_slots;
    // (entity, frame) pairs per workgroup: measured optimum 16 (config 3) ... 64 (configs 4, 5); every workgroup ends with
    // an atomic flush of its row panel, so fewer, longer items win once the grid is large enough to fill the chip
    int per_item = (int)std::min<int64_t>(64, std::max<int64_t>(16, total_pairs / 1024));
    per_item = (per_item + 3) / 4 * 4;
    if (const char *e = getenv("AAR_SCHUR_ITEM")) per_item = std::max(4, atoi(e));  // tuning knob (pairs per workgroup)
    std::vector<int> pair_base(A + 1, 0);
    for (int a = 0; a < A; a++) {
        pair_base[a] = (int)pair_rec.size();
        for (auto &pr : inc[a]) pair_rec.push_back(make_int4(pr.first, pr.second, fslot_start[pr.first], 0));
    }
    pair_base[A] = (int)pair_rec.size();
    // Large problems (the kernel is then bound by re-reading W blocks): items are cut by FRAME WINDOW and ordered window-major,
    // so that the workgroups in flight at any time walk the same frames and find each other's W blocks in L2 / MALL.
    // Small problems: plain runs of per_item pairs, entity-major (fewer, fuller workgroups).
    int window = 0;
    if (total_pairs >= 262144) window = 64;
    if (const char *e = getenv("AAR_SCHUR_WINDOW")) window = std::max(0, atoi(e));
    if (window > 0) {
        std::vector<int> cursor(A, 0);
        for (int f0 = 0; f0 < F; f0 += window)
            for (int a = 0; a < A; a++) {
                int c = cursor[a];
                const int cnt = (int)inc[a].size();
                int e = c;
                while (e < cnt && inc[a][e].first < f0 + window) e++;
                if (e > c) {
                    sw_ent.push_back(a);
                    sw_begin.push_back(pair_base[a] + c);
                    sw_end.push_back(pair_base[a] + e);
                }
                cursor[a] = e;
            }
    } else {
        // XCD-aware order (opt-in, AAR_SCHUR_XCD=1).  Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: with entity-major items every
        // XCD ends up walking ALL frames and pulls the whole of W through its L2 (8 x 3.6 MB per launch at config 3 for 0.67 MB of output).  Here the frames
        // are cut into 8 contiguous ranges instead, an item stays inside one range, and the items of range x sit at positions = x (mod 8): an XCD then
        // only ever touches an eighth of W (holes are padded with empty items; the launch's rider shifts every range by the same XCD).  Measured (round 4,
        // profiles/r04_attempts.txt): k_schur 20.9 -> 20.3 us at config 3, 51.7 -> 59.4 us at config 4 (more, shorter items; every one ends with an atomic
        // flush of its row panel): the kernel is not bound by those fetches -- not the default.
        int xcd = 1;
        if (const char *e = getenv("AAR_SCHUR_XCD")) xcd = atoi(e) != 0 ? 8 : 1;
        std::vector<std::vector<std::array<int32_t, 3>>> lists(xcd);
        for (int a = 0; a < A; a++) {
            const int cnt = (int)inc[a].size();
            int s = 0;
            for (int x = 0; x < xcd; x++) {
                const int f_hi = (int)((int64_t)F * (x + 1) / xcd);
                int e = s;
                while (e < cnt && inc[a][e].first < f_hi) e++;
                for (int c = s; c < e; c += per_item)
                    lists[x].push_back({(int32_t)a, (int32_t)(pair_base[a] + c), (int32_t)(pair_base[a] + std::min(e, c + per_item))});
                s = e;
            }
        }
        size_t longest = 0;
        for (auto &l : lists) longest = std::max(longest, l.size());
        for (size_t k = 0; k < longest; k++)
            for (int x = 0; x < xcd; x++) {
                if (k < lists[x].size()) { sw_ent.push_back(lists[x][k][0]); sw_begin.push_back(lists[x][k][1]); sw_end.push_back(lists[x][k][2]); }
                else if (xcd > 1) { sw_ent.push_back(0); sw_begin.push_back(0); sw_end.push_back(0); }   // a hole: nothing to walk, an empty panel to flush
            }
    }
    P.n_swork = (int)sw_ent.size();
    // deterministic mode: a record per Schur work item, the items of every entity in ascending frame order; a record per
    // pass-B chunk, the chunks of every camera / marker / (camera, marker) pair in ascending order
    std::vector<int64_t> sp_off;
    std::vector<int32_t> se_start, se_items, pbr_start, pbr_chunk, pbr_kind, pbr_a, pbr_b;
    int64_t sp_total = 0;
    if (P.deterministic) {
        sp_off.resize(sw_ent.size());
        std::vector<std::vector<int32_t>> items(A);
        for (size_t w = 0; w < sw_ent.size(); w++) {
            sp_off[w] = sp_total;
            sp_total += ((int64_t)(sw_ent[w] + 1) * 36 + 8);
            items[sw_ent[w]].push_back((int32_t)w);
        }
        se_start.assign(A + 1, 0);
        for (int a = 0; a < A; a++) {
            // frame-ascending: an entity's pairs are frame-ascending in pair_rec, so ordering its items by their first pair does it
            std::sort(items[a].begin(), items[a].end(), [&](int32_t x, int32_t y) { return sw_begin[x] < sw_begin[y]; });
            se_start[a] = (int32_t)se_items.size();
            se_items.insert(se_items.end(), items[a].begin(), items[a].end());
        }
        se_start[A] = (int32_t)se_items.size();
        std::vector<std::vector<int32_t>> by_cam(C), by_mk(A);
        pbr_start.push_back(0);
        auto emit = [&](int kind, int ea, int eb, const std::vector<int32_t> &chs) {
            if (chs.empty()) return;
            pbr_chunk.insert(pbr_chunk.end(), chs.begin(), chs.end());
            pbr_start.push_back((int32_t)pbr_chunk.size());
            pbr_kind.push_back(kind); pbr_a.push_back(ea); pbr_b.push_back(eb);
        };
        std::vector<int32_t> run;
        for (int ch = 0; ch < P.n_chunks; ch++) {
            const ObsIdx &h = b_idx[chunk_start[ch]];
            by_cam[h.cam].push_back(ch);
            by_mk[h.marker].push_back(ch);
            run.push_back(ch);
            const bool last = ch + 1 == P.n_chunks || b_idx[chunk_start[ch + 1]].cam != h.cam || b_idx[chunk_start[ch + 1]].marker != h.marker;
            if (last) { emit(2, h.cam, h.marker, run); run.clear(); }
        }
        for (int c = 0; c < C; c++) emit(0, c, 0, by_cam[c]);
        for (int m = 0; m < A; m++) emit(1, m, 0, by_mk[m]);
        P.n_pbr = (int)pbr_kind.size();
        P.pb_stride = L.oi ? 152 : 90;
    }
    // many shared entities: the MFMA kernel owns 16 x 32-entity blocks of S and streams frame ranges (solve_kernels.hip);
    // frame ranges are cut so that the grid is a few workgroups per CU
    std::vector<int32_t> sm_ga, sm_gb, sm_fb, sm_fe, slot_frame, slot_dense, dense_ent;
    std::vector<int32_t> sm_frames;   // frame lists of the blocks, back to back
    if (schur_mfma && !P.use_pcg) {
        // dense entity 0 = the pseudo entity g_f; then the entities that are seen at all, MOST FREQUENT FIRST (ties: ascending):
        // the often-seen entities form dense groups, the rarely seen ones share groups that whole stretches of frames do not
        // touch at all -- a block of S then only streams the frames in which both of its entity groups are present
        std::vector<int32_t> seen;
        for (int a = 0; a < A; a++)
            if (!inc[a].empty()) seen.push_back(a);
        std::stable_sort(seen.begin(), seen.end(), [&](int32_t x, int32_t y) { return inc[x].size() > inc[y].size(); });
        std::vector<int32_t> dense_of(A, -1);
        dense_ent.push_back(-1);
        for (int32_t a : seen) { dense_of[a] = (int32_t)dense_ent.size(); dense_ent.push_back(a); }
        P.Ad = ((int)dense_ent.size() + 31) / 32 * 32;
        dense_ent.resize(P.Ad, -1);
        const int nga = P.Ad / 16, ngb = P.Ad / 32;
        // frames in which a group of 16 (a side) / 32 (b side) dense entities has anybody present; the pseudo entity is everywhere
        std::vector<std::vector<uint8_t>> pa(nga, std::vector<uint8_t>(F, 0)), pb(ngb, std::vector<uint8_t>(F, 0));
        for (int f = 0; f < F; f++) {
            pa[0][f] = pb[0][f] = 1;
            for (int s = fslot_start[f]; s < fslot_start[f + 1]; s++) {
                const int dd = dense_of[fslot_ent[s]];
                pa[dd / 16][f] = 1;
                pb[dd / 32][f] = 1;
            }
        }
        struct Blk { int ga, gb; std::vector<int32_t> fr; };
        std::vector<Blk> blks;
        int64_t total_steps = 0;
        for (int ga = 0; ga < nga; ga++)
            for (int gb = 0; gb <= (16 * ga + 15) / 32; gb++) {
                Blk bk{ga, gb, {}};
                for (int f = 0; f < F; f++)
                    if (pa[ga][f] && pb[gb][f]) bk.fr.push_back(f);
                if (bk.fr.empty()) continue;
                total_steps += ((int64_t)bk.fr.size() + 1) / 2;
                blks.push_back(std::move(bk));
            }
        // cut every block's frame list into pieces of about total / target frames (even lengths: two frames per step), pieces of
        // the same stretch of frames next to each other so that the workgroups in flight share panels in L2
        int target = 12 * (int)blks.size();   // pieces per block: measured optimum at config 5 (8: -1 %, 24: -9 %); AAR_SCHUR_SPLIT overrides
        if (const char *e = getenv("AAR_SCHUR_SPLIT")) target = std::max(1, atoi(e)) * (int)blks.size();
        const int64_t plen = std::max<int64_t>(16, ((2 * total_steps + target - 1) / target + 3) / 4 * 4);   // whole steps of the kernel's K loop (2 or 4 frames)
        struct Piece { int blk; int64_t b, e; int f0; };
        std::vector<Piece> pieces;
        for (size_t k = 0; k < blks.size(); k++)
            for (int64_t b0 = 0; b0 < (int64_t)blks[k].fr.size(); b0 += plen)
                pieces.push_back({(int)k, b0, std::min<int64_t>(b0 + plen, (int64_t)blks[k].fr.size()), blks[k].fr[b0]});
        std::stable_sort(pieces.begin(), pieces.end(), [](const Piece &x, const Piece &y) { return x.f0 < y.f0; });
        for (const Piece &pc : pieces) {
            sm_ga.push_back(blks[pc.blk].ga); sm_gb.push_back(blks[pc.blk].gb);
            sm_fb.push_back((int32_t)sm_frames.size());
            sm_frames.insert(sm_frames.end(), blks[pc.blk].fr.begin() + pc.b, blks[pc.blk].fr.begin() + pc.e);
            sm_fe.push_back((int32_t)sm_frames.size());
        }
        slot_frame.resize(P.total_slots);
        slot_dense.resize(P.total_slots);
        for (int f = 0; f < F; f++)
            for (int s = fslot_start[f]; s < fslot_start[f + 1]; s++) {
                slot_frame[s] = f;
                slot_dense[s] = dense_of[fslot_ent[s]];
            }
        P.n_smwork = (int)sm_ga.size();
    }

    std::vector<int32_t> ent_fixed(A, 0);
    for (int c = 0; c < C; c++) ent_fixed[c] = (c == L.rc || !L.oc) ? 1 : 0;
    for (int m = 0; m < M; m++) ent_fixed[C + m] = (m == L.rm || !L.om) ? 1 : 0;
    // (intrinsics entities are free, root camera included: fill_io_vec_cam_intrinsics covers ALL cameras, libs/multicam_mapper.cpp:488-498;
    //  their two idle parameters have zero rows and columns and get mu on the diagonal, like the reference's five distortion columns)
    std::vector<double> Kh(d->cam_mats, d->cam_mats + 9 * (size_t)C);

    // ---- upload ----
#define UP(field, vec) if ((rc = dev_upload(pb, &P.field, vec))) return fail(rc)
    UP(K, Kh); UP(a_idx, a_idx); UP(a_uv, a_uv); UP(frame_obs_start, frame_obs_start); UP(frame_stride, frame_stride);
    UP(fslot_start, fslot_start); UP(fslot_ent, fslot_ent); UP(b_idx, b_idx); UP(b_uv, b_uv);
    UP(chunk_start, chunk_start); UP(ent_fixed, ent_fixed); UP(up_start, up_start); UP(up_ent, up_ent);
    UP(sw_ent, sw_ent); UP(sw_begin, sw_begin); UP(sw_end, sw_end); UP(pair_rec, pair_rec);
    if (P.use_pcg) {   // balanced work items: at most `chunk` incidences of one entity each, 1 .. 8 items per entity (PCG_MAX_ITEMS)
        hipDeviceProp_t prop;
        const int cus = (hipGetDeviceProperties(&prop, pb->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 64;
        int64_t longest = 0;
        for (int a = 0; a < A; a++) longest = std::max<int64_t>(longest, pair_base[a + 1] - pair_base[a]);
        int64_t chunk = std::max<int64_t>(256, (total_pairs + 2LL * cus - 1) / (2LL * cus));
        chunk = std::max<int64_t>(chunk, (longest + 7) / 8);
        if (const char *t = getenv("AAR_PCG_CHUNK")) chunk = std::max<int64_t>((longest + 7) / 8, atoll(t));
        std::vector<int32_t> it_ent, it_begin, it_end, ent_item_start(A + 1, 0);
        for (int a = 0; a < A; a++) {
            ent_item_start[a] = (int32_t)it_ent.size();
            const int64_t b0 = pair_base[a], e0 = ent_fixed[a] ? pair_base[a] : pair_base[a + 1];
            int64_t b = b0;
            do {
                const int64_t e = std::min<int64_t>(e0, b + chunk);
                it_ent.push_back(a); it_begin.push_back((int32_t)b); it_end.push_back((int32_t)e);
                b = e;
            } while (b < e0);
        }
        ent_item_start[A] = (int32_t)it_ent.size();
        P.pcg_n_items = (int)it_ent.size();
        UP(pcg_it_ent, it_ent); UP(pcg_it_begin, it_begin); UP(pcg_it_end, it_end); UP(pcg_ent_item_start, ent_item_start);
    }
    if (P.deterministic) { UP(sp_off, sp_off); UP(se_start, se_start); UP(se_items, se_items); UP(pbr_start, pbr_start); UP(pbr_chunk, pbr_chunk); UP(pbr_kind, pbr_kind); UP(pbr_a, pbr_a); UP(pbr_b, pbr_b); }
    if (P.n_smwork) { UP(sm_ga, sm_ga); UP(sm_gb, sm_gb); UP(sm_fb, sm_fb); UP(sm_fe, sm_fe); UP(slot_dense, slot_dense); UP(slot_frame, slot_frame); UP(dense_ent, dense_ent); UP(sm_frames, sm_frames); }
#undef UP
#define AL(field, count) if ((rc = dev_alloc(pb, &P.field, (size_t)(count)))) return fail(rc)
    AL(z[0], 6 * (size_t)(A + F)); AL(z[1], 6 * (size_t)(A + F));
    AL(ent[0], (size_t)(A + F) * ENT_STRIDE); AL(ent[1], (size_t)(A + F) * ENT_STRIDE);
    for (int w = 0; w < 2; w++) {
        AL(blk[w].V, (size_t)F * 36); AL(blk[w].gf, (size_t)F * 6); AL(blk[w].W, (size_t)P.total_slots * 36);
        AL(blk[w].Vinv, (size_t)F * 36); AL(blk[w].hf, (size_t)F * 6);
        // rhs and g0 live right behind S: ONE all-reduce makes all three global on the multi-GPU path (g0, the shared part of
        // B = -J^T r, is added to the right-hand side on first touch and enters delta.B, so every rank needs all of it)
        AL(blk[w].S, (size_t)P.n_pad * P.n_pad + 2 * (size_t)P.n_pad + 8);
        P.blk[w].rhs = P.blk[w].S + (size_t)P.n_pad * P.n_pad;
        P.blk[w].g0 = P.blk[w].rhs + P.n_pad;
        P.blk[w].tail = P.blk[w].g0 + P.n_pad;
        if (hipMemset(P.blk[w].tail, 0, 8 * sizeof(double)) != hipSuccess) return fail(set_error(AAR_ERR_HIP, "hipMemset failed"));
    }
    if (P.deterministic) { AL(sp_part, (size_t)sp_total); AL(pb_part, (size_t)P.n_chunks * P.pb_stride); }
    if (P.use_pcg) {
        AL(pcg_ws, (size_t)P.pcg_n_items * 28 + 6 * (size_t)F + 8); AL(pcg_counter, 8);
        AL(pcg_yg, (size_t)3 * PCG_NYV * P.n_pad + (size_t)28 * A + 160 + 8); AL(pcg_hop, 2 * PCG_HOP_WORDS);   // (+ 160: the coarse operator's accumulator)
        if (const char *t = getenv("AAR_PCG_FUSED")) P.pcg_fused = atoi(t) != 0 ? 1 : 0;
        {   // k_pcgf's operator reads an fp32 copy of W (half the bytes of its pass over the frames; written by pass A instead of the fp64 blocks): non-deterministic runs, one rank or many
            int w32 = 1;
            if (const char *t = getenv("AAR_PCG_W32")) w32 = atoi(t) != 0 ? 1 : 0;
            if (w32 && P.pcg_fused && !P.deterministic && P.pcg_eta >= PCG_W32_MIN_ETA)   // (a caller who asks for residuals below 1e-4 gets fp64 blocks throughout)
                for (int w = 0; w < 2; w++) AL(blk[w].Wf, (size_t)P.total_slots * 36 + 4);
        }
        hipDeviceProp_t prop;
        P.pcg_grid = (hipGetDeviceProperties(&prop, pb->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 64;   // one workgroup per CU: all resident
        // small problems: fewer workgroups make the two grid-wide hand-overs of an iteration cheaper than the passes get slower
        // (measured, LM it/s at config 3: 64 / 128 / 256 workgroups 4174 / 4201 / 3883; config 2: 32 best; config 5: 256 best)
        const int64_t want = std::max<int64_t>(32, ((int64_t)F + 3) / 4);
        P.pcg_grid = (int)std::min<int64_t>(P.pcg_grid, want);
        if (pb->comm) {
            AL(pcgd_setup, (size_t)A * 28 + 152); AL(pcgd_minv, (size_t)A * 36 + 72 + (size_t)12 * A + 8);   // (+ the coarse operator's shares; + its inverse blocks and the table of Z)
            AL(pcgd_state, 2 * (18 * (size_t)A + 8)); AL(pcgd_y, 3 * (6 * (size_t)A + 8));
            if (hipHostMalloc((void **)&pb->h_pcg, 8 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
                return fail(set_error(AAR_ERR_HIP, "hipHostMalloc failed"));
            memset(pb->h_pcg, 0, 8 * sizeof(double));
            if (hipHostGetDevicePointer((void **)&P.pcgd_host, pb->h_pcg, 0) != hipSuccess) return fail(set_error(AAR_ERR_HIP, "hipHostGetDevicePointer failed"));
        }
        if (const char *t = getenv("AAR_PCG_GRID")) P.pcg_grid = std::max(1, atoi(t));
        // the kernels' grid-wide hand-overs need every workgroup resident: never more than the occupancy query admits (the override included)
        {
            hipDeviceProp_t prop2;
            const int cus2 = (hipGetDeviceProperties(&prop2, pb->device) == hipSuccess && prop2.multiProcessorCount > 0) ? prop2.multiProcessorCount : 64;
            P.pcg_grid = std::min(P.pcg_grid, pcg_max_grid(A, cus2));
        }
    }
    if (P.use_spcg) {
        AL(spcg_ws, spcg_ws_doubles(P.n_pad)); AL(spcg_iters, 8); AL(spcg_done, 2);
        if (spcg_coarse_now(P)) AL(spcg_pre, spcg_pre_doubles(P.n_pad));   // (zeroed: the restriction rows' columns of fixed entities stay zero)
        spcg_ws_reset(P, pb->stream);
    }
    if (P.n_smwork) { AL(Wd, (size_t)F * P.Ad * 36); AL(Yd, (size_t)F * P.Ad * 36); }   // zeroed here, once: absent pairs are never written
    AL(Dfac, (size_t)P.nT * CHOL_NB * CHOL_NB); AL(Linv16, (size_t)P.nT * (CHOL_NB / 16) * 256); AL(delta_s, P.n_pad); AL(bs_flags, (size_t)P.nT + 1);
    AL(Lp, (size_t)P.nT * P.n_pad * CHOL_NB); AL(zf, P.n_pad);
    AL(err_part, std::max<size_t>((size_t)F, (size_t)((N + 255) / 256)) + 1);
    AL(lin_part, 2 * (size_t)(F + 1)); AL(scal, 8); AL(flags, 4);
#undef AL
    if ((rc = dev_alloc(pb, &pb->d_diag, P.n_pad))) return fail(rc);
    if (pb->comm && (rc = dev_alloc(pb, &pb->d_frames_all, 6 * (size_t)std::max(Fg, 1)))) return fail(rc);
    if (pb->comm && (rc = dev_alloc(pb, &pb->d_pack, (size_t)P.n_pad * (P.n_pad + 1) / 2 + 2 * (size_t)P.n_pad + 8))) return fail(rc);
    if (hipHostMalloc((void **)&pb->h_scal, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
        return fail(set_error(AAR_ERR_HIP, "hipHostMalloc failed"));
    memset(pb->h_scal, 0, 16 * sizeof(double));
    if (hipHostGetDevicePointer((void **)&P.host_result, pb->h_scal, 0) != hipSuccess)
        return fail(set_error(AAR_ERR_HIP, "hipHostGetDevicePointer failed"));
    if (hipStreamSynchronize(pb->stream) != hipSuccess) return fail(set_error(AAR_ERR_HIP, "upload failed"));
    {   // the runtime brings its device-to-host copy path up on first use (8-13 ms, once per process: scripts/probe/outlier2.py):
        // better here than inside the caller's first solve
        if (pb->h_z.reserve((size_t)6 * (A + F))) return fail(set_error(AAR_ERR_HIP, "hipHostMalloc(pose staging) failed"));
        if (hipEventCreateWithFlags(&pb->up_ev, hipEventDisableTiming) != hipSuccess) { pb->up_ev = nullptr; (void)hipGetLastError(); }
        if (hipMemcpyAsync(pb->h_z.data(), P.z[0], (size_t)6 * (A + F) * sizeof(double), hipMemcpyDeviceToHost, pb->stream) != hipSuccess ||
            hipStreamSynchronize(pb->stream) != hipSuccess)
            return fail(set_error(AAR_ERR_HIP, "device-to-host copy failed"));
    }
    pb->h_fslot_start = fslot_start;
    pb->h_fslot_ent = fslot_ent;
    *out = pb;
    return AAR_OK;
}

int64_t aar_problem_full_len(const aar_problem *pb) { return pb->L.full_len(); }
int64_t aar_problem_num_vars(const aar_problem *pb) { return pb->L.z_len(); }
int64_t aar_problem_local_obs(const aar_problem *pb) { return pb->P.N; }

int aar_problem_set_huber_delta(aar_problem *pb, float delta) {
    if (!pb || !(delta > 0.f)) return set_error(AAR_ERR_INVALID, "aar_problem_set_huber_delta: bad argument");
    pb->hubber_delta = delta;
    if (pb->with_huber) pb->P.huber = delta;
    return AAR_OK;
}
float aar_problem_get_huber_delta(const aar_problem *pb) { return pb ? pb->hubber_delta : 0.f; }

int aar_eval_residuals(aar_problem *pb, const double *x_full, double *r, double *sum_sq) {
    if (!pb || !x_full) return set_error(AAR_ERR_INVALID, "aar_eval_residuals: null argument");
    if (r && pb->comm) return set_error(AAR_ERR_UNSUPPORTED, "residual vector output is single-GPU only");
    HIP_TRY(hipSetDevice(pb->device));
    DeviceProblem &P = pb->P;
    int rc = upload_z(pb, x_full, pb->cur);
    if (rc) return rc;
    double *d_r = nullptr;
    if (r) HIP_TRY(hipMalloc((void **)&d_r, std::max<size_t>(1, 8 * (size_t)P.N) * sizeof(double)));
    launch_unpack(P, pb->cur, pb->stream);
    launch_residual(P, pb->cur, d_r, pb->stream);
    HIP_TRY(hipMemsetAsync(P.lin_part, 0, 2 * (size_t)(P.F + 1) * sizeof(double), pb->stream));
    launch_reduce_scalars(P, residual_blocks(P), false, 0ull, pb->stream);
    rc = allreduce(pb, P.scal, 1, NCCL_SUM);
    if (rc) { if (d_r) (void)hipFree(d_r); return rc; }
    pb->seq++;
    launch_publish(P, pb->seq, pb->stream);
    HIP_TRY(hipStreamSynchronize(pb->stream));
    if (r && (rc = copy_d2h(pb, r, d_r, 8 * (size_t)P.N * sizeof(double)))) { (void)hipFree(d_r); return rc; }
    if ((rc = wait_result(pb))) { if (d_r) (void)hipFree(d_r); return rc; }
    if (d_r) (void)hipFree(d_r);
    if (sum_sq) *sum_sq = pb->h_scal[0];
    pb->lm_ready = false;
    return check_async("residual kernels");
}

int aar_reproj_stats(aar_problem *pb, const double *x_full, double *rmse, double *sum_sq) {
    if (!pb) return set_error(AAR_ERR_INVALID, "aar_reproj_stats: null argument");
    const int saved = pb->P.res_f32;
    const float saved_h = pb->P.huber;
    pb->P.res_f32 = 0;
    pb->P.huber = -1.f;
    double ss = 0;
    int rc = aar_eval_residuals(pb, x_full, nullptr, &ss);
    pb->P.res_f32 = saved;
    pb->P.huber = saved_h;
    if (rc) return rc;
    if (sum_sq) *sum_sq = ss;
    if (rmse) *rmse = std::sqrt(ss / (4.0 * (double)pb->N_global));
    return AAR_OK;
}

int aar_eval_normal_equations(aar_problem *pb, const double *x_full, double *JtJ, double *B, double *sum_sq) {
    if (!pb || !x_full) return set_error(AAR_ERR_INVALID, "aar_eval_normal_equations: null argument");
    if (pb->comm) return set_error(AAR_ERR_UNSUPPORTED, "dense normal-equation output is single-GPU only");
    HIP_TRY(hipSetDevice(pb->device));
    DeviceProblem &P = pb->P;
    const PoseLayout &L = pb->L;
    int rc = upload_z(pb, x_full, pb->cur);
    if (rc) return rc;
    if ((rc = zero_block_set(pb, pb->cur))) return rc;
    P.want_w64 = 1;
    rc = eval_blocks(pb, pb->cur, -1.0, -1);
    P.want_w64 = 0;
    if (rc) return rc;
    const int A = P.A, F = P.F, np = P.n_pad;
    const DeviceProblem::Blocks &bk = P.blk[pb->cur];
    std::vector<double> U0((size_t)np * np), g0(np), V((size_t)F * 36), gf((size_t)F * 6), W((size_t)P.total_slots * 36), ep(F);
    // (copies through the library's page-locked staging, hostcopy.h)
    HIP_TRY(hipStreamSynchronize(pb->stream));
    if ((rc = copy_d2h(pb, U0.data(), bk.S, U0.size() * sizeof(double))) || (rc = copy_d2h(pb, g0.data(), bk.g0, g0.size() * sizeof(double)))) return rc;
    if (F) {
        if ((rc = copy_d2h(pb, V.data(), bk.V, V.size() * sizeof(double))) || (rc = copy_d2h(pb, gf.data(), bk.gf, gf.size() * sizeof(double))) ||
            (rc = copy_d2h(pb, W.data(), bk.W, W.size() * sizeof(double))) || (rc = copy_d2h(pb, ep.data(), P.err_part, ep.size() * sizeof(double)))) return rc;
    }
    pb->lm_ready = false;
    // reference column of each device parameter (or -1): roots, non-optimised groups and the two idle parameters of an
    // intrinsics entity have none
    const int64_t Pz = L.z_len();
    auto par_col = [&](int a, int i) -> int64_t {
        if (a < L.C) { const int s = L.cam_slot(a); return (s < 0 || !L.oc) ? -1 : L.z_cam0() + 6LL * s + i; }
        if (a < L.C + L.M) { const int s = L.mk_slot(a - L.C); return (s < 0 || !L.om) ? -1 : L.z_mk0() + 6LL * s + i; }
        return i < 4 ? L.z_intr0() + 9LL * (a - L.C - L.M) + i : -1;   // fx cx fy cy; the d0..d4 columns stay zero
    };
    if (JtJ) {
        std::fill(JtJ, JtJ + Pz * Pz, 0.0);
        for (int a = 0; a < A; a++)
            for (int b = 0; b <= a; b++)
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) {
                        if (a == b && j > i) continue;
                        const int64_t ca = par_col(a, i), cb = par_col(b, j);
                        if (ca < 0 || cb < 0) continue;
                        const double v = U0[(size_t)(6 * a + i) * np + 6 * b + j];
                        JtJ[ca * Pz + cb] = v;
                        JtJ[cb * Pz + ca] = v;
                    }
        if (L.of)
            for (int f = 0; f < F; f++) {
                const int64_t cf = L.z_fr0() + 6LL * f;
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) JtJ[(cf + i) * Pz + cf + j] = V[(size_t)f * 36 + i * 6 + j];
                for (int s = pb->h_fslot_start[f]; s < pb->h_fslot_start[f + 1]; s++)
                    for (int i = 0; i < 6; i++) {
                        const int64_t ca = par_col(pb->h_fslot_ent[s], i);
                        if (ca < 0) continue;
                        for (int j = 0; j < 6; j++) {
                            const double v = W[(size_t)s * 36 + i * 6 + j];
                            JtJ[ca * Pz + cf + j] = v;
                            JtJ[(cf + j) * Pz + ca] = v;
                        }
                    }
            }
    }
    if (B) {
        std::fill(B, B + Pz, 0.0);
        for (int a = 0; a < A; a++)
            for (int i = 0; i < 6; i++) {
                const int64_t ca = par_col(a, i);
                if (ca >= 0) B[ca] = g0[6 * (size_t)a + i];
            }
        if (L.of)
            for (int f = 0; f < F; f++)
                for (int i = 0; i < 6; i++) B[L.z_fr0() + 6LL * f + i] = gf[(size_t)f * 6 + i];
    }
    if (sum_sq) {
        double s = 0;
        for (int f = 0; f < F; f++) s += ep[f];
        *sum_sq = s;
    }
    return AAR_OK;
}

int aar_eval_damped_step(aar_problem *pb, const double *x_full, double mu, double *delta) {
    if (!pb || !x_full || !delta) return set_error(AAR_ERR_INVALID, "aar_eval_damped_step: null argument");
    HIP_TRY(hipSetDevice(pb->device));
    const PoseLayout &L = pb->L;
    int rc = upload_z(pb, x_full, pb->cur);
    if (rc) return rc;
    if ((rc = zero_block_set(pb, pb->cur))) return rc;
    if ((rc = eval_blocks(pb, pb->cur, -1.0, -1))) return rc;
    pb->vinv_mu = -1;
    pb->schur_mu = -1;
    pb->s_reduced = pb->trial_reduced = false;
    pb->spcg_skip = pb->spcg_backoff = 0;   // (a one-off solve: the problem's own solver gets its chance whatever an earlier LM run ended with)
    pb->last_cg_its = -1;
    pb->near_cap_tries = 0;
    pb->P.spcg_coarse_on = 0;
    pb->P.pcg_e_age = 0;
    if (pb->P.use_pcg && pb->P.pcg_counter) HIP_TRY(hipMemsetAsync(pb->P.pcg_counter + 2, 0, sizeof(int32_t), pb->stream));   // (k_pcgf's coarse space joins by the previous solve's count)
    pb->P.pcg_eta_now = pb->P.pcg_eta;      // (... and the forcing term itself, not the LM's forcing sequence)
    if ((rc = damped_try_fb(pb, mu, false))) return rc == TRY_NOT_POSITIVE_DEFINITE ? AAR_ERR_NUMERIC : rc;
    pb->lm_ready = false;
    std::vector<double> x0(x_full, x_full + L.full_len()), x1(x0);
    if ((rc = download_z(pb, 1 - pb->cur, x1.data()))) return rc;
    // z ordering of the Config
    std::vector<double> z0((size_t)std::max<int64_t>(L.z_len(), 1)), z1(z0);
    extract_z(L, x0.data(), z0.data());
    extract_z(L, x1.data(), z1.data());
    for (int64_t i = 0; i < L.z_len(); i++) delta[i] = z1[i] - z0[i];
    return AAR_OK;
}

// SparseLevMarq::init, libs/sparselevmarq.h:238-249.  The residual of the start point comes out of the same pass that
// builds its normal equations (the first step() needs them anyway).
int aar_lm_init(aar_problem *pb, const double *x_full, const aar_lm_params *prm) {
    if (!pb || !x_full) return set_error(AAR_ERR_INVALID, "aar_lm_init: null argument");
    HIP_TRY(hipSetDevice(pb->device));
    if (prm) pb->prm = *prm;
    DeviceProblem &P = pb->P;
    pb->cur = 0;
    pb->trial_points = 0;
    pb->launches = 0;
    pb->spcg_skip = pb->spcg_backoff = 0;
    pb->last_cg_its = -1;
    pb->near_cap_tries = 0;
    P.spcg_coarse_on = 0;
    P.pcg_e_age = 0;
    if (P.use_pcg && P.pcg_counter) HIP_TRY(hipMemsetAsync(P.pcg_counter + 2, 0, sizeof(int32_t), pb->stream));   // (k_pcgf's coarse space joins by the previous solve's count)
    if (pb->spec_chol_blk >= 0) {   // a factorisation queued ahead of the last step of the previous solve: its pivot flags mean nothing
        HIP_TRY(hipMemsetAsync(P.flags, 0, 4 * sizeof(int32_t), pb->stream));
        pb->spec_chol_blk = -1;
    }
    memset(&pb->times, 0, sizeof pb->times);
    int rc = upload_z(pb, x_full, 0);
    if (rc) return rc;
    // A solve's fixed cost (every 15 steps or so in a bundle adjustment that converges): the first launch turns z into entity rows AND
    // clears block set 0 and the linear-model partials, pass A clears block set 1 on its way, and max diag(J^T J) of the start point
    // (mu_0, libs/sparselevmarq.h:369-377) comes out of the launch that reduces its sum r^2 -- three launches less than one kernel per job
    if (P.F > 0 && P.n_chunks > 0 && !pb->comm) {
        launch_unpack(P, 0, pb->stream, /*zero_blk=*/0);
        pb->launches += 1;
        if ((rc = eval_blocks(pb, 0, -1.0, /*zero_blk=*/1, /*spec_schur=*/false, /*ents_ready=*/true))) return rc;
    } else {
        if ((rc = zero_for_init(pb))) return rc;
        if ((rc = eval_blocks(pb, 0, -1.0, -1))) return rc;
    }
    pb->huber_of_blocks = P.huber;
    pb->mu_seed_valid = false;
    // (single GPU: mu_0 rides to the host with the sum r^2: the first step() then needs no round trip of its own)
    if ((rc = launch_scalars(pb, P.F, pb->comm ? -1 : 0))) return rc;
    // Head start of the first step (single GPU, direct solver): its damping mu_0 = tau * max diag(J^T J) is on the device one kernel before the host
    // can read it, and the frame inverses and the Schur complement need nothing else -- they are queued HERE, with mu_0 read on the device, and
    // run while the record travels to the host and the host queues the factorisation.  (tau changed between init and step: the damped try finds
    // another damping than schur_mu and takes the complement back, as after a mispredicted step.)  AAR_INIT_HEADSTART=0: off.
    const bool headstart = P.tune.init_headstart && !pb->comm && !P.use_pcg && !pb->stage_timers && !pb->profiling && P.F > 0;
    if (headstart) {
        launch_frame_inv(P, 0, 0.0, pb->stream, P.scal + 4, pb->prm.tau);
        launch_schur(P, 0, 1.0, pb->stream, 0, 0, nullptr, false);
        pb->launches += 2;
    }
    if ((rc = wait_result(pb))) return rc;
    if (!pb->comm) { pb->mu_seed = pb->h_scal[4]; pb->mu_seed_valid = true; }
    pb->currErr = pb->prevErr = pb->h_scal[0];
    pb->rel_drop = -1;
    pb->blocks_valid = true;
    pb->vinv_mu = -1;
    pb->schur_mu = -1;
    if (headstart) {
        const double mu0 = pb->mu_seed * pb->prm.tau;   // (the expression aar_lm_step uses: the same double as the device's tau * scal[4])
        pb->vinv_mu = mu0;
        pb->schur_mu = mu0;
        panels_now(pb, 0, mu0);
    }
    pb->s_reduced = pb->trial_reduced = false;
    pb->mu = -1;
    pb->v = 2;  // indeterminate in the reference (libs/sparselevmarq.h:133); every accepted step sets 2 (:411)
    pb->lm_ready = true;
    return AAR_OK;
}

// SparseLevMarq::step(f, J), libs/sparselevmarq.h:349-430
int aar_lm_step(aar_problem *pb, aar_lm_iter *out) {
    if (!pb) return set_error(AAR_ERR_INVALID, "aar_lm_step: null argument");
    if (!pb->lm_ready) return set_error(AAR_ERR_INVALID, "aar_lm_step: call aar_lm_init first");
    HIP_TRY(hipSetDevice(pb->device));
    int rc;
    // verbose: the reference's per-step stage line (:425) needs per-stage device times, i.e. the stage timers for this step
    struct VerboseTimers {
        aar_problem *pb; bool was; aar_stage_times t0;
        explicit VerboseTimers(aar_problem *p) : pb(p), was(p->stage_timers), t0(p->times) { if (pb->prm.verbose) pb->stage_timers = true; }
        ~VerboseTimers() { pb->stage_timers = was; }
    } vt(pb);
    if (!pb->blocks_valid && (rc = rebuild_current(pb))) return rc;  // J, Jt*J, B at curr_z (:353-367)
    if (pb->mu < 0) {                                                  // first time only (:369-377)
        if (pb->mu_seed_valid && pb->cur == 0 && pb->blocks_valid) pb->mu = pb->mu_seed * pb->prm.tau;   // (the blocks of the start point are still the ones it was taken from)
        else if ((rc = initial_mu(pb, pb->prm.tau, &pb->mu))) return rc;
        pb->mu_seed_valid = false;
    }
    double gain = 0, dnorm = 0;
    int ntries = 0;
    bool accepted = false;
    do {
        if (!pb->blocks_valid && (rc = rebuild_current(pb))) return rc;  // a rejected try consumed them
        const double mu_used = pb->mu;
        if (pb->P.use_pcg || pb->P.use_spcg)   // (every rank sees the same errors: the same choice)
            pb->P.pcg_eta_now = (pb->P.pcg_eta_loose > pb->P.pcg_eta && (pb->rel_drop < 0 || pb->rel_drop > pb->P.pcg_eta_switch)) ? pb->P.pcg_eta_loose : pb->P.pcg_eta;
        if ((rc = damped_try_fb(pb, mu_used, true))) {
            if (rc != TRY_NOT_POSITIVE_DEFINITE) return rc;
            // J^T J + mu I came out indefinite in floating point: the reference's LDL^T would hand back the stationary point
            // of an indefinite model and its gain test would reject the trial (libs/sparselevmarq.h:408-419); same outcome here
            pb->mu = mu_used * pb->v;
            pb->v = pb->v * 5;
            pb->trial_reduced = false;
            pb->blocks_valid = false;
            gain = -1;
            continue;
        }
        const double *sc = pb->h_scal;
        const double err = sc[0];
        // L = 0.5 * delta^T (mu*delta - B) (:406); frame pieces were summed over ranks; the shared-parameter pieces
        // |delta_s|^2 and delta_s . g0 are computed from replicated data (g0 was all-reduced with S) and counted once
        const double d2 = sc[1] + sc[5];
        const double dg = (pb->P.use_pcg && pb->comm) ? sc[2] : sc[2] + sc[6];   // (PCG with ranks: the shared piece is inside the rank sum)
        const double Lq = 0.5 * (mu_used * d2 - dg);
        dnorm = std::sqrt(d2);
        gain = (err - pb->prevErr) / Lq;
        if (gain > 0 && ((err - pb->prevErr) < 0)) {  // :409-415
            pb->mu = mu_used * std::max(0.33, 1. - std::pow(2 * gain - 1, 3));
            pb->v = 2.f;
            pb->rel_drop = pb->prevErr > 0 ? (pb->prevErr - err) / pb->prevErr : 0.0;
            pb->currErr = err;
            pb->cur = 1 - pb->cur;  // curr_z = estimated_z; its blocks were built speculatively by the try
            pb->huber_of_blocks = pb->P.huber;
            pb->vinv_mu = mu_used * 0.33;   // what pass A inverted for
            pb->schur_mu = mu_used * 0.33;  // ... and what the speculative Schur complement was taken with
            pb->s_reduced = pb->trial_reduced;   // multi-GPU: ... and whether it has been all-reduced already
            pb->trial_reduced = false;
            pb->blocks_valid = true;        // (a damping other than the predicted one is repaired in damped_try)
            accepted = true;
        } else {
            pb->mu = mu_used * pb->v;
            pb->v = pb->v * 5;
            pb->trial_reduced = false;
        }
    } while (gain <= 0 && ntries++ < 5 && !accepted);
    if (out) {
        out->err = pb->currErr; out->mu = pb->mu; out->gain = gain; out->delta_norm = dnorm;
        out->accepted = accepted ? 1 : 0;
        out->tries = ntries + (accepted ? 1 : 0);
    }
    if (pb->prm.verbose) {
        fprintf(stderr, "Curr Error=%.5g AErr(prev-curr)=%.5g gain=%.5g dumping factor=%.5g\n", pb->currErr,
                (pb->prevErr - pb->currErr) / (8.0 * (double)pb->N_global), gain, pb->mu);
        // the reference's stage names (libs/sparselevmarq.h:425), seconds.  "J" = the fused residual + Jacobian + block accumulation
        // passes (which ARE transpose, Jt*J and B here: nothing is transposed or multiplied afterwards), "Jt*J" = the reduction of
        // those blocks onto the shared parameters (Schur complement), "chol" = dense LDL^T + both back-substitutions
        const aar_stage_times &a = vt.t0, &b = pb->times;
        fprintf(stderr, " J=%.6g transpose=%.6g Jt*J=%.6g B=%.6g chol=%.6g\n", b.jacobian_normal_eq - a.jacobian_normal_eq, 0.0, b.schur - a.schur, 0.0,
                (b.chol - a.chol) + (b.backsub - a.backsub));
    }
    return AAR_OK;
}

int aar_lm_get_solution(aar_problem *pb, double *x_full, double *err) {
    if (!pb || !x_full) return set_error(AAR_ERR_INVALID, "aar_lm_get_solution: null argument");
    if (!pb->lm_ready) return set_error(AAR_ERR_INVALID, "aar_lm_get_solution: no LM state");
    HIP_TRY(hipSetDevice(pb->device));
    if (err) *err = pb->currErr;
    return download_z(pb, pb->cur, x_full);
}

// SparseLevMarq::solve(z, f, J), libs/sparselevmarq.h:440-472
int aar_lm_solve(aar_problem *pb, double *x_full, const aar_lm_params *prm, aar_lm_report *rep) {
    if (!pb || !x_full) return set_error(AAR_ERR_INVALID, "aar_lm_solve: null argument");
    // MultiCamMapper::solve sets hubberDelta = 10 before solver.solve (libs/multicam_mapper.cpp:425) and installs optCallBack as
    // the step callback (:422).  A with_huber problem WITHOUT a caller's step callback gets exactly that pair here; a caller
    // that installs its own callback (aar::MultiCamMapper does) owns both the start value and the schedule.
    const bool own_schedule = pb->with_huber && !pb->step_cb;
    if (own_schedule) {
        pb->hubber_delta = 10;
        pb->P.huber = pb->hubber_delta;
    }
    int rc = aar_lm_init(pb, x_full, prm);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    const double rows = 8.0 * (double)pb->N_global;
    const double initial = pb->currErr;
    int mustExit = 0, iters = 0;
    auto after_step = [&](const aar_lm_iter &it) -> int {   // _step_callback(curr_z), :449 / :463
        if (rep && rep->trace && iters < rep->trace_cap) rep->trace[iters] = it;
        iters++;
        if (pb->step_cb) {
            if (pb->step_want_z && (rc = current_z_for_callbacks(pb, x_full))) return rc;
            pb->step_cb(pb->step_ctx, pb->step_want_z ? pb->cb_z.data() : nullptr, pb->L.z_len());
        } else if (own_schedule && pb->hubber_delta > 2.5) {  // optCallBack, libs/multicam_mapper.cpp:412-417 (float -= double)
            pb->hubber_delta = (float)((double)pb->hubber_delta - 7.5 / 500);
            pb->P.huber = pb->hubber_delta;
        }
        return AAR_OK;
    };
    if (pb->stop_fn) {   // :444-450: do { step; callback } while (!stop(curr_z)) -- no iteration cap, no error-based exit
        bool stop = false;
        do {
            aar_lm_iter it;
            if ((rc = aar_lm_step(pb, &it))) return rc;
            if ((rc = after_step(it))) return rc;
            // (prevErr is NOT advanced on this branch of the reference: every gain test keeps comparing with the error of the
            //  start point, libs/sparselevmarq.h:444-450 against :464)
            if ((rc = current_z_for_callbacks(pb, x_full))) return rc;
            stop = pb->stop_fn(pb->stop_ctx, pb->cb_z.data(), pb->L.z_len()) != 0;
        } while (!stop);
    } else {
        for (int i = 0; i < pb->prm.max_iters && !mustExit; i++) {
            aar_lm_iter it;
            if ((rc = aar_lm_step(pb, &it))) return rc;
            if (pb->currErr < pb->prm.min_error) mustExit = 1;
            if (std::fabs(pb->prevErr - pb->currErr) <= pb->prm.min_step_error_diff ||
                std::fabs((pb->prevErr - pb->currErr) / rows) <= pb->prm.min_average_step_error_diff || !it.accepted)
                mustExit = 2;
            if (pb->currErr > pb->prevErr) mustExit = 3;
            if ((rc = after_step(it))) return rc;
            pb->prevErr = pb->currErr;
        }
    }
    // (the last step's speculative work may still be running: only waited for when somebody reads timers)
    const bool lazy = !pb->comm && !pb->profiling && !pb->stage_timers && iters > 0;
    if (!lazy) HIP_TRY(hipStreamSynchronize(pb->stream));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pb->times.total = secs;
    pb->times.launches = pb->launches;
    if ((rc = download_z(pb, pb->cur, x_full, lazy))) return rc;
    if (rep) {
        rep->iterations = iters;
        rep->stop_code = mustExit;
        rep->initial_err = initial;
        rep->final_err = pb->currErr;
        rep->final_mu = pb->mu;
        rep->solve_seconds = secs;
        rep->trial_points = pb->trial_points;
    }
    return AAR_OK;
}

int aar_lm_set_step_callback(aar_problem *pb, aar_lm_step_callback fn, void *ctx, int32_t want_z) {
    if (!pb) return set_error(AAR_ERR_INVALID, "aar_lm_set_step_callback: null argument");
    pb->step_cb = fn;
    pb->step_ctx = ctx;
    pb->step_want_z = fn && want_z;
    return AAR_OK;
}

int aar_lm_set_stop_function(aar_problem *pb, aar_lm_stop_function fn, void *ctx) {
    if (!pb) return set_error(AAR_ERR_INVALID, "aar_lm_set_stop_function: null argument");
    pb->stop_fn = fn;
    pb->stop_ctx = ctx;
    return AAR_OK;
}

int aar_problem_extract_z(const aar_problem *pb, const double *x_full, double *z) {
    if (!pb || !x_full || !z) return set_error(AAR_ERR_INVALID, "aar_problem_extract_z: null argument");
    extract_z(pb->L, x_full, z);
    return AAR_OK;
}

int aar_problem_merge_z(const aar_problem *pb, const double *z, double *x_full) {
    if (!pb || !x_full || !z) return set_error(AAR_ERR_INVALID, "aar_problem_merge_z: null argument");
    merge_z(pb->L, z, x_full);
    return AAR_OK;
}

int aar_track(aar_problem *pb, double *x_full, const aar_lm_params *prm, int32_t *iterations, double *final_err) {
    if (!pb || !x_full) return set_error(AAR_ERR_INVALID, "aar_track: null argument");
    HIP_TRY(hipSetDevice(pb->device));
    DeviceProblem &P = pb->P;
    aar_lm_params p;
    if (prm) p = *prm; else aar_lm_default_params(&p);
    pb->lm_ready = false;
    const int F = P.F;
    int rc = upload_z(pb, x_full, pb->cur);
    if (rc) return rc;
    int32_t *d_it = nullptr;
    double *d_err = nullptr;
    HIP_TRY(hipMalloc((void **)&d_it, std::max(F, 1) * sizeof(int32_t)));
    if (hipMalloc((void **)&d_err, std::max(F, 1) * sizeof(double)) != hipSuccess) { (void)hipFree(d_it); return set_error(AAR_ERR_HIP, "hipMalloc failed"); }
    launch_unpack(P, pb->cur, pb->stream);  // rows of the fixed cameras / markers
    launch_track(P, pb->cur, p.max_iters, p.min_error, p.min_step_error_diff, p.min_average_step_error_diff, p.tau, d_it, d_err, pb->stream);
    std::vector<int32_t> h_it(std::max(F, 1));
    std::vector<double> h_err(std::max(F, 1));
    rc = copy_d2h(pb, h_it.data(), d_it, std::max(F, 1) * sizeof(int32_t));
    if (!rc) rc = copy_d2h(pb, h_err.data(), d_err, std::max(F, 1) * sizeof(double));
    (void)hipFree(d_it);
    (void)hipFree(d_err);
    if (rc) return rc;
    if ((rc = check_async("track kernel"))) return rc;
    // only the frame poses move; download_z honours the Config flags, so force "frames on, shared off" for this call
    PoseLayout keep = pb->L;
    pb->L.oc = false; pb->L.om = false; pb->L.of = true;
    rc = download_z(pb, pb->cur, x_full);
    pb->L = keep;
    if (rc) return rc;
    // per-frame outputs are indexed by global frame; a sharded problem returns its own range and zeros elsewhere
    if (iterations) { std::fill(iterations, iterations + pb->L.F, 0); for (int f = 0; f < F; f++) iterations[pb->f_begin + f] = h_it[f]; }
    if (final_err) { std::fill(final_err, final_err + pb->L.F, 0.0); for (int f = 0; f < F; f++) final_err[pb->f_begin + f] = h_err[f]; }
    return AAR_OK;
}

int aar_set_kernel_profiling(aar_problem *pb, int on) {
    if (!pb) return set_error(AAR_ERR_INVALID, "aar_set_kernel_profiling: null argument");
    HIP_TRY(hipSetDevice(pb->device));
    HIP_TRY(hipStreamSynchronize(pb->stream));
    pb->ev_kid.clear();
    pb->ev_used = 0;
    pb->profiling = on != 0;
    if (on) {
        memset(pb->k_seconds, 0, sizeof pb->k_seconds);
        memset(pb->k_launches, 0, sizeof pb->k_launches);
        pb->P.hook.pre = prof_pre;
        pb->P.hook.post = prof_post;
        pb->P.hook.ctx = pb;
    } else {
        pb->P.hook = LaunchHook();
    }
    return AAR_OK;
}

int aar_get_kernel_times(aar_problem *pb, double seconds[AAR_NUM_KERNELS], int64_t launches[AAR_NUM_KERNELS]) {
    if (!pb || !seconds || !launches) return set_error(AAR_ERR_INVALID, "aar_get_kernel_times: null argument");
    static_assert(AAR_NUM_KERNELS == KID_COUNT, "kernel id table out of sync with include/aar.h");
    static_assert(ENT_STRIDE == ENT_STRIDE_H, "entity row stride");
    HIP_TRY(hipSetDevice(pb->device));
    HIP_TRY(hipStreamSynchronize(pb->stream));
    prof_harvest(pb);
    for (int i = 0; i < KID_COUNT; i++) { seconds[i] = pb->k_seconds[i]; launches[i] = pb->k_launches[i]; }
    return AAR_OK;
}

const char *aar_kernel_name(int kid) {
    static const char *names[KID_COUNT] = {"k_unpack", "k_residual", "k_passA", "k_passB", "k_maxdiag", "k_frame_inv", "k_schur",
                                           "k_ldl_diag", "k_ldl_trsm", "k_ldl_update", "k_ldl_backsolve", "k_backsub",
                                           "k_reduce_scalars", "k_ldl_panel", "k_pcg", "k_spcg", "k_spcg_pre"};
    return (kid >= 0 && kid < KID_COUNT) ? names[kid] : "?";
}

int aar_problem_pcg_iterations(aar_problem *pb, int32_t out[2]) {
    if (!pb || !out) return set_error(AAR_ERR_INVALID, "aar_problem_pcg_iterations: null argument");
    out[0] = out[1] = 0;
    if (!pb->P.use_pcg && !pb->P.use_spcg) return AAR_OK;
    HIP_TRY(hipSetDevice(pb->device));
    { int rc = copy_d2h(pb, out, pb->P.use_pcg ? pb->P.pcg_counter + 2 : pb->P.spcg_iters, 2 * sizeof(int32_t)); if (rc) return rc; }
    if (out[0] > SPCG_MAX_IT && pb->P.use_spcg) out[0] = SPCG_MAX_IT;   // (a timed-out launch records SPCG_BUFS)
    return AAR_OK;
}

int aar_problem_get_solver_stats(aar_problem *pb, aar_solver_stats *out) {
    if (!pb || !out) return set_error(AAR_ERR_INVALID, "aar_problem_get_solver_stats: null argument");
    // the caller says how large ITS struct is (a caller built against an older header has a shorter one): never write beyond it
    const size_t cap = out->struct_size;
    if (cap < 4 * sizeof(int32_t) || cap > 4096) return set_error(AAR_ERR_INVALID, "aar_solver_stats.struct_size is not set (sizeof(aar_solver_stats) of the caller)");
    aar_solver_stats st;
    memset(&st, 0, sizeof st);
    st.solver = pb->solver;
    st.deterministic = pb->P.deterministic;
    st.pcg_eta = pb->P.pcg_eta;
    st.pcg_max_it = pb->P.use_spcg ? pb->P.spcg_max_it : pb->P.pcg_max_it;
    st.pcg_eta_loose = pb->P.pcg_eta_loose;
    st.pcg_eta_switch = pb->P.pcg_eta_switch;
    st.pcg_abs_tol = pb->P.pcg_abs_tol;
    st.env_overrides = pb->env_overrides;
    st.fallbacks = pb->spcg_fallbacks;
    if (pb->P.use_pcg || pb->P.use_spcg) {
        int32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipSetDevice(pb->device));
        int rc = copy_d2h(pb, c, pb->P.use_spcg ? pb->P.spcg_iters : pb->P.pcg_counter + 2, (pb->P.use_spcg ? 8 : 3) * sizeof(int32_t));
        if (rc) return rc;
        st.last_iterations = std::min(c[0], pb->P.use_spcg ? SPCG_MAX_IT : c[0]);
        st.total_iterations = c[1];
        st.solves = c[2];
        st.same_xcd_solves = c[4];
    }
    st.struct_size = (uint32_t)std::min(cap, sizeof st);
    memcpy(out, &st, std::min(cap, sizeof st));
    return AAR_OK;
}

int aar_problem_set_test_hook(aar_problem *pb, int32_t hook, int32_t value) {
    if (!pb) return set_error(AAR_ERR_INVALID, "aar_problem_set_test_hook: null argument");
    if (hook == AAR_TEST_HOOK_SPCG_DROP) { pb->P.spcg_test_drop = value; return AAR_OK; }
    return set_error(AAR_ERR_INVALID, "aar_problem_set_test_hook: unknown hook %d", hook);
}

int aar_set_stage_timers(aar_problem *pb, int on) {
    if (!pb) return set_error(AAR_ERR_INVALID, "aar_set_stage_timers: null argument");
    HIP_TRY(hipSetDevice(pb->device));
    HIP_TRY(hipStreamSynchronize(pb->stream));
    pb->stage_timers = on != 0;
    memset(&pb->times, 0, sizeof pb->times);
    return AAR_OK;
}

int aar_get_stage_times(aar_problem *pb, aar_stage_times *t) {
    if (!pb || !t) return set_error(AAR_ERR_INVALID, "aar_get_stage_times: null argument");
    *t = pb->times;
    return AAR_OK;
}

}  // extern "C"
