#!/usr/bin/env python3
"""bench.py -- LM iterations/sec of the MI355X-native bundle-adjustment path on synthetic sequences.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 2|3|4|5] [--no-cpu-baseline]

`--gpus N` with N > 1 needs nothing else: when the script is not already running under torch.distributed.run (WORLD_SIZE unset)
it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a
CHILD process -- before this process has made any HIP call -- relays rank 0's JSON line and exits non-zero if any rank failed.

A "step" is one LM iteration = one step() of ucoslam::SparseLevMarq (libs/sparselevmarq.h:349-430): Jacobian /
normal-equation build at the current point, >= 1 damped solve, >= 1 trial residual, accept/reject.  The timed region
runs EXACTLY K steps as back-to-back solve() calls from the same initial guess (a solve takes ~15 steps to stop; the last
one is cut by max_iters so that the total is K), with inputs already resident in HBM, bracketed by a barrier + device
synchronisation, MAX over ranks.  N > 1: launched by `python -m torch.distributed.run` (one rank per GPU); the same problem
is sharded by frame range over the ranks (strong scaling), one RCCL all-reduce of the reduced system per damped solve.

Default workload = BASELINE.json's metric configuration: 8 cameras / 40 markers / 500 frames (configs[2], SURVEY "config 3").
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {2: "synthetic 4-cam/12-marker/100-frame", 3: "synthetic 8-cam/40-marker/500-frame",
             4: "synthetic 8-cam/40-marker/2000-frame", 5: "synthetic 16-cam/200-marker/5000-frame"}
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6      # vector fp64


_RESULT_STREAM = None


def claim_stdout():
    """The result line must be the ONLY thing on this process's standard output (the driver parses it).  Libraries loaded later write there behind Python's back
    (librccl's banner through C stdio, flushed whenever libc likes -- at exit, behind the line): keep a private duplicate of descriptor 1 for the result and point
    descriptor 1 itself at stderr for the rest of the process."""
    global _RESULT_STREAM
    if _RESULT_STREAM is None:
        sys.stdout.flush()
        stdout_to_stderr._flush_c_stdio()
        _RESULT_STREAM = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit_result(line):
    if _RESULT_STREAM is None:
        print(line)
        sys.stdout.flush()
    else:
        _RESULT_STREAM.write(line + "\n")
        _RESULT_STREAM.flush()


class stdout_to_stderr:
    """RCCL greets every new communicator with a version banner on the process's STDOUT (file descriptor 1, from inside librccl): the bench line must stay the only
    thing there, so the descriptor points at stderr while a communicator is made."""
    @staticmethod
    def _flush_c_stdio():
        # librccl writes through C stdio, which buffers fully when stdout is a pipe or a file: without this its banner would sit in libc's buffer past the
        # redirection and come out on descriptor 1 at process exit -- BEHIND the bench line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    def __enter__(self):
        sys.stdout.flush()
        self._flush_c_stdio()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c_stdio()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def run_steps(problem, x0, steps, params_factory):
    """Exactly `steps` LM iterations as repeated solve() calls from x0.  Returns (iterations, trial_points, last x, last report)."""
    done, trials, x, rep = 0, 0, None, None
    while done < steps:
        prm = params_factory(max_iters=steps - done)
        x, rep = problem.lm_solve(x0, params=prm, trace_cap=1)
        if rep["iterations"] <= 0:
            raise RuntimeError("solve made no progress")
        done += rep["iterations"]
        trials += rep["trial_points"]
    return done, trials, x, rep


def kernel_source_hash():
    """sha1 over the HIP sources the kernels are built from: PMC traffic collected for one build is only attached to a bench line
    of the SAME sources (profiles/pmc_traffic.json records the hash, scripts/pmc_summary.py writes it)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "automatic-ar_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


# which roofline bounds a kernel: the observation passes carry ~55-110 flop per byte of the 44-byte record
# (ridge 10 flop/B) -> fp64 vector pipes; the trailing updates / panel solves of the dense LDL^T run on the fp64 matrix pipes;
# single-workgroup kernels are bound by their dependent chain (latency), whatever pipe they use; the rest streams bytes
KERNEL_BOUND = {
    "k_passA": "fp64_valu", "k_passB": "fp64_valu", "k_residual": "fp64_valu", "k_track": "fp64_valu",
    "k_schur": "fp64_valu",            # output-stationary kernel: VALU + LDS atomics (the opt-in AAR_SCHUR_MFMA kernel: fp64_mfma)
    "k_ldl_update": "fp64_mfma", "k_ldl_trsm": "fp64_mfma", "k_ldl_panel": "fp64_mfma",
    "k_ldl_diag": "latency", "k_ldl_backsolve": "latency", "k_reduce_scalars": "latency", "k_maxdiag": "latency",
    "k_backsub": "hbm", "k_frame_inv": "hbm", "k_unpack": "hbm",
    "k_pcg": "hbm",                    # --solver pcg: an iteration is two passes over the W blocks (288 B per (entity, frame) incidence)
    "k_spcg": "latency",               # --solver spcg: one wavefront per entity, an iteration is one hand-over between them (~1.3 us) + a 6 x n matrix-vector product
    "k_spcg_pre": "latency",           # ... its coarse space: one workgroup per entity assembles the augmented rows (one memory round trip + 12 x n products + the stores)
}


FUSED_PANEL_M = int(os.environ.get("AAR_FUSED_PANEL", "3"))     # block columns with at most this many tiles below the diagonal take k_ldl_panel


def _stages(n_pad):
    """(tiles below the diagonal) of every block column that has a panel, split by the kernel(s) that handle it."""
    nT = max(1, n_pad // 96)
    ms = [nT - s - 1 for s in range(nT - 1)]
    return [m for m in ms if m > FUSED_PANEL_M], [m for m in ms if 0 < m <= FUSED_PANEL_M]


def algorithmic_flops(kernel, N, n_pad, sum_kf2, merged_passes):
    """fp64 flops of ONE launch, SURVEY.md 8d's F_iter split by kernel: 4800 per observation for the Jacobian / normal-equation
    work (2600 in pass A incl. the residual, 2200 in pass B, which recomputes projection and Jacobian), 6 s_f^2 = 216 k_f^2 per frame
    for the Schur complement, n^3/3 for the dense LDL^T (per tile: NB^3/3 in the diagonal kernel, rows x NB^2 in the panel solve,
    rows^2 x NB in the trailing update -- both in ONE launch for the block columns k_ldl_panel handles; averaged over the
    launches of each kernel)."""
    NB = 96.0
    split, fused = _stages(n_pad)
    avg = lambda v: (sum(v) / len(v)) if v else 0.0
    # look-ahead (default; AAR_LDL_LOOKAHEAD=0 switches it off): a tall block column's trailing update rides in the NEXT diagonal tile's launch, k_ldl_update is launched
    # only for the last split column -- the riders' flops are k_ldl_diag's then (averaged over its nT launches)
    lookahead = os.environ.get("AAR_LDL_LOOKAHEAD", "1") != "0" and len(split) > 1
    nT = max(1, n_pad // 96)
    ride = sum((m * NB) ** 2 * NB for m in split[:-1]) if lookahead else 0.0
    upd = [(m * NB) ** 2 * NB for m in (split[-1:] if lookahead else split)]
    tab = {
        "k_passA": (4800.0 if merged_passes else 2600.0) * N, "k_passB": 2200.0 * N, "k_residual": 400.0 * N,
        "k_schur": 216.0 * sum_kf2,
        "k_ldl_diag": NB ** 3 / 3.0 + ride / nT,
        "k_ldl_trsm": avg([m * NB * NB * NB for m in split]),
        "k_ldl_update": avg(upd),
        "k_ldl_panel": avg([m * NB * NB * NB + (m * NB) ** 2 * NB for m in fused]),
        "k_ldl_backsolve": 2.0 * n_pad * n_pad / 2.0,
        "k_spcg_pre": 2.0 * n_pad * n_pad * 12.0,   # (A Z): every row against the twelve coarse columns
    }
    return tab.get(kernel, 0.0)


def algorithmic_bytes(kernel, N, A, F, n_pad):
    """Algorithmic HBM bytes of ONE launch: the 44-byte observation record is SURVEY.md 8d's unit;
    the dense factorisation kernels are charged the tiles of the reduced system they must read and write once."""
    P = 6 * (A + F)
    rec = 44 * N
    tile = 8 * 96 * 96
    split, fused = _stages(n_pad)
    avg = lambda v: (sum(v) / len(v)) if v else 0.0
    table = {
        "k_passA": rec + 8 * P,                       # every record once + the pose vector
        "k_passB": rec + 8 * P + 8 * (n_pad * n_pad // 2 + n_pad),   # + the shared system it accumulates (lower triangle)
        "k_residual": rec + 8 * P,
        "k_unpack": 8 * P,
        "k_schur": 8 * (n_pad * n_pad // 2 + n_pad),  # frame-owned W/V traffic is overhead, not algorithmic
        "k_ldl_diag": tile + (sum(tile * m + 2 * tile * m * (m + 1) / 2 for m in split[:-1]) / max(1, n_pad // 96) if (os.environ.get("AAR_LDL_LOOKAHEAD", "1") != "0" and len(split) > 1) else 0),   # lower triangle of the diagonal tile in / out (+ the look-ahead riders' block column and trailing tiles)
        "k_ldl_trsm": avg([2 * tile * m + tile for m in split]),                   # the block column below the diagonal in/out + L_ss
        "k_ldl_update": avg([tile * m + 2 * tile * m * (m + 1) / 2 for m in (split[-1:] if (os.environ.get("AAR_LDL_LOOKAHEAD", "1") != "0" and len(split) > 1) else split)]),   # the block column in, the trailing tiles in/out
        "k_ldl_panel": avg([2 * tile * m + tile + 2 * tile * m * (m + 1) / 2 for m in fused]),   # both of the above in one launch
        "k_ldl_backsolve": 8 * (n_pad * n_pad // 2 + 2 * n_pad),
        "k_frame_inv": 8 * 48 * F * 2, "k_backsub": 8 * P * 2, "k_reduce_scalars": 8 * 3 * F, "k_maxdiag": 8 * (n_pad + 6 * F),
        "k_spcg": 8 * n_pad * n_pad, "k_spcg_pre": 2 * 8 * n_pad * n_pad,          # the reduced system once / read once and its augmented rows written once
    }
    return table.get(kernel, 0)


def stage_timed_pass(problem, x0, params, n_steps):
    """The library's stage timers over >= n_steps LM iterations (never fewer than 120 at the small workloads: the driver's own command asks for 20, and ONE stall of
    this pool's boxes -- tens of milliseconds, seen in round 6 -- then is the whole average), solve by solve; returns ({stage: seconds per LM step, the MEDIAN over
    the solves}, LM steps done).  Every solve of the pass has the same iteration count, so the per-solve figures are comparable."""
    import statistics
    problem.set_stage_timers(True)
    per_solve, done = [], 0
    while done < n_steps:
        _, rep = problem.lm_solve(x0, params=params(max_iters=max(1, n_steps - done)), trace_cap=1)
        it = max(1, rep["iterations"])
        per_solve.append({k: v / it for k, v in problem.stage_times().items()})
        done += it
    problem.set_stage_timers(False)
    keys = set().union(*[d.keys() for d in per_solve])
    return {k: statistics.median([d.get(k, 0.0) for d in per_solve]) for k in keys}, done


def amdahl_split(stage_times, steps, world):
    """Where a step's device time goes, from the library's stage timers (aar_get_stage_times; a separate, instrumented pass: every
    stage is bracketed by HIP events and waited for).  replicated = what EVERY rank does in full whatever the rank count (the dense
    LDL^T chain of the reduced system and its back-substitution, the host's accept / reject turn-around), sharded = what divides by
    the rank count (observation passes, per-frame Schur terms, frame back-substitution), collective = pack + all-reduce + unpack
    (zero without a communicator).  bound_at[n] = the speed-up over one GPU that these measured components allow at n GPUs if
    the sharded part scales perfectly and the collective costs what it costs here -- an upper bound, not a prediction of RCCL."""
    us = lambda k: 1e6 * stage_times.get(k, 0.0) / max(1, steps)
    rep = us("chol") + us("control")
    shard = (us("jacobian_normal_eq") + us("schur") + us("backsub") + us("unpack") + us("residual")) * world   # (per-rank time x ranks = the one-GPU work)
    coll = us("allreduce")
    one_gpu = rep + shard
    out = {"replicated_us": rep, "sharded_us_one_gpu": shard, "collective_us": coll, "n_gpus": world,
           "source": "aar_get_stage_times over %d instrumented steps (event-bracketed stages, host waits between them: the sum exceeds ms_per_step)" % steps,
           "bound_at": {str(n): one_gpu / (rep + shard / n + (coll if n > 1 else 0.0)) for n in (1, 2, 4, 8)}}
    return out


def cpu_baseline(ds, workload, max_threads):
    """The reference solver on the host cores, bounded sample (rank 0, N = 1 only).  OpenMP with every core of a large
    host is not the reference's best case (its mult() does not scale), so a few thread counts are timed and the best
    is reported together with the count that produced it."""
    import oracle_lib as ol
    kind = "reference" if ol.have_ref() else "port"
    if workload >= 5:
        return {"value": None, "unit": "LM iterations/s", "cores": max_threads, "kind": kind,
                "sample": "not run: ONE reference iteration at 1.3 M marker observations takes ~9 min and 15 GB (BASELINE.md section 2)"}
    cap = {2: 15, 3: 4, 4: 1}[workload]
    o = ol.Oracle(ds)
    prm = ol.mapper_params(max_iters=cap)
    what = ("real ucoslam::SparseLevMarq<double> + Eigen::SimplicialLDLT (oracle/_ref, g++ -O3 -march=x86-64-v3 -fopenmp) driving the "
            "restated reference-faithful residual / central-difference float Jacobian") if kind == "reference" else \
           "oracle/ba_oracle.cpp restatement (numeric float Jacobian, map-based Jt*J, own sparse LDL^T)"
    best, tried = None, []
    for th in sorted(set([min(8, max_threads), min(32, max_threads), max_threads])):
        t0 = time.perf_counter()
        fn = o.ref_lm_solve if kind == "reference" else o.lm_solve
        x, rep = fn(ds.x_full, params=prm, jac_mode=ol.JAC_NUMERIC_F32, res_mode=ol.RES_F32, threads=th)
        dt = time.perf_counter() - t0
        rate = rep["iterations"] / dt
        tried.append({"threads": th, "it_per_s": rate, "seconds": dt})
        if best is None or rate > best[0]:
            best = (rate, th, rep, dt)
    rate, th, rep, dt = best
    # the reported sample: ~15 s of CPU work at the best thread count (or the whole solve, if it converges earlier)
    long_cap = int(min(10000, max(cap, 15.0 * rate)))
    reps = 0
    if long_cap > cap:
        its, dt = 0, 0.0
        while dt < 10.0 and reps < 200:      # the whole solve repeated from the same start until ~10 s have been timed
            t0 = time.perf_counter()
            x, rep = fn(ds.x_full, params=ol.mapper_params(max_iters=long_cap), jac_mode=ol.JAC_NUMERIC_F32, res_mode=ol.RES_F32, threads=th)
            dt += time.perf_counter() - t0
            its += rep["iterations"]
            reps += 1
        rate = its / dt
        rep = dict(rep, iterations=its)
    return {"value": rate, "unit": "LM iterations/s", "cores": th, "kind": kind, "host_cores": max_threads, "thread_counts_tried": tried,
            "sample": "%d LM iterations (%d solve(s) from the same start, each run until the solver stops by itself or the cap) of the same %s problem, %.1f s at the best thread count found by the short sweep; %s"
                      % (rep["iterations"], max(reps, 1), WORKLOADS[workload], dt, what),
            "err_after_sample": rep["final_err"]}


def kernel_profile(problem, ds, workload, x0, steps, params, world, intrinsics=False, deterministic=False):
    """Per-kernel device time of `steps` LM iterations of `problem` (an instrumented pass: every launch bracketed by HIP events on the library's own stream) and the
    roofline of the dominant kernel: algorithmic bytes / flops per launch (SURVEY.md 8d's unit) over the measured average duration, with the PMC
    traffic of the same kernel sources attached when profiles/pmc_traffic.json holds it.  Returns (kernels, roofline)."""
    import numpy as np
    A = ds.num_cams + ds.num_markers + (ds.num_cams if intrinsics else 0)     # an intrinsics entity per camera (fx cx fy cy + 2 idle)
    F, n_pad = ds.num_frames, ((6 * A + 95) // 96) * 96
    pcg0 = problem.pcg_iterations()[1]
    problem.set_kernel_profiling(True)
    # in up to eight chunks: a kernel's average is the MEDIAN of its per-chunk averages -- the boxes of this pool stall for tens of milliseconds once in a while
    # (seen: one 75-ms and one 400-ms bracket in 1000-step passes), and one such bracket would otherwise be some kernel's "average"
    # A short timed region (the driver's --steps 20) would leave ONE chunk and no protection: the instrumented pass -- it is not the timed region -- then runs
    # 120 steps (eight whole solves of the headline workload; `launches` and roofline["profiled_steps"] say so).  Config 5 keeps its 45 (three chunks).
    if steps < 120 and workload != 5:
        steps = 120
    n_chunks = max(1, min(8, steps // 15))
    done, prev, per_chunk = 0, {}, {}
    for c in range(n_chunks):
        d, _, _, _ = run_steps(problem, x0, steps // n_chunks + (1 if c < steps % n_chunks else 0), params)
        done += d
        kt = problem.kernel_times()
        for k, (sec, cnt) in kt.items():
            s0, c0 = prev.get(k, (0.0, 0))
            if cnt > c0:
                per_chunk.setdefault(k, []).append((sec - s0) / (cnt - c0))
        prev = kt
    problem.set_kernel_profiling(False)
    pcg_total = problem.pcg_iterations()[1] - pcg0
    med = lambda v: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2])
    kernels = {k: {"total_ms": 1e3 * med(per_chunk[k]) * c, "launches": c, "avg_us": 1e6 * med(per_chunk[k]), "avg_us_mean": 1e6 * s / c} for k, (s, c) in kt.items() if c}
    dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
    n_loc = problem.local_obs
    merged = "k_passB" not in kernels                      # both observation passes ride in k_passA's launch
    kf = np.array([len(set(ds.obs_cam[a:b].tolist())) + len(set(ds.obs_marker[a:b].tolist()))
                   for a, b in zip(*(lambda st: (st[:-1], st[1:]))(np.searchsorted(ds.obs_frame, np.arange(ds.num_frames + 1))))], dtype=np.float64)
    sum_kf2 = float((kf ** 2).sum()) / max(1, world)
    pmc, traffic_tab, src_hash, rocprof_avg = os.path.join(ROOT, "profiles", "pmc_traffic.json"), {}, kernel_source_hash(), {}
    traffic_note = "no PMC collection for this build (profiles/pmc_traffic.json absent)"
    if os.path.exists(pmc):
        try:
            tr = json.load(open(pmc)).get("workload_%d" % workload, {})
            if not tr:
                traffic_note = "no PMC pass was collected for this workload"
            elif tr.get("_kernel_source_sha1") == src_hash:
                traffic_tab, traffic_note = tr, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same kernel sources (%s)" % tr.get("_source")
                rocprof_avg = {("k_passA" if kk == "k_passAB" else kk): vv for kk, vv in tr.get("_rocprofv3_avg_us", {}).items()}
            else:
                traffic_note = "profiles/pmc_traffic.json was collected for other kernel sources (%s): not attached" % str(tr.get("_kernel_source_sha1"))[:12]
        except Exception as e:
            traffic_note = "profiles/pmc_traffic.json unreadable: %s" % e

    def roof(k):
        avg_s = kernels[k]["avg_us"] * 1e-6
        by = algorithmic_bytes(k, n_loc, A, F, n_pad)
        extra = {}
        if k == "k_pcg":
            # SURVEY 8d's accounting: the frame-owned W blocks are NOT algorithmic bytes (an ideal implementation keeps them on chip); what the solver must move per
            # launch is the reduced system once (8 (n^2/2 + n), 8d's "writes then reads the reduced system once").  `utilisation` is the kernel's own byte model instead:
            # one pass over the W blocks for the preconditioner, ONE per CG iteration (k_pcgf; two with k_pcg: deterministic mode, AAR_PCG_FUSED=0), 288 B per incidence
            passes = 2.0 if (deterministic or os.environ.get("AAR_PCG_FUSED") == "0") else 1.0
            # fp32 blocks (144 B per incidence) where the library keeps them: k_pcgf on one rank at a forcing term >= 1e-4 (csrc/ba_capi.hip, PCG_W32_MIN_ETA)
            w32 = (passes == 1.0 and os.environ.get("AAR_PCG_W32") != "0" and problem.solver_stats()["pcg_eta"] >= 1e-4)
            blk_b = 144.0 if w32 else 288.0
            util_by = blk_b * float(kf.sum()) / max(1, world) * (1.0 + passes * pcg_total / float(done))
            by = 8.0 * (n_pad * n_pad / 2 + n_pad)
            extra["utilisation"] = {"bytes_per_launch": util_by, "achieved": util_by / avg_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": util_by / avg_s / 1e9 / HBM_PEAK_GBPS,
                                    "model": "%d B per (entity, frame) incidence per pass over W (%s blocks): the kernel's own traffic model (frame-owned blocks: overhead by SURVEY 8d, not algorithmic bytes)" % (int(blk_b), "fp32" if w32 else "fp64")}
        if k == "k_spcg":      # both triangles of the reduced system once into registers; a 6 x n product per wavefront per iteration (+ the one of the set-up)
            by = 8.0 * n_pad * n_pad
        if k == "k_passA" and merged:
            by += algorithmic_bytes("k_passB", n_loc, A, F, n_pad)
        fl = algorithmic_flops(k, n_loc, n_pad, sum_kf2, merged)
        if k == "k_spcg":
            fl = 2.0 * n_pad * n_pad * (1.0 + pcg_total / float(done))
        kind = KERNEL_BOUND.get(k, "hbm")
        if k == "k_ldl_diag" and os.environ.get("AAR_LDL_LOOKAHEAD", "1") != "0" and len(_stages(n_pad)[0]) > 1:
            kind = "fp64_mfma"     # (its launches carry the tall block columns' trailing updates as riders: matrix-pipe work, not a lone tile's chain)
        if k == "k_schur" and (A >= 96 or os.environ.get("AAR_SCHUR_MFMA") == "1") and os.environ.get("AAR_SCHUR_MFMA") != "0":
            kind = "fp64_mfma"     # from 96 shared entities on: dense panels through the fp64 matrix pipes (k_schur_fill + k_schur_mfma)
        r = {"kernel": k, "avg_us": kernels[k]["avg_us"], "bytes_per_launch": by, "flops_per_launch": fl,
             "hbm": {"achieved": by / avg_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": by / avg_s / 1e9 / HBM_PEAK_GBPS},
             "fp64": {"achieved": fl / avg_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / avg_s / 1e12 / FP64_PEAK_TFLOPS},
             "traffic": traffic_tab.get("k_passAB" if (k == "k_passA" and merged) else k)}
        r.update(extra)
        if k == "k_schur" and r["traffic"] is None and "k_schur_mfma" in traffic_tab:      # two kernels behind one launcher; k_schur_fill only counts when it ran
            fill = traffic_tab.get("raw_k_schur_fill", {}).get("launches", 0) / float(max(1, traffic_tab.get("raw_k_schur_mfma", {}).get("launches", 1)))
            r["traffic"] = traffic_tab["k_schur_mfma"] + fill * traffic_tab.get("k_schur_fill", 0.0)      # (fill launches per mfma launch: ~0.07, the steps whose damping was not the predicted one)
        if k == "k_pcg" and r["traffic"] is None:                                            # the launcher's kernels: k_pcgf on one GPU, k_pcgd_* with ranks
            r["traffic"] = traffic_tab.get("k_pcgf", traffic_tab.get("k_pcgd_iter_f"))
        # `traffic` doubles FETCH_SIZE (the guide's gfx950 rule, stated for wide coalesced read streams: it may overstate a gather kernel); the uncorrected sum beside it
        rk = "k_passAB" if (k == "k_passA" and merged) else ("k_pcgf" if (k == "k_pcg" and "raw_k_pcgf" in traffic_tab) else k)
        raw = traffic_tab.get("raw_" + rk)
        r["traffic_uncorrected"] = (raw["FETCH_SIZE_KB"] + raw["WRITE_SIZE_KB"]) * 1024.0 if isinstance(raw, dict) and r["traffic"] is not None else None
        # the contract's fields, priced against the roofline that applies to this kernel: `bound` says which -- "hbm" (GB/s),
        # "fp64_valu" / "fp64_mfma" (TFLOP/s against the 78.6 TFLOP/s fp64 peak of the vector / matrix pipes), or "latency": a
        # single-workgroup dependent chain, for which a throughput fraction says nothing -- its useful flops over its duration
        # are reported when it has any (k_ldl_diag), and `frac` is left out when it has none
        view = r["hbm"] if kind == "hbm" else r["fp64"]
        r.update(bound=kind, achieved=view["achieved"], peak=view["peak"], unit=view["unit"], frac=view["frac"])
        if kind == "latency" and fl == 0.0:
            for f in ("achieved", "peak", "unit", "frac"):
                r.pop(f)
        if k in rocprof_avg:
            r["avg_us_rocprofv3"] = rocprof_avg[k]     # the same kernel under rocprofv3 --kernel-trace --stats (profiles/, same sources)
        elif k == "k_schur" and "k_schur_mfma" in rocprof_avg:
            r["avg_us_rocprofv3"] = rocprof_avg["k_schur_mfma"]     # (k_schur_fill only runs when the damping was not the predicted one)
        return r
    roofline = roof(dom)
    roofline["traffic_source"] = traffic_note
    roofline["kernel_source_sha1"] = src_hash
    roofline["observation_pass"] = roof("k_passA")     # the streaming scan the north star prices against HBM
    roofline["per_kernel"] = {k: {f: v for f, v in roof(k).items() if f in ("bound", "achieved", "peak", "unit", "frac", "avg_us", "avg_us_rocprofv3", "traffic", "traffic_uncorrected", "utilisation")}
                              for k in kernels if k in KERNEL_BOUND}
    tj = (kernels["k_passA"]["avg_us"] + (kernels["k_passB"]["avg_us"] if "k_passB" in kernels else 0.0)) * 1e-6
    roofline["fp64_valu"] = {"kernels": "k_passA+k_passB" if not merged else "k_passA (passes A and B in one launch)", "achieved": 4800.0 * n_loc / tj / 1e12,
                             "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 4800.0 * n_loc / tj / 1e12 / FP64_PEAK_TFLOPS, "flops_per_observation": 4800,
                             # SURVEY 8d's unit (rows x Jacobian blocks).  The passes in wrench form reach the same blocks through the observation's 6x6 Gram
                             # matrix: ~1 500 (pass A, slot images included) + ~1 700 (pass B) executed flops per observation
                             "executed_flops_per_observation": 3200, "executed_frac": 3200.0 * n_loc / tj / 1e12 / FP64_PEAK_TFLOPS}
    roofline["sum_kernel_us_per_step"] = sum(v["total_ms"] for v in kernels.values()) * 1e3 / float(done)     # all launches of the pass / its LM steps
    roofline["sum_kernel_us_per_step_note"] = ("HIP-event brackets around every launch: each adds ~2-3 us to what rocprofv3 reports for the kernel itself, so this sum EXCEEDS "
                                               "ms_per_step of the un-instrumented run")
    roofline["profiled_steps"] = int(done)             # LM steps of the instrumented pass (>= 120 at configs 2-4 whatever --steps says): `kernels[*].launches` belong to these
    return kernels, roofline


def iteration_hbm(ds, trials_per_step, it_per_s):
    """algorithmic bytes of one LM iteration with t trial points over the measured rate"""
    P = ds.full_len
    Ps = 6 * (ds.num_cams - 1 + ds.num_markers - 1)
    b_iter = 44 * ds.num_obs * (1 + trials_per_step) + 8 * (2 * P + Ps * Ps + Ps)
    return {"bytes_per_iteration": b_iter, "achieved": b_iter * it_per_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_iter * it_per_s / 1e9 / HBM_PEAK_GBPS}


def other_workload(aar, w, device, steps, warmup):
    """One more workload on ONE GPU in the same bench line (`other_workloads`): configs 4 and 5 -- the configurations where a roofline means something (config 3's
    step is 1.8 MB of algorithmic bytes: a latency chain) -- through the library's default solver: LM it/s over exactly `steps` steps after `warmup` (device
    synchronised on both sides), what AUTO resolved to, CG iterations per LM step, final error and POSES against the direct solver on the same problem,
    SURVEY 8d's whole-iteration HBM figure, the dominant kernel's roofline (PMC traffic attached when collected for these sources) and the observation
    passes' fp64 fraction."""
    from pose_metrics import pose_delta_max
    ds = aar.synth(w)
    params = lambda **kw: aar.lm_default_params(**kw)
    out = {"workload": WORKLOADS[w], "survey_config": w, "marker_observations": int(ds.num_obs), "steps": steps, "warmup": warmup, "unit": "LM iterations/s"}
    with aar.Problem(ds, residual_mode=aar.RES_F32, device=device) as problem:          # the library's default options: solver AUTO
        for _ in range(2):
            problem.lm_solve(ds.x_full, params=params(), trace_cap=1)
        run_steps(problem, ds.x_full, warmup, params)
        st0 = problem.solver_stats()
        aar.lib().aar_device_synchronize()
        t0 = time.perf_counter()
        done, trials, _, _ = run_steps(problem, ds.x_full, steps, params)
        aar.lib().aar_device_synchronize()
        dt = time.perf_counter() - t0
        st1 = problem.solver_stats()
        x_fin, rep_fin = problem.lm_solve(ds.x_full, params=params())
        rmse, _ = problem.reproj_stats(x_fin)
        kernels, roofline = kernel_profile(problem, ds, w, ds.x_full, steps, params, 1)
        out.update(value=done / dt, ms_per_step=1e3 * dt / done, solver_resolved=st1["solver"], solver_fallbacks=st1["fallbacks"],
                   cg_iterations_per_lm_step=((st1["total_iterations"] - st0["total_iterations"]) / float(done)) if st1["solver"] != "direct" else None,
                   forcing_sequence={"pcg_eta_loose": st1["pcg_eta_loose"], "pcg_eta": st1["pcg_eta"], "pcg_eta_switch": st1["pcg_eta_switch"]},
                   final_rmse_px=rmse, lm_iterations_to_stop=rep_fin["iterations"], trial_points_per_step=trials / float(done),
                   iteration_hbm=iteration_hbm(ds, trials / float(done), done / dt), roofline=roofline, kernels=kernels)
    if out["solver_resolved"] != "direct":
        with aar.Problem(ds, residual_mode=aar.RES_F32, device=device, solver="direct") as pdir:
            pdir.lm_solve(ds.x_full, params=params(), trace_cap=1)
            n_d = min(steps, 30)
            aar.lib().aar_device_synchronize()
            t0 = time.perf_counter()
            run_steps(pdir, ds.x_full, n_d, params)
            aar.lib().aar_device_synchronize()
            dt_d = time.perf_counter() - t0
            x_d, rep_d = pdir.lm_solve(ds.x_full, params=params())
            rmse_d, _ = pdir.reproj_stats(x_d)
        dR, dT = pose_delta_max(ds, x_fin, x_d)
        out.update(direct={"it_per_s": n_d / dt_d, "steps": n_d, "lm_iterations_to_stop": rep_d["iterations"], "final_rmse_px": rmse_d},
                   rmse_delta_vs_direct_px=abs(rmse - rmse_d), pose_delta_vs_direct={"rotation_matrix_entries": dR, "translation_m": dT})
    if out["solver_resolved"] == "pcg":
        out["block_storage"] = pcg_block_storage(aar, ds, device, steps, warmup, params, x_fin)
    out["amdahl"] = single_rank_comm_leg(aar, ds, device, steps, warmup, params, out["value"])
    return out


def single_rank_comm_leg(aar, ds, device, steps, warmup, params, one_gpu_it_per_s):
    """What a SCALE run of this workload can be held against: the same problem behind a SINGLE-RANK RCCL communicator on this GPU -- the
    code path of N > 1 (fused all-reduce of S | rhs | g0 | scalars per damped solve, or of 8 n bytes per CG iteration with PCG; replicated solve; speculative next
    solve) with nothing to shard -- timed, its collectives counted, and a stage-timed pass split into replicated / sharded / collective microseconds per LM step.
    predicted_upper_bound_it_per_s[n] = 1e6 / (replicated + sharded / n + collective): perfect sharding, the collective at its single-rank cost (a real ring over
    n GPUs costs more), no skew.  PCG: the damped solve itself shards (its frame passes), one kernel launch + one all-reduce per CG iteration: the split then counts
    the solve as sharded and reports launches and all-reduces per LM step."""
    try:
        with stdout_to_stderr():
            comm = aar.Comm(aar.Comm.make_id(), 1, 0, device)
    except Exception as e:   # no RCCL on this box: say so, keep the line
        return {"error": "single-rank RCCL communicator: %s" % e}
    out = {"source": "the workload behind a single-rank RCCL communicator on this GPU (the N > 1 code path, nothing to shard)"}
    with aar.Problem(ds, residual_mode=aar.RES_F32, device=device, comm=comm, solver="auto") as pc:
        for _ in range(2):
            pc.lm_solve(ds.x_full, params=params(), trace_cap=1)
        n_t = max(15, min(steps, 60))
        run_steps(pc, ds.x_full, min(warmup, 15), params)
        c0 = comm.stats()
        aar.lib().aar_device_synchronize()
        t0 = time.perf_counter()
        done, _, _, _ = run_steps(pc, ds.x_full, n_t, params)
        aar.lib().aar_device_synchronize()
        dt = time.perf_counter() - t0
        c1 = comm.stats()
        st = pc.solver_stats()
        acc, done_am = stage_timed_pass(pc, ds.x_full, params, max(n_t, 60))     # (per LM step, median over the solves)
    us = lambda k: 1e6 * acc.get(k, 0.0)
    pcg = st["solver"] == "pcg"
    rep = us("control") + (0.0 if pcg else us("chol"))
    shard = us("jacobian_normal_eq") + us("schur") + us("backsub") + us("unpack") + us("residual") + (us("chol") if pcg else 0.0)
    coll = us("allreduce")
    # the stage-timed pass serialises host and device: scale its split to the step time measured without timers
    t_step = 1e6 * dt / done
    k = t_step / max(1e-9, rep + shard + coll)
    rep, shard, coll = rep * k, shard * k, coll * k
    out.update(solver_resolved=st["solver"], it_per_s_single_rank_rccl=done / dt, us_per_step_single_rank_rccl=t_step, it_per_s_without_communicator=one_gpu_it_per_s,
               communicator_overhead_us_per_step=t_step - 1e6 / one_gpu_it_per_s,
               allreduce_calls_per_lm_step=(c1["allreduce_calls"] - c0["allreduce_calls"]) / float(done),
               allreduce_bytes_per_lm_step=(c1["allreduce_bytes"] - c0["allreduce_bytes"]) / float(done),
               cg_iterations_per_lm_step=st["total_iterations"] / max(1, st["solves"]) if st["solver"] != "direct" else None,
               replicated_us=rep, sharded_us_one_gpu=shard, collective_us=coll,
               split_note="stage-timer shares of an instrumented pass, scaled to the un-instrumented step time; PCG: the solve counts as sharded, one launch + one all-reduce per CG iteration" if pcg
               else "stage-timer shares of an instrumented pass, scaled to the un-instrumented step time; the reduced-system solve is replicated",
               bound_at={str(n): (rep + shard + coll) / (rep + shard / n + coll) for n in (1, 2, 4, 8)},
               predicted_upper_bound_it_per_s={str(n): 1e6 / (rep + shard / n + coll) for n in (2, 4, 8)})
    return out


def pcg_block_storage(aar, ds, device, steps, warmup, params, x_fin, intrinsics=False, solver=None):
    """The PCG mode keeps the frame-entity coupling blocks W in fp32 (storage only: every product and sum is fp64; forcing terms >= 1e-4).  So that the
    figure of this line can be judged: the SAME measurement with fp64 blocks (AAR_PCG_W32=0, read when a problem is created) and how far the two runs' final poses are apart."""
    from pose_metrics import pose_delta_max
    if os.environ.get("AAR_PCG_W32") == "0":
        return {"W": "fp64 (AAR_PCG_W32=0)"}
    os.environ["AAR_PCG_W32"] = "0"
    try:
        with aar.Problem(ds, residual_mode=aar.RES_F32, device=device, intrinsics=intrinsics, solver=solver) as p64:
            x0 = p64.x_with_intrinsics(ds.x_full) if intrinsics else ds.x_full
            p64.lm_solve(x0, params=params(), trace_cap=1)
            n = min(steps, 30)
            run_steps(p64, x0, min(warmup, 15), params)
            aar.lib().aar_device_synchronize()
            t0 = time.perf_counter()
            run_steps(p64, x0, n, params)
            aar.lib().aar_device_synchronize()
            dt = time.perf_counter() - t0
            x64, rep64 = p64.lm_solve(x0, params=params())
    finally:
        del os.environ["AAR_PCG_W32"]
    dR, dT = pose_delta_max(ds, x_fin, x64)
    return {"W": "fp32 storage, fp64 arithmetic (pass A writes the fp32 blocks; k_pcgf, the back-substitution read them)", "switch": "AAR_PCG_W32=0 keeps fp64 blocks",
            "with_fp64_blocks": {"it_per_s": n / dt, "ms_per_step": 1e3 * dt / n, "steps": n, "lm_iterations_to_stop": rep64["iterations"]},
            "pose_delta_fp32_vs_fp64_blocks": {"rotation_matrix_entries": dR, "translation_m": dT}}


def scaling_workload(aar, w, world, rank, local_rank, comm, dist, steps, warmup):
    """One more workload under the SAME communicator (N > 1 runs: configs 4 and 5, the ones BASELINE.json shards over 8 GPUs), with the solver AUTO picks:
    LM it/s (barrier + device synchronisation on both sides, max over ranks), what AUTO resolved to, the all-reduce traffic, the split of a step into
    replicated / sharded / collective time.  Every rank calls this; every rank gets the same dict (rank 0 prints it)."""
    import numpy as np
    ds = aar.synth(w)
    problem = aar.Problem(ds, residual_mode=aar.RES_F32, device=local_rank, comm=comm, solver="auto")
    params = lambda **kw: aar.lm_default_params(**kw)
    calls0 = comm.stats()["allreduce_calls"] if comm is not None else 0

    def barrier():
        aar.lib().aar_device_synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(2):
        problem.lm_solve(ds.x_full, params=params(), trace_cap=1)
    run_steps(problem, ds.x_full, warmup, params)
    barrier()
    t0 = time.perf_counter()
    done, trials, _, _ = run_steps(problem, ds.x_full, steps, params)
    aar.lib().aar_device_synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        dist.barrier()
    x_fin, rep_fin = problem.lm_solve(ds.x_full, params=params())
    rmse, _ = problem.reproj_stats(x_fin)
    st = problem.solver_stats()
    calls1 = comm.stats()["allreduce_calls"] if comm is not None else 0     # (before the instrumented pass below, which has collectives of its own)
    amdahl = None
    if st["solver"] != "pcg":
        per_step, done_am = stage_timed_pass(problem, ds.x_full, params, max(60, min(steps, 120)))
        amdahl = amdahl_split(per_step, 1, world)
        amdahl["source"] = amdahl["source"].replace("over 1 instrumented steps", "per LM step, the median over the solves of %d instrumented steps" % done_am)
    per_rank = [int(problem.local_obs)]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, int(problem.local_obs))
    cs = comm.stats() if comm is not None else None
    out = {"workload": WORKLOADS[w], "value": done / dt, "unit": "LM iterations/s", "ms_per_step": 1e3 * dt / done, "steps": steps, "warmup": warmup,
           "solver_resolved": st["solver"], "cg_iterations_per_lm_step": st["total_iterations"] / max(1, st["solves"]) if st["solver"] != "direct" else None,
           "solver_fallbacks": st["fallbacks"], "final_rmse_px": rmse, "lm_iterations_to_stop": rep_fin["iterations"], "local_obs": per_rank,
           "ranks_seen": cs["ranks_seen"] if cs else 1, "allreduce_bytes": cs["system_allreduce_bytes"] if cs else 0,
           "allreduce_calls_per_lm_step": ((calls1 - calls0) / float(max(1, done + warmup + 3 * rep_fin["iterations"]))) if cs else 0.0,
           "amdahl": amdahl}
    problem.close()
    return out


def self_launch(n, argv):
    """Parent of an N-GPU run: one fresh child process per GPU through torch.distributed.run.  This process never touches the GPU
    (no `import aar`, no torch.cuda call), so nothing that has initialised HIP is ever re-executed or forked."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["AAR_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for raw in proc.stdout:                       # relay: the one JSON line of rank 0 goes to stdout, anything else to stderr
        txt = raw.strip()
        if txt.startswith("{") and '"metric"' in txt:
            try:
                json.loads(txt)
                line = txt
                continue
            except ValueError:
                pass
        sys.stderr.write(raw)
    rc = proc.wait()
    if line is not None and rc == 0:
        print(line)
        sys.stdout.flush()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank child run failed (exit code %d)\n" % (n, rc))
        raise SystemExit(rc if 0 < rc < 256 else 1)
    if line is None:
        sys.stderr.write("bench.py: the child run printed no result line\n")
        raise SystemExit(1)


def plumbing_only(args, world, rank, dist):
    """--plumbing-only: everything of the N-rank path that needs no GPU -- rendezvous, shard plan, the gathers and the
    max-over-ranks of the timing, the JSON line -- so that the launcher can be tested on a CPU-only machine (tests/test_bench_launcher.py)."""
    import numpy as np

    import aar
    ds = aar.synth(args.workload)
    begin = aar.plan_shards(np.bincount(ds.obs_frame, minlength=ds.num_frames), world)
    local_obs = int(np.sum((ds.obs_frame >= begin[rank]) & (ds.obs_frame < begin[rank + 1])))
    fail = os.environ.get("AAR_BENCH_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        raise SystemExit(3)                       # test hook: a rank that dies after the rendezvous
    per_rank, dt = [local_obs], 0.001 * (rank + 1)
    if dist is not None:
        import torch
        dist.barrier()
        per_rank = [None] * world
        dist.all_gather_object(per_rank, local_obs)
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        dist.barrier()
    # the rank logic of scaling_workloads without a GPU: every rank plans the shards of the extra workloads and the ranks agree on the per-rank counts
    scaling = None
    if world > 1 and not args.no_scaling_workloads:
        scaling = {}
        for w in (4, 5):
            if w == args.workload:
                continue
            dw = aar.synth(w) if w != 5 else aar.synth(5, num_frames=int(os.environ.get("AAR_BENCH_PLUMBING_FRAMES5", "5000")))
            bw = aar.plan_shards(np.bincount(dw.obs_frame, minlength=dw.num_frames), world)
            mine = int(np.sum((dw.obs_frame >= bw[rank]) & (dw.obs_frame < bw[rank + 1])))
            counts = [mine]
            if dist is not None:
                counts = [None] * world
                dist.all_gather_object(counts, mine)
            scaling[str(w)] = {"workload": WORKLOADS[w], "local_obs": counts, "marker_observations": int(dw.num_obs), "frames": [int(bw[r + 1] - bw[r]) for r in range(world)]}
    if rank == 0:
        emit_result(json.dumps({"metric": "LM iterations/sec", "value": None, "unit": "LM iterations/s", "n_gpus": world, "plumbing_only": True,
                                "ranks_seen": world, "local_obs": per_rank, "max_rank_seconds": dt, "scaling_workloads": scaling,
                                "config": {"workload": WORKLOADS[args.workload], "marker_observations": int(ds.num_obs)}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", type=int, default=3, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--intrinsics", action="store_true", help="the reference's default Config: optimize_cam_intrinsics on (9 more parameters per camera, "
                    "libs/multicam_mapper.h:75-81); the headline metric is quoted WITHOUT it (SURVEY.md section 8 row f4)")
    ap.add_argument("--no-amdahl", action="store_true", help="skip the stage-timer pass behind the `amdahl` object")
    ap.add_argument("--solver", choices=("direct", "spcg", "pcg", "auto"), default=None, help="aar_solver_options.solver.  Not given: the problem is created with a NULL "
                    "options pointer, i.e. THE LIBRARY'S DEFAULT (AUTO: picks by size) -- what aar_find_solution and MultiCamMapper run.  direct (Schur complement + dense "
                    "LDL^T, the reference's step to rounding), spcg (the same Schur complement, then CG on the explicit reduced system, csrc/spcg_kernels.hip), pcg (CG "
                    "through the frame blocks, no Schur complement, csrc/pcg_kernels.hip)")
    ap.add_argument("--pcg-eta", type=float, default=None, help="aar_solver_options.pcg_eta (forcing term of the inexact solvers; default: the library's)")
    ap.add_argument("--pcg-eta-loose", type=float, default=None, help="aar_solver_options.pcg_eta_loose (> pcg_eta: a forcing sequence, loose early / pcg_eta late; default: none)")
    ap.add_argument("--pcg-abs-tol", type=float, default=None, help="aar_solver_options.pcg_abs_tol (absolute tolerance of an inner solve in pose units, beside the relative one)")
    ap.add_argument("--deterministic", action="store_true", help="aar_solver_options.deterministic: fixed-order sums instead of fp64 atomics (bit-identical runs)")
    ap.add_argument("--no-scaling-workloads", action="store_true", help="N > 1: skip the extra measurements of configs 4 and 5 (scaling_workloads)")
    ap.add_argument("--no-other-workloads", action="store_true", help="N = 1: skip the extra measurements of configs 4 and 5 (other_workloads)")
    ap.add_argument("--no-direct", action="store_true", help="skip the comparison leg through the direct solver (profiling runs: only the chosen solver's kernels in the trace)")
    ap.add_argument("--plumbing-only", action="store_true", help="launcher / rendezvous / JSON relay only, no GPU work (CPU test of the N-rank plumbing)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])      # before anything below can touch the GPU
    claim_stdout()      # (a rank process: from here on only emit_result() reaches the real standard output)
    if os.environ.get("AAR_BENCH_TEST_NOISE") == "1":   # (tests/test_bench_launcher.py: a library writing to standard output through C stdio, as librccl does)
        import ctypes
        ctypes.CDLL(None).puts(b"noise from a library on C stdout")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or leave the launching to bench.py)" % (args.gpus, world, args.gpus))
    dist = None
    if world > 1 or os.environ.get("AAR_FORCE_COMM") == "1":
        # torch.distributed is rendezvous plumbing only (id broadcast, barrier, max over ranks); the data path is RCCL inside libaar
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # single node: rendezvous over loopback, no host-name lookups
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    if args.plumbing_only:
        return plumbing_only(args, world, rank, dist)
    import numpy as np

    import aar

    if aar.device_count() < 1:
        raise SystemExit("bench.py: no HIP device (the product has no CPU path)")
    ds = aar.synth(args.workload)
    comm = None
    if world > 1 or os.environ.get("AAR_FORCE_COMM") == "1":   # the env switch exercises the RCCL path on one GPU
        uid = [aar.Comm.make_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        with stdout_to_stderr():
            comm = aar.Comm(uid[0], world, rank, local_rank)
    problem = aar.Problem(ds, residual_mode=aar.RES_F32, device=local_rank, comm=comm, intrinsics=args.intrinsics, solver=args.solver,
                          deterministic=True if args.deterministic else None, pcg_eta=args.pcg_eta, pcg_eta_loose=args.pcg_eta_loose, pcg_abs_tol=args.pcg_abs_tol)
    # x_full of the default Config: the pose vector, then fx cx fy cy d0..d4 per camera (fill_io_vec_cam_intrinsics, :488-498)
    x0 = problem.x_with_intrinsics(ds.x_full) if args.intrinsics else ds.x_full

    def barrier():
        aar.lib().aar_device_synchronize()
        if dist is not None:
            dist.barrier()

    params = lambda **kw: aar.lm_default_params(**kw)
    # ---- device wake-up (untimed; NOT the LM warm-up, which follows) ----
    # The first milliseconds of GPU activity of a fresh process carry a one-off stall of ~8-13 ms (seen on every MI355X box of
    # the pool: power state / copy-engine bring-up, 2-3 ms after the first kernel; scripts/probe/outlier2.py).  With --steps 20
    # the timed region is 3 ms, so the stall landed inside it in ~30 % of the runs (1.2-1.8 k it/s instead of 6.9 k).  Whole
    # solves of the same problem for 50 ms of wall time put it behind us; nothing is cached from them (every solve restarts
    # from x0 and rebuilds all blocks).
    t_w, n_w = time.perf_counter(), 0
    # (with several ranks every solve is collective: a FIXED count then, the same on every rank, never a clock)
    while (n_w < 12) if comm is not None else (time.perf_counter() - t_w < 0.05):
        problem.lm_solve(x0, params=params(), trace_cap=1)
        n_w += 1
    aar.lib().aar_device_synchronize()
    wakeup = {"seconds": round(time.perf_counter() - t_w, 4), "solves": n_w}
    # ---- warmup (untimed) ----
    if args.warmup > 0:
        run_steps(problem, x0, args.warmup, params)
    # ---- timed region: exactly K steps ----
    solver = problem.solver_stats()["solver"]          # what AUTO resolved to
    pcg0 = problem.pcg_iterations()[1]
    barrier()
    t0 = time.perf_counter()
    done, trials, _, _ = run_steps(problem, x0, args.steps, params)
    aar.lib().aar_device_synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        dist.barrier()
    assert done == args.steps
    pcg_total = problem.pcg_iterations()[1] - pcg0

    # ---- full solve for the accuracy half of the metric ----
    x_fin, rep_fin = problem.lm_solve(x0, params=params())
    rmse, ss = problem.reproj_stats(x_fin)
    pose_delta = None
    # ---- the same through the direct solver (the reference's step to rounding): rate, LM steps to stop, final error -- what an inexact solver is judged against ----
    direct = None
    if solver != "direct" and not args.no_direct:
        with aar.Problem(ds, residual_mode=aar.RES_F32, device=local_rank, comm=comm, intrinsics=args.intrinsics, solver="direct",
                         deterministic=True if (args.deterministic and ds.num_cams + ds.num_markers < 96) else None) as pdir:
            for _ in range(3):
                pdir.lm_solve(x0, params=params(), trace_cap=1)
            n_d = min(args.steps, 200)
            run_steps(pdir, x0, min(args.warmup, 50), params)
            barrier()
            t0 = time.perf_counter()
            run_steps(pdir, x0, n_d, params)
            aar.lib().aar_device_synchronize()
            dt_d = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt_d], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_d = float(t[0])
            x_d, rep_d = pdir.lm_solve(x0, params=params())
            rmse_d, _ = pdir.reproj_stats(x_d)
            direct = {"it_per_s": n_d / dt_d, "steps": n_d, "lm_iterations_to_stop": rep_d["iterations"], "final_rmse_px": rmse_d}
            from pose_metrics import pose_delta_max
            dR, dT = pose_delta_max(ds, x_fin, x_d)        # final POSES as transforms: largest rotation-matrix-entry / translation difference against the direct solver's
            pose_delta = {"rotation_matrix_entries": dR, "translation_m": dT}

    # ---- per-kernel device time: a second, instrumented pass over the same steps (HIP events on the library's stream) ----
    roofline, kernels = None, None
    if not args.no_kernel_profile:
        kernels, roofline = kernel_profile(problem, ds, args.workload, x0, args.steps, params, world, intrinsics=args.intrinsics, deterministic=args.deterministic)

    # ---- where the step's time goes by how it scales with the rank count: a third, stage-timed pass ----
    amdahl = None
    if not args.no_amdahl and solver != "pcg":     # (the PCG solver has no replicated part to speak of: nothing for this split to say)
        per_step, done_am = stage_timed_pass(problem, x0, params, max(120, min(args.steps, 300)))
        amdahl = amdahl_split(per_step, 1, world)
        amdahl["source"] = amdahl["source"].replace("over 1 instrumented steps", "per LM step, the median over the solves of %d instrumented steps" % done_am)

    # ---- next-row extra (not the headline metric): track(), every frame's own 6-DoF LM in one launch ----
    track = None
    if world == 1 and not args.intrinsics:
        ns = 6 * (ds.num_cams - 1) + 6 * (ds.num_markers - 1)
        x_tr = np.array(x_fin)
        x_tr[ns:] = ds.x_full[ns:]                 # cameras / markers at the solution, frame poses back at the initial guess
        problem.track(x_tr)                         # warm-up
        aar.lib().aar_device_synchronize()
        t1 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            xt, it_t, err_t = problem.track(x_tr)
        aar.lib().aar_device_synchronize()
        dt_tr = (time.perf_counter() - t1) / reps
        track = {"frames": int(ds.num_frames), "seconds_per_call": dt_tr, "frames_per_s": ds.num_frames / dt_tr,
                 "mean_lm_iterations_per_frame": float(np.mean(it_t)), "max_pose_delta_vs_bundle_solution": float(np.abs(xt[ns:] - x_fin[ns:]).max())}

    # ---- N = 1: configs 4 and 5 on this GPU in the same line (the configurations with a meaningful roofline) ----
    others = None
    if world == 1 and comm is None and not args.no_other_workloads and not args.intrinsics and args.solver is None and not args.deterministic and args.pcg_eta is None and args.pcg_eta_loose is None and args.pcg_abs_tol is None:
        others = {}
        for w, (st_w, wu_w) in ((4, (90, 30)), (5, (45, 15))):
            if w != args.workload:
                others[str(w)] = other_workload(aar, w, local_rank, st_w, wu_w)

    # ---- PCG keeps its W blocks in fp32: the same measurement with fp64 blocks beside it ----
    block_storage = None
    if solver == "pcg" and world == 1 and comm is None and not args.deterministic and (args.pcg_eta is None or args.pcg_eta >= 1e-4):
        block_storage = pcg_block_storage(aar, ds, local_rank, args.steps, args.warmup, params, x_fin, intrinsics=args.intrinsics, solver=args.solver)

    # ---- N > 1 (or AAR_BENCH_SCALING=1 behind a single-rank communicator): the workloads BASELINE.json shards over 8 GPUs, in the same line ----
    scaling = None
    if (world > 1 or os.environ.get("AAR_BENCH_SCALING") == "1") and not args.no_scaling_workloads:
        scaling = {}
        for w, (st_w, wu_w) in ((4, (90, 30)), (5, (45, 15))):
            if w != args.workload:
                scaling[str(w)] = scaling_workload(aar, w, world, rank, local_rank, comm, dist, st_w, wu_w)

    per_rank_obs = [int(problem.local_obs)]
    comm_stats = comm.stats() if comm is not None else None
    if dist is not None:
        per_rank_obs = [None] * world
        dist.all_gather_object(per_rank_obs, int(problem.local_obs))
    if rank != 0:
        problem.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    P = ds.full_len
    Ps = 6 * (ds.num_cams - 1 + ds.num_markers - 1)
    t_avg = trials / float(done)
    out = {
        "metric": "LM iterations/sec", "value": done / dt, "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / done, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload], "survey_config": args.workload, "cams": ds.num_cams, "markers": ds.num_markers,
                   "frames": ds.num_frames, "marker_observations": int(ds.num_obs), "residual_rows": int(8 * ds.num_obs), "unknowns": int(P),
                   "reduced_unknowns": int(Ps), "parallelism": "frames sharded over %d GPU(s)" % world, "seed": 20190219 + args.workload,
                   "residual_mode": "float32-faithful", "jacobian": "analytic",
                   "optimize_cam_intrinsics": bool(args.intrinsics), "solver": args.solver or "library default (NULL options: auto)", "solver_resolved": solver, "deterministic": bool(args.deterministic)},
        "final_rmse_px": rmse, "final_sum_sq": ss, "lm_iterations_to_stop": rep_fin["iterations"], "trial_points_per_step": t_avg,
        "iteration_hbm": iteration_hbm(ds, t_avg, done / dt),
        "roofline": roofline, "kernels": kernels, "amdahl": amdahl, "track": track,
        "pcg_iterations_per_lm_step": (pcg_total / float(done)) if solver != "direct" else None,
        "solver_stats": problem.solver_stats(),
        # the direct solver on the same problem (the reference's step to rounding): what the inexact default is judged against
        "direct_it_per_s": direct["it_per_s"] if direct else None, "direct": direct,
        "rmse_delta_vs_direct_px": abs(rmse - direct["final_rmse_px"]) if direct else None, "pose_delta_vs_direct": pose_delta, "block_storage": block_storage,
        # N = 1: configs 4 and 5 on the same GPU through the same default path (other_workload above)
        "other_workloads": others,
        # multi-GPU bookkeeping: ranks RCCL itself reports for the communicator, observations per rank (frame-range shards
        # balanced by observation count), payload of ONE all-reduce of the reduced system (packed lower triangle | rhs | g0 | scalars)
        "ranks_seen": comm_stats["ranks_seen"] if comm_stats else 1, "local_obs": per_rank_obs,
        "allreduce_bytes": comm_stats["system_allreduce_bytes"] if comm_stats else 0,
        "allreduce_calls": comm_stats["allreduce_calls"] if comm_stats else 0,
        "device_wakeup": wakeup,   # untimed whole solves before the W warm-up steps (see above); not part of any reported time
        # N > 1: configs 4 and 5 under the same communicator with the solver AUTO picks for them (scaling_workload above)
        "scaling_workloads": scaling,
    }
    if world == 1 and not args.no_cpu_baseline:
        threads = os.cpu_count() or 1
        cb = cpu_baseline(ds, args.workload, threads)
        out["cpu_baseline"] = cb
        if cb.get("value"):
            out["gpu_over_cpu"] = out["value"] / cb["value"]
        # accuracy half of the metric: |RMSE(GPU) - RMSE(reference-faithful CPU, full solve)|, only where the CPU solve is seconds
        if args.workload == 2:
            import oracle_lib as ol
            o = ol.Oracle(ds)
            xc, _ = (o.ref_lm_solve if ol.have_ref() else o.lm_solve)(ds.x_full, jac_mode=ol.JAC_NUMERIC_F32, res_mode=ol.RES_F32, threads=threads)
            out["rmse_delta_vs_cpu_px"] = abs(rmse - o.reproj_stats(xc)["rmse"])
        if track is not None and args.workload <= 3:
            import oracle_lib as ol
            nf = min(ds.num_frames, 100)
            sub = aar.Dataset.__new__(aar.Dataset)
            sub.__dict__.update(ds.__dict__)
            keep = ds.obs_frame < nf
            for k in ("obs_frame", "obs_cam", "obs_marker", "obs_uv"):
                setattr(sub, k, getattr(ds, k)[keep])
            sub.num_obs, sub.num_frames, sub.frame_ids = int(keep.sum()), nf, ds.frame_ids[:nf]
            t2 = time.perf_counter()
            ol.track_frames(sub, x_tr[: ns + 6 * nf], use_ref=ol.have_ref())
            track["cpu_reference_frames_per_s"] = nf / (time.perf_counter() - t2)
            track["cpu_sample"] = "first %d frames, real SparseLevMarq::solve(z,f) with its own calcDerivates, 1 thread" % nf
    else:
        out["cpu_baseline"] = None
    emit_result(json.dumps(out))
    problem.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
