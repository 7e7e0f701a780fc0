"""bench.py --gpus N launches its own ranks: the plumbing of that path on a CPU-only machine.

`--plumbing-only` runs everything of an N-rank benchmark run that needs no GPU: the parent spawns `python -m torch.distributed.run`
children (never re-executing itself), the ranks meet over gloo on 127.0.0.1, plan the frame-range shards, gather the per-rank
observation counts, take the max-over-ranks of a timing, rank 0 prints ONE JSON line and the parent relays it.  The data path
itself (RCCL all-reduce of the reduced system inside libaar) needs GPUs and is covered by the -m gpu tests.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if env_extra:
        env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_two_rank_self_launch_relays_one_json_line():
    out = run(["--gpus", "2", "--plumbing-only", "--workload", "2"], {"AAR_BENCH_PLUMBING_FRAMES5": "200"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                      # rank 0's line and nothing else on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["plumbing_only"] is True
    assert len(rec["local_obs"]) == 2 and sum(rec["local_obs"]) == rec["config"]["marker_observations"]
    assert max(rec["local_obs"]) / (sum(rec["local_obs"]) / 2.0) < 1.1          # balanced by observation count
    assert abs(rec["max_rank_seconds"] - 0.002) < 1e-12                          # MAX over ranks (rank r contributed 0.001 (r + 1))
    # N > 1: the workloads BASELINE.json shards over 8 GPUs ride in the same line (scaling_workloads): here their shard plans, agreed on by the ranks
    sw = rec["scaling_workloads"]
    assert set(sw) == {"4", "5"} and "2000-frame" in sw["4"]["workload"] and "16-cam" in sw["5"]["workload"]
    for w in sw.values():
        assert len(w["local_obs"]) == 2 and sum(w["local_obs"]) == w["marker_observations"] and max(w["local_obs"]) / (sum(w["local_obs"]) / 2.0) < 1.1
        assert len(w["frames"]) == 2 and all(f > 0 for f in w["frames"])


def test_child_failure_propagates_and_prints_no_result():
    out = run(["--gpus", "2", "--plumbing-only", "--workload", "2"], {"AAR_BENCH_FAIL_RANK": "1"})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert "child run failed" in out.stderr


def test_a_librarys_c_stdio_output_never_reaches_the_result_stream():
    # librccl greets a new communicator with a banner on the process's standard output through C stdio -- fully buffered on a pipe, so it used to come out at
    # process exit, BEHIND the result line (found on the GPU box in round 6).  bench.py keeps a private duplicate of descriptor 1 for the line and points the
    # descriptor itself at stderr: whatever a library writes there, and whenever libc flushes it, the driver reads exactly one line
    for gpus in ("1", "2"):
        out = run(["--gpus", gpus, "--plumbing-only", "--workload", "2"], {"AAR_BENCH_TEST_NOISE": "1", "AAR_BENCH_PLUMBING_FRAMES5": "200"})
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and json.loads(lines[0])["plumbing_only"] is True, out.stdout
        assert "noise from a library" in out.stderr


def test_single_rank_needs_no_launcher():
    out = run(["--gpus", "1", "--plumbing-only", "--workload", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["local_obs"] == [rec["config"]["marker_observations"]]


def test_world_size_mismatch_is_refused():
    out = run(["--gpus", "4", "--plumbing-only"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


def test_amdahl_object_and_kernel_models():
    # the `amdahl` object of the bench line: replicated / sharded / collective microseconds per step from the stage timers and the
    # speed-up bound they allow at 1, 2, 4, 8 GPUs (pure arithmetic, checked here on made-up stage times)
    sys.path.insert(0, ROOT)
    import bench
    st = {"chol": 90e-6 * 100, "control": 0.0, "jacobian_normal_eq": 14e-6 * 100, "schur": 20e-6 * 100, "backsub": 6e-6 * 100, "allreduce": 0.0}
    a = bench.amdahl_split(st, 100, 1)
    assert abs(a["replicated_us"] - 90.0) < 1e-9 and abs(a["sharded_us_one_gpu"] - 40.0) < 1e-9 and a["collective_us"] == 0.0 and a["n_gpus"] == 1
    assert abs(a["bound_at"]["1"] - 1.0) < 1e-12 and abs(a["bound_at"]["8"] - 130.0 / 95.0) < 1e-9      # Amdahl: (90 + 40) / (90 + 40 / 8)
    b = bench.amdahl_split(dict(st, allreduce=17e-6 * 100, jacobian_normal_eq=7e-6 * 100, schur=10e-6 * 100, backsub=3e-6 * 100), 100, 2)   # a 2-rank run
    assert abs(b["sharded_us_one_gpu"] - 40.0) < 1e-9 and abs(b["collective_us"] - 17.0) < 1e-9
    assert abs(b["bound_at"]["2"] - 130.0 / (90.0 + 20.0 + 17.0)) < 1e-9
    # every kernel the library can time has a bound and a flop / byte model; k_ldl_panel is priced as panel solve + trailing update
    assert bench.KERNEL_BOUND["k_ldl_panel"] == "fp64_mfma" and bench.KERNEL_BOUND["k_ldl_diag"] == "latency" and bench.KERNEL_BOUND["k_spcg"] == "latency"
    nb3 = 96.0 ** 3
    assert bench.algorithmic_flops("k_ldl_panel", 0, 288, 0.0, True) == ((2 * nb3 + 4 * nb3) + (nb3 + nb3)) / 2      # m = 2 and m = 1
    assert bench.algorithmic_flops("k_ldl_trsm", 0, 288, 0.0, True) == 0.0                                           # no split stage at three tiles
    assert bench.algorithmic_flops("k_ldl_trsm", 0, 1344, 0.0, True) == sum(m * nb3 for m in range(4, 14)) / 10.0   # the last three block columns take k_ldl_panel
    assert bench.algorithmic_bytes("k_ldl_panel", 0, 48, 500, 288) > bench.algorithmic_bytes("k_ldl_diag", 0, 48, 500, 288)


def test_committed_bench_lines_keep_the_contract():
    # profiles/r06_bench_cfg{3,5}.json and r06_bench_cfg3_driver_flags.json are bench.py's own lines from the GPU box (scripts/collect_profiles.sh, then scripts/collect_bench_lines.sh
    # once the PMC fold exists): the fields the driver and the judge read must be there, with the metric of BASELINE.json, the roofline of the dominant kernel with its PMC traffic, the CPU
    # baseline beside it, what the library's DEFAULT options resolved to, and the direct solver's figures -- error AND poses -- on the same problem
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for name, workload, solver in (("r06_bench_cfg3.json", "8-cam/40-marker/500-frame", "spcg"), ("r06_bench_cfg5.json", "16-cam/200-marker/5000-frame", "pcg")):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "roofline", "cpu_baseline", "amdahl", "final_rmse_px", "direct_it_per_s", "rmse_delta_vs_direct_px", "pose_delta_vs_direct", "lm_iterations_to_stop", "solver_stats"):
            assert k in d, (name, k)
        assert d["metric"].startswith("LM iterations/sec") and base["metric"].startswith("LM iterations/sec")
        assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
        assert workload in d["config"]["workload"] and "model" not in d["config"]
        assert d["config"]["solver"].startswith("library default") and d["config"]["solver_resolved"] == solver == d["solver_stats"]["solver"]
        assert d["solver_stats"]["pcg_eta_loose"] == 0.0 and d["solver_stats"]["env_overrides"] == 0
        assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
        assert d["rmse_delta_vs_direct_px"] < 1e-6 and d["lm_iterations_to_stop"] == d["direct"]["lm_iterations_to_stop"]
        assert max(d["pose_delta_vs_direct"].values()) < 1e-5                      # the default path's final poses as transforms against the direct solver's
        r = d["roofline"]
        assert r["bound"] in ("hbm", "fp64_valu", "fp64_mfma", "latency") and r["kernel"] in r["per_kernel"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["traffic"] is not None and r["traffic"] > 0
    d5 = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_cfg5.json")))["roofline"]
    assert d5["kernel"] == "k_pcg" and d5["utilisation"]["frac"] > 10 * d5["frac"]            # k_pcg twice: its own traffic model (utilisation) and SURVEY 8d's algorithmic fraction
    d3 = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_cfg3.json")))
    assert set(d3["amdahl"]["bound_at"]) == {"1", "2", "4", "8"} and d3["amdahl"]["bound_at"]["1"] == 1.0
    cb = d3["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    # the driver's command (--gpus 1 --steps 20 --warmup 5): configs 4 and 5 ride in the same line
    dd = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_cfg3_driver_flags.json")))
    assert dd["steps"] == 20 and dd["warmup"] == 5 and set(dd["other_workloads"]) == {"4", "5"}
    for w, sol in (("4", "spcg"), ("5", "pcg")):
        o = dd["other_workloads"][w]
        for k in ("value", "ms_per_step", "solver_resolved", "cg_iterations_per_lm_step", "final_rmse_px", "rmse_delta_vs_direct_px", "pose_delta_vs_direct", "iteration_hbm", "roofline", "direct"):
            assert k in o, (w, k)
        assert o["solver_resolved"] == sol and o["rmse_delta_vs_direct_px"] < 1e-6 and max(o["pose_delta_vs_direct"].values()) < 1e-5
        assert o["roofline"]["traffic"] and "fp64_valu" in o["roofline"] and 0 < o["iteration_hbm"]["frac"] < 1
        assert o["roofline"]["traffic_uncorrected"] and o["roofline"]["traffic_uncorrected"] <= o["roofline"]["traffic"]      # (raw FETCH + WRITE beside the doubled-FETCH figure)
        # what a SCALE run can be held against: the workload behind a single-rank RCCL communicator, split and extrapolated
        am = o["amdahl"]
        for k in ("it_per_s_single_rank_rccl", "communicator_overhead_us_per_step", "allreduce_calls_per_lm_step", "allreduce_bytes_per_lm_step", "replicated_us", "sharded_us_one_gpu",
                  "collective_us", "bound_at", "predicted_upper_bound_it_per_s"):
            assert k in am, (w, k)
        assert am["solver_resolved"] == sol and set(am["bound_at"]) == {"1", "2", "4", "8"} and am["bound_at"]["1"] == 1.0 and set(am["predicted_upper_bound_it_per_s"]) == {"2", "4", "8"}
        assert 0 < am["it_per_s_single_rank_rccl"] <= 1.02 * o["value"] and am["predicted_upper_bound_it_per_s"]["8"] > am["it_per_s_single_rank_rccl"]
        assert am["allreduce_calls_per_lm_step"] >= (1.0 if sol == "spcg" else o["cg_iterations_per_lm_step"])
    # config 5 runs PCG with fp32 W blocks (storage only): the line carries the same measurement with fp64 blocks and the distance between the two runs' final poses
    bs = dd["other_workloads"]["5"]["block_storage"]
    assert bs["W"].startswith("fp32 storage") and bs["with_fp64_blocks"]["lm_iterations_to_stop"] == dd["other_workloads"]["5"]["lm_iterations_to_stop"]
    assert 0 < bs["with_fp64_blocks"]["it_per_s"] < dd["other_workloads"]["5"]["value"] and max(bs["pose_delta_fp32_vs_fp64_blocks"].values()) < 1e-7
    assert dd["block_storage"] is None and "block_storage" not in dd["other_workloads"]["4"]          # (SPCG: fp64 throughout)
    # the N > 1 line's extra workloads, as measured behind a single-rank communicator
    sw = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_cfg3_single_rank_rccl.json")))["scaling_workloads"]
    assert sw["4"]["solver_resolved"] == "spcg" and sw["5"]["solver_resolved"] == "pcg" and sw["5"]["value"] > 500 and sw["4"]["amdahl"]["bound_at"]["8"] > 1.5
