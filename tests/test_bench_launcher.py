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
    out = run(["--gpus", "2", "--plumbing-only", "--workload", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                      # rank 0's line and nothing else on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["plumbing_only"] is True
    assert len(rec["local_obs"]) == 2 and sum(rec["local_obs"]) == rec["config"]["marker_observations"]
    assert max(rec["local_obs"]) / (sum(rec["local_obs"]) / 2.0) < 1.1          # balanced by observation count
    assert abs(rec["max_rank_seconds"] - 0.002) < 1e-12                          # MAX over ranks (rank r contributed 0.001 (r + 1))


def test_child_failure_propagates_and_prints_no_result():
    out = run(["--gpus", "2", "--plumbing-only", "--workload", "2"], {"AAR_BENCH_FAIL_RANK": "1"})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert "child run failed" in out.stderr


def test_single_rank_needs_no_launcher():
    out = run(["--gpus", "1", "--plumbing-only", "--workload", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["local_obs"] == [rec["config"]["marker_observations"]]


def test_world_size_mismatch_is_refused():
    out = run(["--gpus", "4", "--plumbing-only"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)
