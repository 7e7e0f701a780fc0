#!/bin/bash
# The bench lines of a round, taken AFTER scripts/fold_profiles.sh has written profiles/pmc_traffic.json for the final kernel sources, so that every line
# carries the PMC traffic of its dominant kernel (bench.py attaches it only when the sources' hash matches).  Lands in gpurun_out/collect/ like the rest.
#   gpurun --timeout 1800 -- 'bash scripts/collect_bench_lines.sh'
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/collect
mkdir -p "$OUT"
B() { timeout 900 python3 "$ROOT/bench.py" "$@" --no-other-workloads 2>> "$OUT/bench.err" | grep '^{'; }
for w in 2 3 4 5; do
  extra=""; [ "$w" = 5 ] && extra="--steps 45 --warmup 15"
  B --workload $w $extra > "$OUT/bench_cfg$w.json"
done
timeout 900 python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 2>> "$OUT/bench.err" | grep '^{' > "$OUT/bench_cfg3_driver_flags.json"   # the driver's command: other_workloads included
B --workload 3 --solver direct --no-cpu-baseline > "$OUT/bench_cfg3_direct.json"
B --workload 5 --solver direct --steps 45 --warmup 15 --no-cpu-baseline > "$OUT/bench_cfg5_direct.json"
B --workload 5 --solver spcg --steps 45 --warmup 15 --no-cpu-baseline > "$OUT/bench_cfg5_spcg.json"
B --workload 3 --solver pcg --no-cpu-baseline > "$OUT/bench_cfg3_pcg.json"
B --workload 5 --solver pcg --steps 45 --warmup 15 --no-cpu-baseline > "$OUT/bench_cfg5_pcg.json"
B --workload 3 --solver spcg --pcg-eta-loose 0.1 --pcg-eta 0.02 --pcg-abs-tol 1 --no-cpu-baseline > "$OUT/bench_cfg3_round4_forcing_sequence.json"   # round 4's default forcing sequence, for comparison (opt-in now)
B --workload 3 --intrinsics --no-cpu-baseline > "$OUT/bench_cfg3_intrinsics.json"
B --workload 5 --intrinsics --steps 45 --warmup 15 --no-cpu-baseline > "$OUT/bench_cfg5_intrinsics.json"
B --workload 3 --deterministic --no-cpu-baseline > "$OUT/bench_cfg3_deterministic.json"
B --workload 5 --deterministic --steps 45 --warmup 15 --no-cpu-baseline > "$OUT/bench_cfg5_deterministic.json"
AAR_FORCE_COMM=1 AAR_BENCH_SCALING=1 timeout 900 python3 "$ROOT/bench.py" --workload 3 --no-cpu-baseline --no-other-workloads 2>> "$OUT/bench.err" | grep '^{' > "$OUT/bench_cfg3_single_rank_rccl.json"
for i in 1 2 3 4 5 6 7 8; do B --steps 20 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'])"; done > "$OUT/bench_cfg3_driver_flags_repeats.txt"
ls -la "$OUT"/*.json | head -30
