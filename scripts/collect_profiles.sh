#!/bin/bash
# Round artefacts in ONE GPU-box call: bench lines (cfg 2-5 with the default solver AUTO, + the direct solver, + the intrinsics Config, + the
# single-rank RCCL path with its scaling workloads), rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
# kernel-trace only) for every workload.  Everything lands under gpurun_out/collect/; fold afterwards with scripts/fold_profiles.sh <round>.
#   gpurun --timeout 2700 -- 'bash scripts/collect_profiles.sh'
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/collect
rm -rf "$OUT"; mkdir -p "$OUT"
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
B --workload 3 --intrinsics --no-cpu-baseline > "$OUT/bench_cfg3_intrinsics.json"
B --workload 5 --intrinsics --steps 45 --warmup 15 --no-cpu-baseline > "$OUT/bench_cfg5_intrinsics.json"
AAR_FORCE_COMM=1 AAR_BENCH_SCALING=1 timeout 900 python3 "$ROOT/bench.py" --workload 3 --no-cpu-baseline --no-other-workloads 2>> "$OUT/bench.err" | grep '^{' > "$OUT/bench_cfg3_single_rank_rccl.json"
cd /tmp
for w in 2 3 4 5; do
  extra=""; pm="--steps 60 --warmup 10"; [ "$w" = 5 ] && extra="--steps 45 --warmup 15" && pm="--steps 12 --warmup 3"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats$w" -- \
    python3 "$ROOT/bench.py" --workload $w $extra --no-cpu-baseline --no-kernel-profile --no-amdahl --no-direct --no-other-workloads > "$OUT/stats$w.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc${w}_$c" -- \
      python3 "$ROOT/bench.py" --workload $w $pm --no-cpu-baseline --no-kernel-profile --no-amdahl --no-direct --no-other-workloads > "$OUT/pmc${w}_$c.log" 2>&1
  done
done
# the direct solver's kernels (k_ldl_*; at config 5 also k_schur_mfma): their own trace and PMC passes, merged into the same tables
for w in 3 5; do
  extra=""; pm="--steps 60 --warmup 10"; [ "$w" = 5 ] && extra="--steps 45 --warmup 15" && pm="--steps 12 --warmup 3"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats${w}direct" -- \
    python3 "$ROOT/bench.py" --workload $w $extra --solver direct --no-cpu-baseline --no-kernel-profile --no-amdahl --no-other-workloads > "$OUT/stats${w}direct.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc${w}direct_$c" -- \
      python3 "$ROOT/bench.py" --workload $w $pm --solver direct --no-cpu-baseline --no-kernel-profile --no-amdahl --no-other-workloads > "$OUT/pmc${w}direct_$c.log" 2>&1
  done
done
# hardware counters of the observation passes at config 5 (VALU busy, waiting, LDS): scripts/kernel_pmc.sh prints per-launch averages
cd "$ROOT"
bash scripts/kernel_pmc.sh k_passA 5 12 > "$OUT/kpmc_passA_cfg5.txt" 2>&1
bash scripts/kernel_pmc.sh k_passB 5 12 > "$OUT/kpmc_passB_cfg5.txt" 2>&1
bash scripts/kernel_pmc.sh k_pcg 5 12 > "$OUT/kpmc_pcg_cfg5.txt" 2>&1
rm -rf "$ROOT/gpurun_out/kpmc"
# keep what is judged, drop the bulky traces
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT"; find "$OUT" -type f | head -80
