#!/bin/bash
# Round artefacts in ONE GPU-box call: bench lines (cfg 2-5, + the intrinsics Config, + the opt-in PCG solver), rocprofv3 kernel stats
# (cfg 3, 5) and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only).  Everything lands under
# gpurun_out/collect/; fold the PMC CSVs afterwards with scripts/pmc_summary.py and copy the summaries into profiles/.
#   gpurun --timeout 1800 -- 'bash scripts/collect_profiles.sh'
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/collect
rm -rf "$OUT"; mkdir -p "$OUT"
for w in 2 3 4 5; do
  extra=""; [ "$w" = 5 ] && extra="--steps 45 --warmup 15"
  timeout 600 python3 bench.py --workload $w $extra 2> "$OUT/bench_cfg$w.err" | grep '^{' > "$OUT/bench_cfg$w.json"
done
timeout 600 python3 bench.py --steps 20 --warmup 5 2> /dev/null | grep '^{' > "$OUT/bench_cfg3_driver_flags.json"
timeout 600 python3 bench.py --workload 3 --intrinsics --no-cpu-baseline 2> "$OUT/bench_cfg3_intr.err" | grep '^{' > "$OUT/bench_cfg3_intrinsics.json"
timeout 600 python3 bench.py --workload 5 --intrinsics --steps 45 --warmup 15 --no-cpu-baseline 2> /dev/null | grep '^{' > "$OUT/bench_cfg5_intrinsics.json"
timeout 600 python3 bench.py --workload 3 --solver pcg --no-cpu-baseline 2> /dev/null | grep '^{' > "$OUT/bench_cfg3_pcg.json"
timeout 600 python3 bench.py --workload 5 --solver pcg --steps 45 --warmup 15 --no-cpu-baseline 2> /dev/null | grep '^{' > "$OUT/bench_cfg5_pcg.json"
AAR_FORCE_COMM=1 timeout 600 python3 bench.py --workload 3 --no-cpu-baseline 2> /dev/null | grep '^{' > "$OUT/bench_cfg3_single_rank_rccl.json"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats3" -- \
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-profile --no-amdahl > "$OUT/stats3.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats5" -- \
  python3 "$ROOT/bench.py" --workload 5 --steps 45 --warmup 15 --no-cpu-baseline --no-kernel-profile --no-amdahl > "$OUT/stats5.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats5pcg" -- \
  python3 "$ROOT/bench.py" --workload 5 --solver pcg --steps 45 --warmup 15 --no-cpu-baseline --no-kernel-profile --no-amdahl > "$OUT/stats5pcg.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc3_$c" -- \
    python3 "$ROOT/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-profile --no-amdahl > "$OUT/pmc3_$c.log" 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc5_$c" -- \
    python3 "$ROOT/bench.py" --workload 5 --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-amdahl > "$OUT/pmc5_$c.log" 2>&1
done
# keep what is judged, drop the bulky traces
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT"; find "$OUT" -type f | head -60
