#!/bin/bash
# After scripts/collect_profiles.sh has run on the GPU box (results merged into gpurun_out/collect/): fold the PMC passes and copy what is
# judged into profiles/ under the round's prefix.   usage: bash scripts/fold_profiles.sh r04
set -eu
R=$1; C=gpurun_out/collect
L() { ls -t $1 | head -1; }
for w in 2 3 4 5; do
  F=$(L "$C/pmc${w}_FETCH_SIZE/runc/*_counter_collection.csv"); W=$(L "$C/pmc${w}_WRITE_SIZE/runc/*_counter_collection.csv"); S=$(L "$C/stats$w/runc/*_kernel_stats.csv")
  python scripts/pmc_summary.py $F $W $w ${R}_cfg$w $S
  cp $S profiles/${R}_cfg${w}_rocprofv3_kernel_stats.csv
  cp $C/bench_cfg$w.json profiles/${R}_bench_cfg$w.json
done
for w in 3 5; do
  F=$(L "$C/pmc${w}direct_FETCH_SIZE/runc/*_counter_collection.csv"); W=$(L "$C/pmc${w}direct_WRITE_SIZE/runc/*_counter_collection.csv"); S=$(L "$C/stats${w}direct/runc/*_kernel_stats.csv")
  python scripts/pmc_summary.py $F $W $w ${R}_cfg${w}_direct $S --merge
  cp $S profiles/${R}_cfg${w}_direct_rocprofv3_kernel_stats.csv
done
NAMES="bench_cfg3_driver_flags bench_cfg3_direct bench_cfg5_direct bench_cfg5_spcg bench_cfg3_pcg bench_cfg5_pcg bench_cfg3_intrinsics bench_cfg5_intrinsics bench_cfg3_single_rank_rccl bench_cfg3_deterministic bench_cfg5_deterministic"
[ -f $C/bench_cfg3_deterministic.json ] || NAMES=$(echo $NAMES | sed "s/ bench_cfg3_deterministic bench_cfg5_deterministic//")
[ -f $C/bench_cfg5_pcg.json ] || NAMES=$(echo $NAMES | sed "s/ bench_cfg5_pcg//")
[ -f $C/bench_cfg3_round4_forcing_sequence.json ] && NAMES="$NAMES bench_cfg3_round4_forcing_sequence"
[ -f $C/bench_cfg3_driver_flags_repeats.txt ] && cp $C/bench_cfg3_driver_flags_repeats.txt profiles/${R}_bench_cfg3_driver_flags_repeats.txt
for f in $NAMES; do cp $C/$f.json profiles/${R}_$f.json; done
python - <<PY
import json
for f in ["bench_cfg2","bench_cfg3","bench_cfg4","bench_cfg5"] + "$NAMES".split():
    d=json.load(open("profiles/${R}_%s.json"%f)); r=d["roofline"]
    print("%-30s %9.1f it/s %8.1f us/step  solver %-6s rmse %.9f  dominant %s (%s, frac %s, traffic %s)" % (f, d["value"], 1e3*d["ms_per_step"], d["config"].get("solver_resolved"), d["final_rmse_px"], r["kernel"], r["bound"], r.get("frac"), r.get("traffic")))
PY
