#!/bin/bash
# After scripts/collect_profiles.sh has run on the GPU box (results merged into gpurun_out/collect/): fold the PMC passes and copy what is
# judged into profiles/ under the round's prefix.   usage: bash scripts/fold_profiles.sh r03
set -eu
R=$1; C=gpurun_out/collect
L() { ls -t $1 | head -1; }
F3=$(L "$C/pmc3_FETCH_SIZE/runc/*_counter_collection.csv"); W3=$(L "$C/pmc3_WRITE_SIZE/runc/*_counter_collection.csv")
F5=$(L "$C/pmc5_FETCH_SIZE/runc/*_counter_collection.csv"); W5=$(L "$C/pmc5_WRITE_SIZE/runc/*_counter_collection.csv")
S3=$(L "$C/stats3/runc/*_kernel_stats.csv"); S5=$(L "$C/stats5/runc/*_kernel_stats.csv"); S5P=$(L "$C/stats5pcg/runc/*_kernel_stats.csv")
python scripts/pmc_summary.py $F3 $W3 3 ${R}_cfg3 $S3
python scripts/pmc_summary.py $F5 $W5 5 ${R}_cfg5 $S5
cp $S3 profiles/${R}_cfg3_rocprofv3_kernel_stats.csv; cp $S5 profiles/${R}_cfg5_rocprofv3_kernel_stats.csv; cp $S5P profiles/${R}_cfg5_pcg_rocprofv3_kernel_stats.csv
for w in 2 3 4 5; do cp $C/bench_cfg$w.json profiles/${R}_bench_cfg$w.json; done
for f in bench_cfg3_driver_flags bench_cfg3_intrinsics bench_cfg5_intrinsics bench_cfg3_pcg bench_cfg5_pcg bench_cfg3_single_rank_rccl; do cp $C/$f.json profiles/${R}_$f.json; done
python - <<PY
import json
for f in ("bench_cfg2","bench_cfg3","bench_cfg4","bench_cfg5","bench_cfg3_driver_flags","bench_cfg3_intrinsics","bench_cfg5_intrinsics","bench_cfg3_pcg","bench_cfg5_pcg","bench_cfg3_single_rank_rccl"):
    d=json.load(open("profiles/${R}_%s.json"%f)); r=d["roofline"]
    print("%-28s %9.1f it/s %8.1f us/step  rmse %.9f  dominant %s (%s, frac %s, traffic %s)" % (f, d["value"], 1e3*d["ms_per_step"], d["final_rmse_px"], r["kernel"], r["bound"], r.get("frac"), r.get("traffic")))
PY
