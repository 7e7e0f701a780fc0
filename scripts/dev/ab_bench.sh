#!/bin/bash
# A/B of tuning switches on the GPU box: LM it/s of bench.py (config 3 by default) per setting, kernels' HIP-event averages
#   scripts/dev/ab_bench.sh "AAR_SCHUR2=0" "AAR_SCHUR2=1" ...      (WORKLOAD=4 for another config)
W=${WORKLOAD:-3}
for setting in "$@"; do
  env $setting python bench.py --gpus 1 --steps ${STEPS:-200} --warmup 20 --workload $W --no-cpu-baseline --no-other-workloads --no-amdahl > /tmp/ab.json 2>/tmp/ab.err || { echo "$setting: FAILED"; tail -3 /tmp/ab.err; continue; }
  python - "$setting" <<'PY'
import json,sys
b=json.load(open("/tmp/ab.json"))
print("%-40s %8.0f it/s  %.1f us/step  cg %.2f  %s" % (sys.argv[1], b["value"], 1e3*b["ms_per_step"], b["pcg_iterations_per_lm_step"] or 0, {k:round(v["avg_us"],1) for k,v in b["kernels"].items() if k not in ("k_unpack","k_reduce_scalars","k_frame_inv")}))
PY
done
