#!/bin/bash
# A/B at config 5 with fp64 blocks (AAR_PCG_W32=0): scripts/dev/ab_bench5_64.sh "AAR_PCG_RESIDENT64=0" ...
for setting in "$@"; do
  env AAR_PCG_W32=0 $setting python bench.py --gpus 1 --steps ${STEPS:-45} --warmup 15 --workload 5 --no-cpu-baseline --no-other-workloads --no-amdahl > /tmp/ab.json 2>/tmp/ab.err || { echo "$setting: FAILED"; tail -3 /tmp/ab.err; continue; }
  python - "$setting" <<'PY'
import json,sys
b=json.load(open("/tmp/ab.json"))
print("%-32s %7.0f it/s  %.1f us/step  cg %.2f  %s  poses %s" % (sys.argv[1], b["value"], 1e3*b["ms_per_step"], b["pcg_iterations_per_lm_step"] or 0, {k:round(v["avg_us"],1) for k,v in b["kernels"].items() if k in ("k_pcg","k_passA","k_passB","k_backsub")}, b.get("pose_delta_vs_direct")))
PY
done
