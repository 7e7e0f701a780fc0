#!/bin/bash
# the driver's own command three times: value, amdahl bounds, wall seconds
for i in 1 2 3; do
  t0=$(date +%s.%N)
  python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/b.err > gpurun_out/b.json
  t1=$(date +%s.%N)
  python - "$t0" "$t1" <<'PY'
import json, sys
b = json.load(open("gpurun_out/b.json"))
print("%.1f s wall  %.1f it/s  bound_at %s  other %s" % (float(sys.argv[2]) - float(sys.argv[1]), b["value"], {k: round(v, 2) for k, v in b["amdahl"]["bound_at"].items()},
      {k: (round(v["value"], 1), round(v["amdahl"]["bound_at"]["8"], 2)) for k, v in b["other_workloads"].items()}))
PY
done
