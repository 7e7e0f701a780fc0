#!/bin/bash
# VALU occupancy of the Initializer's kernels from hardware counters (one derived metric per pass, kernel-trace only):
#   gpurun --timeout 900 -- 'bash scripts/init_pmc.sh'      -> gpurun_out/init_pmc/*  (fold into profiles/ by hand)
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/init_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for c in VALUBusy VALUUtilization; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- \
    python3 "$ROOT/scripts/init_bench.py" --cams 16 --markers 200 --frames 1000 > "$OUT/$c.log" 2>&1
done
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
python3 - <<PY
import csv, glob, collections
for c in ("VALUBusy", "VALUUtilization"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % c):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(c, k, "launches", len(v), "mean %.2f max %.2f" % (sum(v) / len(v), max(v)))
PY
