#!/bin/bash
# VALU / MFMA occupancy of the solver's kernels from hardware counters (one derived metric per pass, kernel-trace only):
#   gpurun --timeout 1200 -- 'bash scripts/solver_pmc.sh'   -> prints a per-kernel summary (copied into profiles/ by hand)
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/solver_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
MF=$(rocprofv3 --list-avail 2>/dev/null | grep -B3 "SQ_VALU_MFMA_BUSY_CYCLES,sum)/(reduce(GRBM_GUI_ACTIVE" | grep Counter_Name | head -1 | awk '{print $3}')
echo "mfma metric: $MF"
cd /tmp
for w in 3 5; do
  extra="--steps 60 --warmup 10"; [ $w = 5 ] && extra="--steps 12 --warmup 3"
  for c in VALUBusy $MF; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/w${w}_$c" -- \
      python3 "$ROOT/bench.py" --workload $w $extra --no-cpu-baseline --no-kernel-profile --no-other-workloads > "$OUT/w${w}_$c.log" 2>&1
  done
done
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
python3 - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$OUT/w*_*")):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            if "aar::" in k: print(tag, k, "launches", len(v), "mean %.2f" % (sum(v) / len(v)))
PY
