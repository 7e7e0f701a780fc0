#!/bin/bash
# Per-kernel hardware counters of one kernel of the solver (one counter group per rocprofv3 pass, kernel-trace only).
#   gpurun -- 'bash scripts/kernel_pmc.sh <kernel name substring> <workload> <steps> [ENV=VAL ...]'
set -u
export TMPDIR=/tmp
ROOT=$(pwd); K=$1; W=$2; ST=$3; shift 3
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/kpmc; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for grp in "MfmaUtil" "VALUBusy" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python3 "$ROOT/bench.py" --workload $W --steps $ST --warmup 4 --no-cpu-baseline --no-kernel-profile --no-amdahl --no-direct --no-other-workloads > "$OUT/$tag.log" 2>&1
  f=$(find "$OUT/$tag" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" "$K" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Kernel_Name"]:
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-28s per-launch avg %.6g  (launches %d)" % (k, s / n, n))
PY
done
find "$OUT" -name '*.csv' -size +2M -delete
