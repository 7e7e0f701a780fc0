#!/bin/bash
# Memory-path counters of one kernel (one rocprofv3 pass per counter, kernel-trace only):  gpurun -- 'bash scripts/kernel_pmc2.sh <kernel substring> <workload> <steps>'
set -u
export TMPDIR=/tmp
ROOT=$(pwd); K=$1; W=$2; ST=$3; shift 3
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/kpmc2; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for grp in "MemUnitBusy" "MemUnitStalled" "L2CacheHit" "LDSBankConflict" "TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum" "TCC_HIT_sum" "TCC_MISS_sum" "TCC_BUSY_avr" "TCC_EA_RDREQ_sum" "SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS" "SQ_INSTS_VALU" "SQ_INSTS_SALU" "SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM" "SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python3 "$ROOT/bench.py" --workload $W --steps $ST --warmup 4 --no-cpu-baseline --no-kernel-profile --no-amdahl --no-direct --no-other-workloads > "$OUT/$tag.log" 2>&1
  f=$(find "$OUT/$tag" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$K" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Kernel_Name"]:
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-32s per-launch avg %.6g  (launches %d)" % (k, s / n, n))
PY
  else echo "$grp: not collected ($(tail -1 $OUT/$tag.log | cut -c1-120))"; fi
done
rm -rf "$OUT"
