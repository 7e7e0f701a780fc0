#!/bin/bash
# Instruction mix, wait cycles and memory-side counters of ONE kernel (name substring), one counter group per rocprofv3 pass (kernel-trace only).
#   gpurun -- 'bash scripts/kernel_pmc3.sh <kernel name substring> <workload> <steps> [ENV=VAL ...]'
set -u
export TMPDIR=/tmp
ROOT=$(pwd); K=$1; W=$2; ST=$3; shift 3
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/kpmc3; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TA_BUSY_avr" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" "VALUBusy" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python3 "$ROOT/bench.py" --workload $W --steps $ST --warmup 4 --no-cpu-baseline --no-kernel-profile --no-amdahl --no-direct --no-other-workloads > "$OUT/$tag.log" 2>&1
  f=$(find "$OUT/$tag" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$K" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Kernel_Name"]:
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-36s per-launch avg %.6g  (launches %d)" % (k, s / n, n))
PY
  else echo "($grp: no counter file)"; tail -2 "$OUT/$tag.log"; fi
done
find "$OUT" -name '*.csv' -size +2M -delete
