#!/bin/bash
# LDS activity and bank conflicts of every solver kernel of one workload (one rocprofv3 counter pass, kernel-trace only).
#   gpurun -- 'bash scripts/lds_conflicts.sh <workload> <steps>'
set -u
export TMPDIR=/tmp
ROOT=$(pwd); W=$1; ST=$2
OUT=$ROOT/gpurun_out/ldsc; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p" -- python3 "$ROOT/bench.py" --workload $W --steps $ST --warmup 4 --no-cpu-baseline --no-kernel-profile --no-other-workloads > "$OUT/run.log" 2>&1
f=$(find "$OUT/p" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r'\(.*', '', row["Kernel_Name"]).replace('aar::', '').replace('void ', '')
    a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("%-28s %8s %12s %12s %12s %8s" % ("kernel", "launches", "gui_cycles", "lds_active", "bank_confl", "confl/act"))
for k, c in sorted(acc.items()):
    g = c["GRBM_GUI_ACTIVE"]; n = max(g[1], 1)
    act = c["SQ_LDS_IDX_ACTIVE"][0] / n; bc = c["SQ_LDS_BANK_CONFLICT"][0] / n
    print("%-28s %8d %12.4g %12.4g %12.4g %8.2f" % (k[:28], n, g[0] / n, act, bc, bc / act if act else 0.0))
PY
find "$OUT" -name '*.csv' -size +2M -delete
