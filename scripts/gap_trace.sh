#!/bin/bash
# Idle time between consecutive kernels of the LM step (rocprofv3 kernel trace of a short bench run).
#   gpurun -- 'bash scripts/gap_trace.sh <workload> <steps>'
set -u
export TMPDIR=/tmp
ROOT=$(pwd); W=$1; ST=$2
OUT=$ROOT/gpurun_out/gaps; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 "$ROOT/bench.py" --workload $W --steps $ST --warmup 20 --no-cpu-baseline --no-kernel-profile --no-other-workloads --no-amdahl --no-direct > "$OUT/run.log" 2>&1
f=$(find "$OUT/t" -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, re, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("aar::", "").replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))))
gap = collections.defaultdict(lambda: [0, 0]); dur = collections.defaultdict(lambda: [0, 0])
for (s0, e0, k0), (s1, e1, k1) in zip(rows[:-1], rows[1:]):
    g = gap[(k0[:18], k1[:18])]; g[0] += s1 - e0; g[1] += 1
for s, e, k in rows:
    d = dur[k[:18]]; d[0] += e - s; d[1] += 1
print("transition                                   count   avg gap us")
for (a, b), (t, n) in sorted(gap.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%-20s -> %-20s %6d %10.2f" % (a, b, n, t / n / 1e3))
tot_gap = sum(t for t, n in gap.values()); tot_dur = sum(t for t, n in dur.values())
print("total kernel time %.1f ms, total gaps %.1f ms" % (tot_dur / 1e6, tot_gap / 1e6))
PY
rm -rf "$OUT/t"
