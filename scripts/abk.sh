#!/bin/bash
# per-kernel A/B (HIP-event averages inside bench.py): libaar_ab.so (A) against libaar.so (B).  usage: bash scripts/abk.sh [workload] [rounds] [kernel-substring]
W=${1:-3}; R=${2:-2}; K=${3:-ldl}
for i in $(seq $R); do
  for v in A B; do
    L=""; [ $v = A ] && L="$(pwd)/automatic-ar_amd/libaar_ab.so"
    AAR_LIB=$L python bench.py --workload $W --no-cpu-baseline --no-amdahl 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), {k:round(v['avg_us'],2) for k,v in d['kernels'].items() if '$K' in k})"
  done
done
