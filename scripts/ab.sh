#!/bin/bash
# A/B of two builds on the same box: libaar.so (B) against libaar_ab.so (A), alternating.  usage: bash scripts/ab.sh [workload] [rounds]
W=${1:-3}; R=${2:-3}
E=""; [ "$W" = 5 ] && E="--steps 45 --warmup 15"
for i in $(seq $R); do
  for v in A B; do
    L=""; [ $v = A ] && L="$(pwd)/automatic-ar_amd/libaar_ab.so"
    AAR_LIB=$L python bench.py --workload $W $E --no-cpu-baseline --no-amdahl --no-kernel-profile 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(1e3*d['ms_per_step'],2))"
  done
done
