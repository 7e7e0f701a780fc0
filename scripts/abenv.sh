#!/bin/bash
# A/B of one build under two environments on the same box, alternating.  usage: bash scripts/abenv.sh "VAR=val [VAR2=val2]" [workload] [rounds]
# A = with the given variables set, B = without.
ENVA=$1; W=${2:-3}; R=${3:-3}
E=""; [ "$W" = 5 ] && E="--steps 45 --warmup 15"
for i in $(seq $R); do
  for v in A B; do
    X=""; [ $v = A ] && X="$ENVA"
    env $X python bench.py --workload $W $E --no-cpu-baseline --no-kernel-profile --no-amdahl 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(1e3*d['ms_per_step'],2), d['final_rmse_px'])"
  done
done
