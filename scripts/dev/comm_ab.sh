#!/bin/bash
# single-rank RCCL communicator on one GPU: what pack + all-reduce + unpack + the host's turn-around cost per step (before any link latency)
W=${1:-3}
E=""; [ "$W" = 5 ] && E="--steps 45 --warmup 15"
for i in 1 2; do
  for v in nocomm r2 nopack spec; do
    X=""; [ $v = r2 ] && X="AAR_FORCE_COMM=1 AAR_PACK_SYSTEM=1 AAR_SPEC_CHOL=0"; [ $v = nopack ] && X="AAR_FORCE_COMM=1 AAR_SPEC_CHOL=0"; [ $v = spec ] && X="AAR_FORCE_COMM=1"
    env $X python bench.py --workload $W $E --no-cpu-baseline --no-kernel-profile --no-amdahl 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(1e3*d['ms_per_step'],2), d['allreduce_bytes'], d['final_rmse_px'])"
  done
done
