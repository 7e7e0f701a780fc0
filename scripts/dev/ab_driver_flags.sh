#!/bin/bash
# A/B of two builds with the driver's own flags (--steps 20 --warmup 5), alternating: libaar_ab.so (A) against libaar.so (B)
for i in $(seq ${1:-8}); do
  for v in A B; do
    L=""; [ $v = A ] && L="$(pwd)/automatic-ar_amd/libaar_ab.so"
    AAR_LIB=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-amdahl 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(1e3*d['ms_per_step'],2), d['final_rmse_px'])"
  done
done
