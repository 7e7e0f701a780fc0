for w in 3 4 5; do for it in 8 16 32 64 128; do
  AAR_SCHUR_ITEM=$it python bench.py --workload $w --steps 45 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('w',$w,'item',$it,'it/s %.1f'%d['value'],'schur %.1f us'%d['kernels']['k_schur']['avg_us'])"
done; done
