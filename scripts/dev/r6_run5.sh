python scripts/dev/r6_dbg.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pcg" 2>&1 | grep -E "passed|failed|Error|assert|error" | tail -8
for v in "AAR_PCG_COARSE=1" "AAR_PCG_COARSE=0"; do env $v AAR_FORCE_COMM=1 python bench.py --workload 5 --steps 45 --warmup 15 --no-cpu-baseline --no-other-workloads --no-kernel-profile --no-amdahl 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('single-rank rccl cfg5 $v', round(d['value'],1), round(1e3*d['ms_per_step'],1), 'cg/step', d['pcg_iterations_per_lm_step'], d['final_rmse_px'])"; done
