python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-workloads --no-amdahl 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(1e3*d['ms_per_step'],2), d['pcg_iterations_per_lm_step'], d['roofline']['kernel'], {k:(round(v['avg_us'],1), v['launches']) for k,v in d['kernels'].items()})"
