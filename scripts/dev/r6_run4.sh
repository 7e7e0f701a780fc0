mkdir -p gpurun_out/r6d
AAR_LIB=$PWD/automatic-ar_amd/libaar_st.so python scripts/dev/pcg_stamps.py 5 2>&1 | head -3
for v in "AAR_PCG_COARSE=0" "AAR_PCG_COARSE=1" "AAR_PCG_COARSE_FROM=0" "AAR_PCG_COARSE_FROM=6"; do
  env $v python bench.py --workload 5 --steps 45 --warmup 15 --no-cpu-baseline --no-amdahl --no-other-workloads --no-direct 2>gpurun_out/r6d/err.txt | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(1e3*d['ms_per_step'],1), 'cg/step', d['pcg_iterations_per_lm_step'], {k:round(v['avg_us'],1) for k,v in (d['kernels'] or {}).items() if k in ('k_pcg',)})"
done
python scripts/dev/pose_delta.py 5 2>&1 | grep -E "^[0-9]|pcg|auto" | cut -c1-220
