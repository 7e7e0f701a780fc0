mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
R=$(pwd)
python scripts/dev/spcg_check.py 3 2>&1 | tail -5
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6b/prof -- python3 $R/bench.py --workload 3 --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-profile --no-amdahl --no-other-workloads --no-direct > $R/gpurun_out/r6b/bench_prof.txt 2>&1
cd $R
find gpurun_out/r6b/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r6b/kernel_stats.csv
rm -rf gpurun_out/r6b/prof
head -7 gpurun_out/r6b/kernel_stats.csv | cut -c1-140
for f in 0 8 10 12 14 16; do echo "AAR_SPCG_COARSE_FROM=$f"; AAR_SPCG_COARSE_FROM=$f python bench.py --workload 3 --no-cpu-baseline --no-kernel-profile --no-amdahl --no-other-workloads --no-direct 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(1e3*d['ms_per_step'],2), d['pcg_iterations_per_lm_step'])"; done
bash scripts/abenv.sh "AAR_SPCG_COARSE=0" 3 2
