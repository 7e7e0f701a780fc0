mkdir -p gpurun_out/r6a
scripts/probe/spcg_probe 3 1e-6 1 > gpurun_out/r6a/probe_co.txt 2>&1
scripts/probe/spcg_probe 3 1e-6 0 > gpurun_out/r6a/probe_bj.txt 2>&1
python scripts/dev/spcg_check.py 3 > gpurun_out/r6a/spcg_check.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_solvers.py -x -q -m gpu > gpurun_out/r6a/pytest_solvers.txt 2>&1
bash scripts/abenv.sh "AAR_SPCG_COARSE=0" 3 2 > gpurun_out/r6a/ab3.txt 2>&1
bash scripts/abenv.sh "AAR_SPCG_COARSE=0" 4 2 > gpurun_out/r6a/ab4.txt 2>&1
python scripts/dev/pose_delta.py 3 4 > gpurun_out/r6a/pose.txt 2>&1
for f in probe_co probe_bj spcg_check pytest_solvers ab3 ab4 pose; do echo "== $f"; tail -n 14 gpurun_out/r6a/$f.txt; done
