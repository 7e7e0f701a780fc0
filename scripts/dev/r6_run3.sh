mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6c/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r6c/pytest_gpu.txt
bash scripts/abenv.sh "AAR_SPCG_COARSE=0" 4 2 > gpurun_out/r6c/ab4.txt 2>&1; cat gpurun_out/r6c/ab4.txt
python scripts/dev/pose_delta.py 3 4 g1_cfg2 g1_cfg2_retry g1_cfg2_far g1_cfg3_cut > gpurun_out/r6c/pose.txt 2>&1; grep -E "^[0-9g]|auto|spcg " gpurun_out/r6c/pose.txt | cut -c1-230
