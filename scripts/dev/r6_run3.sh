mkdir -p gpurun_out/r6c
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r6c/pytest_gpu.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r6c/pytest_gpu.txt | tail -8
