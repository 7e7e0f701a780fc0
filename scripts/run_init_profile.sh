export TMPDIR=/tmp
ROOT=$(pwd)
python -m pytest tests/test_initializer.py -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests.log
AAR_INIT_VERBOSE=1 python scripts/init_bench.py --frames 500 2000 5000 > gpurun_out/init_bench.log 2>&1
AAR_INIT_VERBOSE=1 python scripts/init_bench.py --cams 16 --markers 200 --frames 1000 5000 >> gpurun_out/init_bench.log 2>&1
python tests/tools/init_oracle_time.py --frames 200 500 >> gpurun_out/init_bench.log 2>&1
cat gpurun_out/init_bench.log
cd /tmp
rm -rf $ROOT/gpurun_out/init_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/init_stats -- python3 $ROOT/scripts/init_bench.py --cams 16 --markers 200 --frames 5000 > $ROOT/gpurun_out/init_stats.log 2>&1
find $ROOT/gpurun_out/init_stats -name '*kernel_trace.csv' -delete; find $ROOT/gpurun_out/init_stats -name '*agent_info.csv' -delete
